import sys, json, os, time
sys.path.insert(0,'/root/repo')
import numpy as np
from oracle import oracle as O
def load(cid):
    p=f'/root/repo/tests/golden/inputs/{cid}_matrix.txt'
    if os.path.exists(p): return O.parse_if_text(open(p,'rb').read())
    z=np.load(f'/root/repo/tests/golden/inputs/{cid}_upper.npz'); n=int(z['n']); m=np.zeros((n,n)); iu=np.triu_indices(n); m[iu]=z['upper']; m.T[iu]=z['upper']; return m
def schedule(nmin=3000, hot_T=2000., cool_w_noe=1.0):
    rows=[(2,200,0.0,1.0,20.,0.5,0.)]
    for n,w,wv,rs in [(125,.1,20,.5),(125,.2,20,.5),(125,.2,.01,.9),(500,.4,.003,.9),(125,1.,.003,.9)]:
        rows.append((0,n,0.003,w,wv,rs,hot_T))
    ncycle=int(hot_T/25); nstep=int(1000/ncycle)
    vdw_step=(4.0/0.003)**(1./ncycle); rad_step=(1.0-0.85)/ncycle
    radius=1.0; kv=0.003; bath=hot_T
    for i in range(ncycle+1):
        rows.append((1,nstep,0.005,1.0,kv,radius,bath))
        radius=max(0.85,radius-rad_step); kv=min(4.0,kv*vdw_step); bath-=25.
    rows.append((2,nmin,0.0,1.0,1.0,0.85,0.))
    return rows
def stats(cid,IF,X,rr):
    n=len(X)
    b=np.linalg.norm(X[1:]-X[:-1],axis=1)
    d=np.linalg.norm(X[:,None]-X[None],axis=-1); iu=np.triu_indices(n,1)
    sat,dev=O.assess(np.round(X,3),rr)
    rg=np.sqrt(((X-X.mean(0))**2).sum(1).mean())
    b2=np.linalg.norm(X[2:]-X[:-2],axis=1)
    return dict(sp=round(-O.spearman_if_dist(IF,np.round(X,3),3),4), bond=(round(b.mean(),2),round(b.std(),2),round(b.min(),2),round(b.max(),2)), clash=int((d[iu]<=3.5).sum()), sat=sat, dev=round(dev,1), rg=round(rg,2), i2=(round(b2.mean(),2),round(b2.std(),2),round(b2.min(),2)))
if __name__=='__main__':
    cid=sys.argv[1]; kw=json.loads(sys.argv[2]) if len(sys.argv)>2 else {}
    nrep=int(kw.pop('nrep',3)); nmin=int(kw.pop('nmin',3000))
    IF=load(cid); n=len(IF); d10=O.if_to_dist10(IF); rr=O.dist_to_rr(d10)
    fire=O.default_fire(**{k[5:]:v for k,v in kw.items() if k.startswith('fire_')})
    m=O.default_model(n, **{k:v for k,v in kw.items() if not k.startswith('fire_')})
    st=O.make_stages(schedule(nmin))
    for r in range(nrep):
        t=time.time(); X,v,ev=O.run_schedule(m,d10,st,fire,82364,r,gtol=kw.get('gtol',0.0) if False else 0.0)
        F,e=O.energy_force(m,d10,X,1,1,0.85)
        print(cid,r,stats(cid,IF,X,rr),'Enoe',round(e[0],1),'Eb',round(e[1],1),'Erep',round(e[2],1),'|F|rms',round(float(np.sqrt((F**2).mean())),4),'evals',ev,f'{time.time()-t:.1f}s',flush=True)
