import sys, json, itertools
sys.path.insert(0,'/root/repo/tools/calib'); sys.path.insert(0,'/root/repo')
from calib import *
REF={'chr21_1mb':0.8447,'chr22_1mb':0.7393,'chr20_1mb':0.8353,'chr13_1mb':0.9152,'chr19_500kb':0.8311,'chr4_1mb':0.9488,'chr1_500kb':0.8722,'chr21_500kb':0.9090}
def run(cid,nrep=4,nmin=3000,**kw):
    IF=load(cid); n=len(IF); d10=O.if_to_dist10(IF); rr=O.dist_to_rr(d10)
    m=O.default_model(n,**kw); fire=O.default_fire(); st=O.make_stages(schedule(nmin))
    res=[]
    for r in range(nrep):
        X,v,ev=O.run_schedule(m,d10,st,fire,82364,r)
        F,e=O.energy_force(m,d10,X,1,1,0.85)
        s=stats(cid,IF,X,rr); s['enoe']=e[0]; res.append(s)
    res.sort(key=lambda s:s['enoe'])
    return res
if __name__=='__main__':
    cids=sys.argv[1].split(',')
    grid=json.loads(sys.argv[2])
    keys=list(grid)
    for vals in itertools.product(*[grid[k] for k in keys]):
        kw=dict(zip(keys,vals))
        out=[]
        for cid in cids:
            res=run(cid,**kw)
            best=res[0]
            out.append(f"{cid}: sp1={best['sp']:.3f} (ref {REF[cid]:.3f}) spmax={max(r['sp'] for r in res):.3f} E={best['enoe']:.0f} b={best['bond'][0]:.2f}±{best['bond'][1]:.2f}[{best['bond'][2]:.1f},{best['bond'][3]:.1f}] cl={best['clash']} rg={best['rg']:.1f} i2={best['i2'][0]:.1f}±{best['i2'][1]:.1f}")
        print(kw,' | '.join(out),flush=True)
