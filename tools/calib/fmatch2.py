import sys, glob
sys.path.insert(0,'/root/repo/tools/calib'); sys.path.insert(0,'/root/repo')
from calib import *
def load_pdb(p):
    return np.array([[float(l[30:38]),float(l[38:46]),float(l[46:54])] for l in open(p) if l.startswith('ATOM')])
POT=int(sys.argv[1]); MASYM=float(sys.argv[2]); SEPS=eval(sys.argv[3]); allb=[];alla=[]
for cid in ['chr21_1mb','chr22_1mb','chr20_1mb','chr13_1mb','chr21_500kb','chr19_500kb']:
    IF=load(cid); n=len(IF); d10=O.if_to_dist10(IF)
    X=load_pdb(glob.glob(f'/root/repo/tests/golden/models/{cid}_rank*')[0])
    m=O.default_model(n,noe_pot=POT,masym=MASYM,k_bond=0.0,k_rep=0.0)
    Fn,_=O.energy_force(m,d10,X,1,0,0.85)
    cols=[];meta=[]
    for sep in SEPS:
        for i in range(n-sep):
            u=X[i]-X[i+sep]; d=np.linalg.norm(u); u/=d
            c=np.zeros((n,3)); c[i]=u; c[i+sep]=-u   # tension>0 pushes apart
            cols.append(c.ravel()); meta.append((sep,i,d))
    A=np.array(cols).T
    sol,res,rk,sv=np.linalg.lstsq(A,-Fn.ravel(),rcond=None)
    r=np.linalg.norm(A@sol+Fn.ravel())/np.linalg.norm(Fn)
    print(cid,'resid frac with free sep1,sep2 tensions',round(r,3))
    for (sep,i,d),t in zip(meta,sol):
        (allb if sep==1 else alla).append((d,t)) if sep<3 else None
allb=np.array(allb); alla=np.array(alla)
# tension t>0 means pair pushes apart => for bond: t = -dE/dd = -2k(d-b0)
for name,arr,bins in [('bond',allb,np.arange(3.4,5.8,0.2)),('i+2',alla,np.arange(3.0,10.5,0.5))]:
    print(name)
    idx=np.digitize(arr[:,0],bins)
    for b in range(1,len(bins)):
        s=arr[idx==b]
        if len(s): print(f'  d in [{bins[b-1]:.1f},{bins[b]:.1f}) n={len(s):3d} mean tension={s[:,1].mean():9.1f} median={np.median(s[:,1]):9.1f}')
k,b=np.polyfit(allb[:,0],allb[:,1],1); print('bond linear fit: tension = %.1f*d + %.1f => k=%.1f b0=%.2f'%(k,b,-k/2,-b/k))
