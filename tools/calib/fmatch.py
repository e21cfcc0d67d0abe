import sys, glob
sys.path.insert(0,'/root/repo/tools/calib'); sys.path.insert(0,'/root/repo')
from calib import *
def load_pdb(p):
    return np.array([[float(l[30:38]),float(l[38:46]),float(l[46:54])] for l in open(p) if l.startswith('ATOM')])
for cid in ['chr21_1mb','chr22_1mb','chr20_1mb','chr13_1mb','chr21_500kb','chr19_500kb']:
    IF=load(cid); n=len(IF); d10=O.if_to_dist10(IF)
    X=load_pdb(glob.glob(f'/root/repo/tests/golden/models/{cid}_rank*')[0])
    for pot,kw in [(0,{}),(1,{}),(2,{}),(3,{'masym':0.1}),(3,{'masym':1.0})]:
        m=O.default_model(n,noe_pot=pot,k_bond=0.0,k_rep=0.0,**kw)
        Fn,_=O.energy_force(m,d10,X,1,0,0.85)
        m2=O.default_model(n,noe_pot=pot,k_bond=1.0,s_noe=0.0,k_rep=0.0)
        Fb,_=O.energy_force(m2,d10,X,1,0,0.85)
        # project onto bond directions: only the component of F along chain can be balanced by bonds; do LSQ over k
        k=-(Fn*Fb).sum()/(Fb*Fb).sum()
        res=np.linalg.norm(Fn+k*Fb)/np.linalg.norm(Fn)
        print(cid,'pot',pot,kw,'|Fnoe| rms',round(float(np.sqrt((Fn**2).sum(1).mean())),1),'k_fit',round(k,1),'resid frac',round(res,3))
