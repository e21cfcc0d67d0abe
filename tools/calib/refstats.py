import sys, glob
sys.path.insert(0,'/root/repo/tools/calib'); sys.path.insert(0,'/root/repo')
from calib import *
def load_pdb(p):
    return np.array([[float(l[30:38]),float(l[38:46]),float(l[46:54])] for l in open(p) if l.startswith('ATOM')])
for cid in ['chr21_1mb','chr22_1mb','chr20_1mb','chr13_1mb','chr19_500kb','chr21_500kb','chr4_1mb','chr1_500kb']:
    IF=load(cid); n=len(IF); d10=O.if_to_dist10(IF); rr=O.dist_to_rr(d10)
    X=load_pdb(glob.glob(f'/root/repo/tests/golden/models/{cid}_rank*')[0])
    m=O.default_model(n,noe_pot=1)
    F,e=O.energy_force(m,d10,X,1,1,0.85)
    b2=np.linalg.norm(X[2:]-X[:-2],axis=1)
    d=np.linalg.norm(X[:,None]-X[None],axis=-1); 
    print(cid,n,stats(cid,IF,X,rr),'Easym',round(e[0]),'i+2',round(b2.mean(),2),round(b2.std(),2),round(b2.min(),2),'minnb',round(np.sort(d[np.triu_indices(n,2)])[0],2))
