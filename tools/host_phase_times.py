"""Host time of every call of a job through the C ABI, ms per chromosome over the eight largest 500 kb matrices, one after the other on one
context (round 6: found c3d_set_if_matrix at 15-18 ms with the IF-rank worker started late).   python tools/host_phase_times.py [prefetch_ranks 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from chromosome3d_amd import batch
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
mats = batch.load_matrices(os.path.join(root, "tests", "golden", "all45"), "_500kb", set())
cids=list(mats)[:8]
s=Solver(0)
if len(sys.argv) > 1: s.set_option("prefetch_ranks", float(sys.argv[1]))
acc={}
def T(name, f):
    t=time.perf_counter(); r=f(); acc[name]=acc.get(name,0)+time.perf_counter()-t; return r
for rep in range(2):
    acc.clear()
    for cid in cids:
        IF=mats[cid]
        T("set_option", lambda: (s.set_option("cluster_xcd_base",0), s.set_option("cluster_xcd_count",8)))
        T("set_model", lambda: s.set_model(default_model()))
        T("K1", lambda: pipeline.IF2dist_new(s, IF))
        T("set_schedule", lambda: s.set_schedule(default_schedule(3000), None, 1e-2, 250))
        T("init_replicas", lambda: s.init_replicas(20, 82364, 0))
        T("run", lambda: s.run())
        T("coords", lambda: s.coords())
        T("energies", lambda: s.energies())
        T("score", lambda: s.score(IF))
print({k: round(1e3*v/len(cids),2) for k,v in acc.items()}, "ms per chromosome; total", round(1e3*sum(acc.values())/len(cids),2))
