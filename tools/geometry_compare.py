"""Cluster-kernel geometries on one problem: microseconds per SA step of the full default schedule (one launch).
    python tools/geometry_compare.py <chromosome id> <replicas> [CWxRPWxNH, e.g. 12x4x4: option cluster_geometry]"""
import os, sys
sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import load_if
cid, nrep = sys.argv[1], int(sys.argv[2])
s = Solver(0)
geom = sys.argv[3] if len(sys.argv) > 3 else "auto"
if geom != "auto":
    cw, rpw, nh = (int(v) for v in geom.split("x"))
    s.set_option("cluster_geometry", 100 * cw + 10 * rpw + nh)
try:
    IF = load_if(cid)
except FileNotFoundError:                         # any of the 45 bundled matrices
    import numpy as np
    z = np.load(os.path.join("tests", "golden", "all45", f"{cid}_upper.npz"))
    n = int(z["n"]); IF = np.zeros((n, n)); iu = np.triu_indices(n); IF[iu] = z["upper"]; IF.T[iu] = z["upper"]
s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
s.set_schedule(default_schedule(3000), None, 0.0, 250)
for _ in range(2):
    s.init_replicas(nrep, 82364, 0); s.run_steps(10 ** 7)
ms, steps, la = s.last_timing()
print(f"{geom:8s} {cid} x{nrep}: {1e3 * ms / steps:.3f} us/step ({la} launches) parts {s.stat('cluster_parts'):.0f} cw {s.stat('cluster_compute_waves'):.0f} rpw {s.stat('cluster_rows_per_wave'):.0f} path {s.stat('last_path'):.0f}")
