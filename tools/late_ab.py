"""Tile sums late (H0 fetches them after the next step has started) against gathered with the rows, per problem: us per SA step of the whole
default schedule in one launch (option cluster_late_tiles 1 / 0; the planner's rule is cluster_late_ok in c3d_cluster.hip).
    python tools/late_ab.py [cid:replicas ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = [a.split(":") for a in sys.argv[1:]] or [("chr1_500kb", "20"), ("chr1_500kb", "10"), ("chr1_500kb", "5"), ("chr3_500kb", "20"), ("chr7_500kb", "20"),
                                                  ("chr10_500kb", "20"), ("chr4_1mb", "20"), ("chr1_1mb", "20"), ("chr13_1mb", "20"), ("chr19_500kb", "20"),
                                                  ("chr18_1mb", "20"), ("chr21_500kb", "20")]
for cid, nrep in cases:
    z = np.load(os.path.join(ROOT, "tests", "golden", "all45", f"{cid}_upper.npz"))
    n = int(z["n"]); IF = np.zeros((n, n)); iu = np.triu_indices(n); IF[iu] = z["upper"]; IF.T[iu] = z["upper"]
    row = []
    for late in (1, 0, 1, 0):
        s = Solver(0)
        s.set_option("cluster_late_tiles", late)
        s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
        s.set_schedule(default_schedule(3000), None, 0.0, 250)
        for _ in range(2):
            s.init_replicas(int(nrep), 82364, 0); s.run_steps(10 ** 7)
        ms, steps, la = s.last_timing()
        row.append((1e3 * ms / steps, s.stat("cluster_late_tiles"), s.stat("cluster_parts"), s.stat("cluster_compute_waves"), s.stat("cluster_rows_per_wave")))
        s.close()
    print(f"{cid:12s} n {n:4d} x{nrep:>2s}  parts {row[0][2]:.0f} cw {row[0][3]:.0f} rpw {row[0][4]:.0f}   late(planner) {row[0][0]:.3f} {row[2][0]:.3f} [late in use {row[0][1]:.0f}]   "
          f"never late {row[1][0]:.3f} {row[3][0]:.3f}", flush=True)
