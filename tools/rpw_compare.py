"""Debug aid: k_step at every rows-per-wave setting and the cluster kernel (optionally at a forced geometry,
option cluster_geometry in the dict below) against k_step<rows per wave 2>: all must agree bit for bit."""
import sys
import numpy as np
sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, default_schedule
from tests.util import load_if
def state(cid, k, **opts):
    s = Solver(0)
    s.set_model(default_model()); s.set_if_matrix(load_if(cid))
    s.set_schedule(default_schedule(300), None, 0.0, 250)
    s.set_option("resident_min_ops", 1)
    for a, b in opts.items(): s.set_option(a, b)
    s.init_replicas(1, 82364, 0)
    s.run_steps(k)
    return s.coords(), s.velocities(), (s.stat("cluster_compute_waves"), s.stat("cluster_rows_per_wave"), s.stat("cluster_parts"), s.stat("last_path"))
cid = sys.argv[1]
ks = [int(a) for a in sys.argv[2:]] or [6, 16, 300]
for k in ks:
    ref = state(cid, k, resident=0, rows_per_wave=2, use_graph=0)
    for name, o in (("rpw1", dict(resident=0, rows_per_wave=1, use_graph=0)), ("rpw4", dict(resident=0, rows_per_wave=4, use_graph=0)), ("cluster", dict(resident=1))):
        x, v, g = state(cid, k, **o)
        bad = sorted(set(int(b[1]) for b in np.argwhere(np.abs(v - ref[1]) > 0)))
        print(cid, "k", k, name, g if name == "cluster" else "", "dx", np.abs(x - ref[0]).max(), "dv", np.abs(v - ref[1]).max(), "beads", bad[:24], flush=True)
