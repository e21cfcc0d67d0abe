import sys
import numpy as np
sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, make_stages
from tests.util import load_if
def state(cid, k, stages, **opts):
    s = Solver(0)
    s.set_model(default_model()); s.set_if_matrix(load_if(cid))
    s.set_schedule(make_stages(stages), None, 0.0, 250)
    s.set_option("resident_min_ops", 1)
    for a, b in opts.items(): s.set_option(a, b)
    s.init_replicas(1, 82364, 0)
    s.run_steps(k)
    return s.coords(), s.velocities()
cid = sys.argv[1]
for name, st in (("md", [(0, 400, 0.003, 0.4, 0.003, 0.9, 2000.0)]), ("fire", [(2, 400, 0.0, 1.0, 1.0, 0.85, 0.0)])):
    for k in (1, 2, 3, 4, 5, 6, 7, 8):
        ref = state(cid, k, st, resident=0, use_graph=0)
        x, v = state(cid, k, st, resident=1)
        badv = np.argwhere(np.abs(v - ref[1]) > 0)
        print(cid, name, "k", k, "dx", np.abs(x - ref[0]).max(), "dv", np.abs(v - ref[1]).max(), "nbad", len(badv), "of", v.size, "beads", sorted(set(int(b[1]) for b in badv))[:16], flush=True)
