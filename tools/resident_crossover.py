"""Per-step time of the per-step (hipGraph) path vs the resident kernel over every bundled matrix size.

    python tools/resident_crossover.py [replicas=20] [option=value ...]
Needs tests/golden/_all (tools/pack_all_inputs.py).  One line per matrix: N, tiles, us/step of both paths
for the full default schedule (5172 steps, fixed length).
"""
import glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline

ALL = os.path.join(ROOT, "tests", "golden", "_all")
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
opts = [kv.split("=") for kv in sys.argv[2:]]          # e.g. resident_waves=8


def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m


s = Solver(0)
cids = sorted({os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_upper.npz")})
sizes = sorted((load(c).shape[0], c) for c in cids)
seen = set()
for n, cid in sizes:
    if n // 16 in seen:
        continue
    seen.add(n // 16)
    IF = load(cid)
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
    out = []
    for res in (0, 1):
        s.set_option("resident", res)
        for k, v in opts:
            s.set_option(k, float(v))
        for rep in range(2):                      # first pass builds the graphs
            s.init_replicas(nrep, 82364, 0)
            s.run_steps(s.schedule_length)
        ms, steps, launches = s.last_timing()
        out.append(1e3 * ms / steps)
    print(f"{cid:14s} N={n:4d} tiles={(n + 7) // 8:3d} replicas={nrep}: per-step {out[0]:6.2f} us  resident {out[1]:6.2f} us  ratio {out[0] / out[1]:.2f}", flush=True)
