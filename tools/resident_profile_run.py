"""Workload for `rocprofv3 --kernel-trace --stats`: resident-kernel anneals of two small matrices (20 replicas)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import load_if
s = Solver(0)
for cid in ("chr21_1mb", "chr19_500kb"):
    IF = load_if(cid)
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), None, 0.0, 250)
    for rep in range(3):
        s.init_replicas(20, 82364, 20 * rep); s.run_steps(10 ** 7)
    ms, steps, la = s.last_timing()
    print(f"{cid}: N={IF.shape[0]} 20 replicas: {steps} steps in {ms:.2f} ms = {1e3 * ms / steps:.2f} us/step, {la} launch(es)", flush=True)
