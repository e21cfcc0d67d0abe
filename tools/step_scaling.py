"""Diagnostic: device time per SA-step launch vs problem size and replica count."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
s = Solver(0)
rpws = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2]
opts = dict(kv.split("=") for kv in sys.argv[2].split(",")) if len(sys.argv) > 2 else {}
for k, v in opts.items():
    s.set_option(k, float(v)) if False else None
for k, v in opts.items():
    s.set_option(k, float(v))
print("options", opts)
for cid in ("chr21_1mb", "chr1_500kb"):
    IF = load_if(cid)
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    for kind, st in (("fire", [(2, 2000, 0.0, 1.0, 1.0, 0.85, 0.0)]), ("md", [(1, 2000, 0.005, 1.0, 0.01, 1.0, 300.0)])):
        s.set_schedule(make_stages(st))
        for rpw in rpws:
            s.set_option("rows_per_wave", rpw)
            for nrep in (1, 20, 160):
                s.init_replicas(nrep, 1, 0)
                s.run_steps(10**6)            # builds graphs
                s.init_replicas(nrep, 1, 0)
                s.run_steps(10**6)
                ms, steps, la = s.last_timing()
                print(f"{cid:11s} n={IF.shape[0]:4d} {kind:4s} rpw={rpw} nrep={nrep:4d}  {1e3*ms/steps:8.3f} us/step  {nrep*steps/ms*1e3/1e6:8.3f} M replica-steps/s", flush=True)
