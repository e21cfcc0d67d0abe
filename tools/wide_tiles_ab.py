"""Config 5 (synthetic N = 2500, 8 replicas) and two more large problems: the per-step kernel's wide form (16 rows a workgroup, four a wave,
option wide_tiles 1 = default) against the narrow one, microseconds per SA step.  python tools/wide_tiles_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import synthetic_if
s = Solver(0)
for n, nrep in ((2500, 8), (1500, 8), (1100, 20)):
    IF = synthetic_if(n)[0]
    for wide in (1, 0, 1, 0):
        for groups in (2, 1):
            s.set_option("wide_tiles", wide); s.set_option("replica_groups", groups)
            s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
            s.set_schedule(default_schedule(300), None, 0.0, 250)
            s.init_replicas(nrep, 82364, 0); s.run_steps(10 ** 7)
            s.init_replicas(nrep, 82364, 0); s.run_steps(10 ** 7)
            ms, steps, la = s.last_timing()
            print(f"n {n} x {nrep} wide {wide} groups {groups}: {1e3 * ms / steps:.2f} us/step, {nrep * steps * n * n / (ms * 1e-3) / 1e12:.2f} Tpair/s  {s.step_kernel_name}", flush=True)
