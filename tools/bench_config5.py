"""Config 5 (synthetic N = 2500, 8 replicas): microseconds per SA step, symmetric-tile kernels against the full-row step kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import synthetic_if
IF = synthetic_if(2500)[0]
s = Solver(0)
for opts in ({"symmetric": 1}, {"symmetric": 1, "replica_groups": 1}, {"symmetric": 0}, {"symmetric": 0, "rows_per_wave": 4}):
    for k, v in opts.items(): s.set_option(k, v)
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(300), None, 0.0, 250)
    s.init_replicas(8, 82364, 0); s.run_steps(10 ** 7)
    s.init_replicas(8, 82364, 0); s.run_steps(10 ** 7)
    ms, steps, la = s.last_timing()
    n = 2500; R = s.num_restraints; B = 4 * R + 72 * n
    print(f"{opts}: {1e3 * ms / steps:.2f} us/step, algorithmic {8 * steps * B / (ms * 1e-3) / 1e9:.0f} GB/s ({8 * steps * B / (ms * 1e-3) / 8e12:.3f} of 8 TB/s), "
          f"{8 * steps * n * n / (ms * 1e-3) / 1e12:.2f} Tpair/s", flush=True)
    print("   kernel", s.step_kernel_name, flush=True)
    s.set_option("replica_groups", 2); s.set_option("rows_per_wave", 2); s.set_option("symmetric", 0)
