"""Measurement: chr1_500kb x 20 with mrswitch = 1 (where 1 / mrswitch = 1 and a build with -DC3D_EXP_FOLD_D computes the same numbers):
us per step of 2000-step multi-step launches.  A/B of two builds (tools/ab pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
import hashlib
s = Solver(0)
s.set_model(default_model(mrswitch=1.0)); pipeline.IF2dist_new(s, load_if("chr1_500kb"))
s.set_schedule(make_stages([(1, 2000, 0.005, 1.0, 0.01, 1.0, 300.0), (2, 2000, 0.0, 1.0, 1.0, 0.85, 0.0)]))
s.set_option("resident", 1)
for rep in range(3):
    s.init_replicas(20, 82364, 0)
    s.run_steps(2001); a = 1e3 * s.last_timing()[0] / 2000
    s.run_steps(2000); b = 1e3 * s.last_timing()[0] / 2000
    print(f"MD {a:.4f} FIRE {b:.4f} us/step  {s.step_kernel_name}  md5 {hashlib.md5(s.coords().tobytes()).hexdigest()[:12]}", flush=True)
