"""Fixed cost of one multi-step launch of the cluster kernel: kernel duration (its own start/stop events) against the number
of SA steps in the launch, chr1_500kb x 20; the intercept is the prologue (slot claim, targets into registers, coordinates
into LDS) plus the drain, the slope the step time.    python tools/launch_overhead.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
s = Solver(0)
s.set_model(default_model()); pipeline.IF2dist_new(s, load_if(sys.argv[1] if len(sys.argv) > 1 else "chr1_500kb"))
s.set_schedule(make_stages([(1, 100000, 0.005, 1.0, 0.01, 1.0, 300.0)]))
s.set_option("resident", 1); s.set_option("kernel_timing", 1)
s.init_replicas(20, 82364, 0)
s.run_steps(500)
rows = []
for k in (4, 8, 12, 20, 40, 100, 400, 2000):
    t = []
    for _ in range(30 if k <= 100 else 6):
        s.run_steps(k)
        t.append(s.stat("last_kernel_us"))
    rows.append((k, float(np.median(t))))
    print(f"{k:5d} steps per launch: kernel {rows[-1][1]:9.2f} us = {rows[-1][1] / k:6.3f} us per step")
k = np.array([r[0] for r in rows[:5]], float); t = np.array([r[1] for r in rows[:5]])
a, b = np.polyfit(k, t, 1)
print(f"fit over 4..40 steps: {b:.2f} us per launch + {a:.3f} us per step")
