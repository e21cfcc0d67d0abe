"""How far the fp64 device path (precision = 64, k64_step) ends from the CPU restatement after hundreds of chaotic steps: the figures
behind tests/test_gpu_parity.py::test_fp64_path_follows_the_oracle_over_long_trajectories (bound there: 2e-5 A, the grain of the fp32
read-back at |x| ~ 50 A).      python tools/f64_deviation.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, make_stages, pipeline
from oracle import oracle as O
from tests.util import load_if, oracle_fire_from, oracle_model_from

s = Solver(0)
s.set_option("precision", 64)
for cid, (a, b, c, d) in (("chr21_1mb", (60, 250, 120, 150)), ("chr20_1mb", (40, 150, 60, 80)), ("chr13_1mb", (100, 600, 300, 300))):
    stages = [(2, a, 0.0, 1.0, 20.0, 0.5, 0.0), (0, b, 0.003, 0.4, 0.003, 0.9, 2000.0), (1, c, 0.005, 1.0, 0.05, 1.0, 1500.0), (2, d, 0.0, 1.0, 1.0, 0.85, 0.0)]
    IF = load_if(cid); m = default_model(); fire = default_fire()
    s.set_model(m); d10 = pipeline.IF2dist_new(s, IF)
    s.set_schedule(make_stages(stages), fire)
    s.init_replicas(2, 82364, 0)
    x0 = s.coords()
    s.run_steps(10 ** 6)
    x, v = s.coords(), s.velocities()
    om, of = oracle_model_from(m, IF.shape[0]), oracle_fire_from(fire)
    for r in range(2):
        xo, vo, ev = O.run_schedule(om, d10, O.make_stages(stages), of, 82364, r, x0=x0[r].astype(np.float64))
        xc = x[r].astype(np.float64); xc -= xc.mean(0)
        print(f"{cid} replica {r}: {ev} steps, max |x - x_oracle| = {np.abs(xc - xo).max():.3e} A, max |v - v_oracle| = {np.abs(v[r] - vo).max():.3e}, |x| max {np.abs(xo).max():.1f}")
