"""Random synthetic problems (tests.util.synthetic_if: n = 30 .. 760 beads, 4-20 replicas, random seeds), whole default anneals with the
product's exit test, the final stage as FIRE (final_minimiser 0) and as shipped (two-point steps, then FIRE): anneals that used up the
stage, steps, and how far the two results are apart (lowest and median total energy, best-energy Spearman).
    python tools/final_minimiser_fuzz.py [seed=1] [seconds=60]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
from tests.util import synthetic_if

if __name__ == "__main__":
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    s = Solver(0); t0 = time.time(); it = 0
    steps = np.zeros(2); used_up = [0, 0]; de_min, de_med, drho = [], [], []
    while time.time() - t0 < budget:
        n = int(rng.integers(30, 761)); nrep = int(rng.integers(4, 21)); seed = int(rng.integers(1, 1 << 30))
        IF, _ = synthetic_if(n, seed=int(rng.integers(1, 1 << 30)))
        res = []
        for fm in (0, 1):
            s.set_model(default_model()); pipeline.IF2dist_new(s, IF); s.set_option("final_minimiser", fm)
            s.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250); s.init_replicas(nrep, seed, 0); s.run()
            st = s.last_timing()[1]; e = s.energies(); x = s.coords()
            steps[fm] += st - 2172; used_up[fm] += st >= 5172
            res.append((e.sum(axis=1), -pipeline.spearman_IF_pdb(IF, x[int(np.argmin(e[:, 0]))])))
        de_min.append((res[1][0].min() - res[0][0].min()) / res[0][0].min()); de_med.append((np.median(res[1][0]) - np.median(res[0][0])) / np.median(res[0][0]))
        drho.append(res[1][1] - res[0][1]); it += 1
    s.set_option("final_minimiser", 1)
    de_min, de_med, drho = np.array(de_min), np.array(de_med), np.array(drho)
    print(f"{it} random problems: final-stage steps FIRE {steps[0]:.0f} / shipped {steps[1]:.0f} ({steps[1] / steps[0]:.2f}); anneals that used up the 3000 steps: FIRE {used_up[0]}, shipped {used_up[1]}; "
          f"shipped - FIRE, relative: lowest total energy mean {de_min.mean():+.1e} (|max| {np.abs(de_min).max():.1e}), median total energy mean {de_med.mean():+.1e} (|max| {np.abs(de_med).max():.1e}); "
          f"Spearman of the best-energy model: mean {drho.mean():+.5f}, |max| {np.abs(drho).max():.4f}, within 1e-3: {(np.abs(drho) < 1e-3).sum()}")
