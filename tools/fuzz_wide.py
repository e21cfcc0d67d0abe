"""Random large problems (n beyond the multi-step kernel's reach) through the per-step kernel's wide form and its narrow form: short MD + FIRE
trajectories must agree within rounding (another order of a row's sum), the narrow form with resident and with per-step pair constants bit for
bit.  python tools/fuzz_wide.py [seed] [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, make_stages, pipeline
from tests.util import synthetic_if

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
s = Solver(0)
t0 = time.time()
done = bad = 0
worst = 0.0
names = set()
while time.time() - t0 < budget:
    n = int(rng.choice([int(rng.integers(1025, 1300)), int(rng.integers(1300, 2700)), int(rng.choice([1025, 1032, 1033, 1040, 1041, 1279, 1280, 1281, 2047, 2048, 2049, 2559, 2560]))]))
    nrep = int(rng.integers(1, 7))
    steps = [(2, int(rng.integers(2, 8)), 0.0, 1.0, 20.0, 0.5, 0.0), (0, int(rng.integers(4, 14)), 0.003, 0.4, 0.003, 0.9, 2000.0),
             (1, int(rng.integers(2, 8)), 0.005, 1.0, 0.5, 0.9, 1000.0), (int(rng.choice([2, 5])), int(rng.integers(2, 10)), 0.0, 1.0, 1.0, 0.85, 0.0)]    # (kind 5: two-point steps)
    total = sum(st[1] for st in steps)
    IF, _ = synthetic_if(n, seed=int(rng.integers(1, 10 ** 6)))
    out = {}
    for wide, on in ((1, 1), (0, 1), (0, 0)):
        s.set_option("wide_tiles", wide); s.set_option("pair_targets", on); s.set_option("replica_groups", int(rng.integers(1, 4)))
        s.set_model(default_model())
        pipeline.IF2dist_new(s, IF)
        s.set_schedule(make_stages(steps), default_fire(), 0.0, 250)
        s.init_replicas(nrep, 82364, 0)
        assert s.run_steps(10 ** 6) == total
        names.add(s.step_kernel_name)
        out[wide, on] = (s.coords().copy(), s.velocities().copy())
    dx = float(np.abs(out[1, 1][0] - out[0, 1][0]).max())
    ok = np.isfinite(out[1, 1][0]).all() and dx < 2e-3 and np.array_equal(out[0, 1][0], out[0, 0][0]) and np.array_equal(out[0, 1][1], out[0, 0][1])
    worst = max(worst, dx)
    done += 1
    if not ok:
        bad += 1
        print(f"BAD n {n} nrep {nrep} steps {total}: wide - narrow {dx:.3g}", flush=True)
    if done % 10 == 0:
        print(f"... {done} problems, {bad} bad, largest wide - narrow coordinate difference {worst:.3g} A after {time.time() - t0:.0f} s", flush=True)
print(f"{done} random problems (n 1025..2699, 1-6 replicas, {seed=}), {bad} bad; largest wide - narrow coordinate difference {worst:.3g} A; kernels: {sorted(names)}")
