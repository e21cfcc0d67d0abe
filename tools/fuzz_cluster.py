"""Fuzz of the multi-step cluster kernel against the per-step kernel: random bead counts (20..760), replica counts (1..24), short
four-stage schedules (the last one FIRE or two-point steps with a random hand-over to FIRE) run in randomly sized c3d_run_steps calls, either
hand-off form of the tile sums; both launch forms must end
in the same bits with no abandoned or incomplete launch.      python tools/fuzz_cluster.py [seed = 1] [seconds = 60] [xcd]
("xcd": the multi-step side on a random XCD set per problem, moved between launches — round 5)
(On an MI355X: 4 minutes, 16 463 problems through 36 instantiations of k_cluster, 0 differences; 8 minutes with the single-workgroup
sizes included, 32 366 problems, 0 differences.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import synthetic_if


def fuzz(s, seed=1, seconds=60.0, out=print, fallbacks_are_bad=True, xcd_sets=False, xcd_fixed=None, nmax=760, repmax=24):
    """fallbacks_are_bad=False: another context shares the GPU, abandoned launches (re-run step by step) are expected; only the bits count.
    xcd_sets=True (round 5): the multi-step side of every problem runs on a random XCD set (cluster_xcd_count 1..8, cluster_xcd_base anywhere it
    fits, the base changed again in the middle of the run): the per-step path is what it must equal, wherever a geometry exists for the set.
    xcd_fixed=(count, base): every multi-step launch on that set (two contexts on disjoint halves: tests)."""
    rng = np.random.default_rng(seed)
    t0 = time.time(); it = 0; bad = 0; kernels = {}; t_note = t0
    while time.time() - t0 < seconds:
        n = int(rng.integers(20, nmax + 1)); nrep = int(rng.integers(1, repmax + 1)); late = int(rng.integers(0, 2))
        IF, _ = synthetic_if(n, seed=int(rng.integers(1, 1 << 30)))
        k = [int(rng.integers(3, 40)) for _ in range(4)]
        last = int(rng.choice([2, 5])); hand_over = int(rng.integers(2, 45))       # (kind 5: two-point steps, FIRE after `hand_over` of them — round 5)
        stages = [(2, k[0], 0.0, 1.0, 20.0, 0.5, 0.0), (0, k[1], 0.003, 0.4, 0.003, 0.9, 2000.0), (1, k[2], 0.005, 1.0, 0.05, 1.0, 1500.0),
                  (last, k[3], 0.0, 1.0, 1.0, 0.85, 0.0)]
        res = []
        fb0, inc0 = s.stat("resident_fallbacks"), s.stat("cluster_incomplete")      # counters since c3d_create
        for resident in (0, 1):
            s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
            s.set_option("final_minimiser_steps", hand_over)
            s.set_schedule(make_stages(stages)); s.set_option("resident", resident); s.set_option("cluster_late_tiles", late)
            count = xcd_fixed[0] if xcd_fixed else (int(rng.integers(1, 9)) if (xcd_sets and resident) else 8)
            s.set_option("cluster_xcd_base", 0); s.set_option("cluster_xcd_count", count)
            s.set_option("cluster_xcd_base", xcd_fixed[1] if xcd_fixed else int(rng.integers(0, 9 - count)))
            s.init_replicas(nrep, 82364 + it, 0)
            used = set()
            while True:                                 # odd chunking of the range: launches of different lengths
                if s.run_steps(int(rng.integers(1, 50))) == 0:
                    break
                used.add(s.step_kernel_name)
                if xcd_sets and resident and rng.integers(0, 4) == 0:
                    s.set_option("cluster_xcd_base", int(rng.integers(0, 9 - count)))       # the set may move between launches
            res.append((s.coords(), s.velocities(), used, s.stat("resident_fallbacks") - fb0, s.stat("cluster_incomplete") - inc0))
        same = np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        for name in res[1][2]:
            kernels[name] = kernels.get(name, 0) + 1
        if not same or (fallbacks_are_bad and (res[1][3] or res[1][4])):
            bad += 1
            out(f"{'MISMATCH' if not same else 'FALLBACK'} n={n} replicas={nrep} late={late} stages={k} last kind {last} hand-over {hand_over} {sorted(res[1][2])} fallbacks={res[1][3]} incomplete={res[1][4]}")
        it += 1
        if time.time() - t_note > 30.0:            # a sign of life for long runs
            t_note = time.time(); out(f"... {it} problems, {bad} bad after {t_note - t0:.0f} s")
    s.set_option("resident", -1); s.set_option("cluster_late_tiles", 1)
    s.set_option("final_minimiser_steps", 1000)
    s.set_option("cluster_xcd_base", 0); s.set_option("cluster_xcd_count", 8)
    return it, bad, kernels


if __name__ == "__main__":
    it, bad, kernels = fuzz(Solver(0), int(sys.argv[1]) if len(sys.argv) > 1 else 1, float(sys.argv[2]) if len(sys.argv) > 2 else 60.0,
                            xcd_sets=len(sys.argv) > 3 and sys.argv[3] == "xcd")
    print(f"{it} random problems, {bad} bad; kernels used: {dict(sorted(kernels.items()))}")
