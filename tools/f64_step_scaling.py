"""A/B of the fp64 step's rows per wave (option f64_rows: 2 or 3), replica groups and replica counts on a bundled matrix: the bench's
fp64 leg (W warm-up steps, then 200 steps, median of 7 regions).    python tools/f64_rows_ab.py [name]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_fire, default_schedule, pipeline
from tests.util import load_if

name = sys.argv[1] if len(sys.argv) > 1 else "chr1_500kb"
IF = load_if(name)
for M in (5, 20, 40):
    for groups in (1, 2, 3):
        for rows in (2, 3):
            s = Solver(0)
            s.set_option("precision", 64)
            s.set_option("f64_rows", rows)
            s.set_option("replica_groups", groups)
            s.set_model(default_model())
            pipeline.IF2dist_new(s, IF)
            s.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
            warm = 400
            regs = []
            for rep in range(8):
                s.init_replicas(M, 82364, 0)
                s.run_steps(warm)
                t0 = time.perf_counter()
                did = s.run_steps(200)
                wall = time.perf_counter() - t0
                if rep >= 3:
                    regs.append((wall, s.last_timing()[0], did))
            wall, dev_ms, did = sorted(regs)[len(regs) // 2]
            print(f"{name} x {M:2d}  groups {groups}  f64_rows {rows}: {1e6 * wall / did:7.3f} us per step (wall), {1e3 * dev_ms / did:7.3f} (device), "
                  f"{M * did / wall / 1e6:.3f} M replica-steps/s", flush=True)
            s.close()
