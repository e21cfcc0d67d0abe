"""The fp64 step (option precision = 64) against the replica count and the number of replica groups (streams) on a bundled matrix:
regions of 200 hot-MD steps after 400 warm-up steps, median of 5.  Shows how much of a step is launch boundary + one wave's critical
path (the 5-replica rows) and how much is throughput (the 20 -> 40 difference); profiles/r04_f64_step_rows_and_helper_experiments.txt
was made with this loop over two experimental builds of k64_step (three rows per wave; a helper wave), both reverted.
    python tools/f64_step_scaling.py [name]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_fire, default_schedule, pipeline
from tests.util import load_if

name = sys.argv[1] if len(sys.argv) > 1 else "chr1_500kb"
IF = load_if(name)
for M in (5, 10, 20, 40):
    for groups in (1, 2, 3):
        s = Solver(0)
        s.set_option("precision", 64)
        s.set_option("replica_groups", groups)
        s.set_model(default_model())
        pipeline.IF2dist_new(s, IF)
        s.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
        regs = []
        for rep in range(8):
            s.init_replicas(M, 82364, 0)
            s.run_steps(400)
            t0 = time.perf_counter()
            did = s.run_steps(200)
            wall = time.perf_counter() - t0
            if rep >= 3:
                regs.append((wall, s.last_timing()[0], did))
        wall, dev_ms, did = sorted(regs)[len(regs) // 2]
        print(f"{name} x {M:2d}  groups {groups}: {1e6 * wall / did:7.3f} us per step (wall), {1e3 * dev_ms / did:7.3f} (device), "
              f"{M * did / wall / 1e6:.3f} M replica-steps/s", flush=True)
        s.close()
