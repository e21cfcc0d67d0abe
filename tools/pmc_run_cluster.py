"""Fixed workload for rocprofv3 runs of the cluster kernel: chr1_500kb (or argv[2]), argv[1] replicas, 2000 MD + 2000 FIRE steps as
two multi-step launches (argv[3] = 0: the per-step path instead, eager; argv[4] = 0: option narrow_columns 0, round 2's column layout —
four columns per lane in every 256-column block, 8 column slots at N = 455 instead of 7 + 7 left-over columns)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cid = sys.argv[2] if len(sys.argv) > 2 else "chr1_500kb"
cluster = int(sys.argv[3]) if len(sys.argv) > 3 else 1
narrow = int(sys.argv[4]) if len(sys.argv) > 4 else 1
s = Solver(0)
s.set_option("narrow_columns", narrow)
s.set_model(default_model()); pipeline.IF2dist_new(s, load_if(cid))
s.set_schedule(make_stages([(1, 2000, 0.005, 1.0, 0.01, 1.0, 300.0), (2, 2000, 0.0, 1.0, 1.0, 0.85, 0.0)]))
s.set_option("resident", 1 if cluster else 0); s.set_option("cluster", cluster); s.set_option("use_graph", 0)
s.init_replicas(nrep, 82364, 0)
s.run_steps(2001)
print("MD   ms/steps/launches", s.last_timing(), "path", s.stat("last_path"), "us/step", 1e3 * s.last_timing()[0] / 2000)
s.run_steps(2000)
print("FIRE ms/steps/launches", s.last_timing(), "path", s.stat("last_path"), "us/step", 1e3 * s.last_timing()[0] / 2000)
print("kernel", s.step_kernel_name)
