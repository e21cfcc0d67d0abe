import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import synthetic_if
IF = synthetic_if(2500)[0]
s = Solver(0)
s.set_option("symmetric", int(sys.argv[1]) if len(sys.argv) > 1 else 1); s.set_option("use_graph", 0)
s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
s.set_schedule(default_schedule(100), None, 0.0, 250)
s.init_replicas(8, 82364, 0); s.run_steps(600)
print(s.last_timing(), s.step_kernel_name)
