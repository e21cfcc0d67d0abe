"""Small fixed workload for rocprofv3 --pmc runs: 20 replicas of chr1_500kb, 200 MD + 200 FIRE steps, eager."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = Solver(0)
s.set_model(default_model()); pipeline.IF2dist_new(s, load_if("chr1_500kb"))
s.set_schedule(make_stages([(1, 200, 0.005, 1.0, 0.01, 1.0, 300.0), (2, 200, 0.0, 1.0, 1.0, 0.85, 0.0)]))
s.set_option("use_graph", 0)
s.init_replicas(nrep, 82364, 0)
s.run_steps(10**6)
print("ms/steps/launches", s.last_timing())
