"""Step rate of the other BASELINE.json configs on one GPU (parity-test cases, not bench lines):
config 2 chr21_1mb x20, config 3 chr1_500kb x20, config 5 synthetic N=2500 x8 (HBM-roofline stress)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import load_if, synthetic_if
s = Solver(0)
for name, IF, nrep, nmin in (("config2 chr21_1mb", load_if("chr21_1mb"), 20, 3000), ("config3 chr1_500kb", load_if("chr1_500kb"), 20, 3000),
                             ("config5 synthetic N=2500", synthetic_if(2500)[0], 8, 1000)):
    n = IF.shape[0]
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(nmin), None, 0.0, 250)
    s.init_replicas(nrep, 82364, 0); s.run_steps(10 ** 7)            # graphs
    s.init_replicas(nrep, 82364, 0); s.run_steps(10 ** 7)
    ms, steps, la = s.last_timing()
    R = s.num_restraints
    B = 4 * R + 72 * n
    print(f"{name}: N={n} R={R} replicas={nrep}: {1e3 * ms / steps:.2f} us/step ({la} launches), {nrep * steps / ms * 1e3 / 1e6:.3f} M replica-steps/s, "
          f"full schedule {ms:.1f} ms, algorithmic {nrep * steps * B / (ms * 1e-3) / 1e9:.0f} GB/s ({nrep * steps * B / (ms * 1e-3) / 8e12:.3f} of 8 TB/s), "
          f"pair rate {nrep * steps * n * n / (ms * 1e-3) / 1e12:.2f} Tpair/s", flush=True)
