"""Config 5 (synthetic N = 2500 x 8, the one HBM-streaming configuration) with the device's clocks sampled WHILE it steps (rocm-smi from a second
thread): its step time has moved between boxes (25.2-33 us) and a slow box should be told from a fast one by its clocks and its HBM counters
(profiles/r06_pmc_hbm_traffic_config5_k_step.txt).    python tools/config5_clocks.py        (GPU box)"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import synthetic_if

IF = synthetic_if(2500)[0]
s = Solver(0)
s.set_model(default_model())
pipeline.IF2dist_new(s, IF)
s.set_schedule(default_schedule(1000), None, 0.0, 250)
samples = []
stop = False


def sample():
    while not stop:
        t = time.perf_counter()
        p = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        keep = [l.split(":", 1)[-1].strip() for l in p.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "Power")) and "GPU[" in l]
        samples.append((t, " | ".join(keep)))
        time.sleep(0.2)


th = threading.Thread(target=sample)
th.start()
per = []
t0 = time.perf_counter()
for rep in range(12):
    s.init_replicas(8, 82364, 0)
    s.run_steps(10 ** 7)
    ms, steps, la = s.last_timing()
    per.append((time.perf_counter() - t0, 1e3 * ms / steps))
stop = True
th.join()
print("anneals of N = 2500 x 8 (3172 steps each), us per SA step by anneal: " + " ".join(f"{u:.2f}" for _, u in per))
print(f"kernel: {s.step_kernel_name}")
for t, line in samples:
    print(f"  t = {t - t0:6.2f} s  {line}")
