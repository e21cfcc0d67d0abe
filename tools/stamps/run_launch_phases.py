"""Diagnostic (libc3d_stamps.so): where the fixed cost of one cluster launch goes — s_memrealtime stamps (100 MHz) of the stamping
workgroup at kernel entry, end of the prologue, start of the first step, end of the last step, against the kernel's own
start/stop events.    python tools/stamps/run_launch_phases.py [steps per launch = 20]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chromosome3d_amd.lib as lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "stamps", "libc3d_stamps.so")
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
import numpy as np
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
s = Solver(0)
L = lib.load()
L.c3d_debug_cluster_pstamps.argtypes = [C.POINTER(C.c_ulonglong)]
s.set_model(default_model()); pipeline.IF2dist_new(s, load_if("chr1_500kb"))
s.set_schedule(make_stages([(1, 100000, 0.005, 1.0, 0.01, 1.0, 300.0)])); s.set_option("resident", 1); s.set_option("kernel_timing", 1)
s.init_replicas(20, 82364, 0)
s.run_steps(200)
rows = []
for _ in range(40):
    s.run_steps(K)
    buf = (C.c_ulonglong * 8)()
    L.c3d_debug_cluster_pstamps(buf)
    t = [0.01 * (buf[k] - buf[0]) for k in range(8)]
    rows.append(t + [s.stat("last_kernel_us")])
m = np.median(np.array(rows), axis=0)
print(f"{K} steps per launch, chr1_500kb x 20, medians of 40 launches (us since the stamping workgroup's first instruction):")
print(f"  H0: slot known           {m[6]:7.2f}")
print(f"  compute wave 0: targets in registers, pair constants in LDS  {m[4]:7.2f}")
print(f"  compute wave 0: its share of the coordinates in LDS          {m[5]:7.2f}")
print(f"  compute wave 0 at the first barrier                          {m[7]:7.2f}")
print(f"  H0: prologue done        {m[1]:7.2f}")
print(f"  first step starts        {m[2]:7.2f}")
print(f"  last step finished       {m[3]:7.2f}   -> {(m[3] - m[2]) / K:.3f} us per step")
print(f"  kernel start-stop events {m[8]:7.2f}   -> {m[8] - m[3]:.2f} us outside the workgroup's own time (dispatch before, drain after)")
