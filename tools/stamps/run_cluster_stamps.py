"""Diagnostic: cycles per phase of one workgroup of the cluster kernel (libc3d_stamps.so)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chromosome3d_amd.lib as lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "stamps", "libc3d_stamps.so")
from chromosome3d_amd import Solver, default_model, make_stages, pipeline
from tests.util import load_if
import numpy as np
s = Solver(0)
L = lib.load()
L.c3d_debug_cluster_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
names = ["lds-write+loop", "B1 wait", "H0 sums+scalars", "B2 wait (compute waves)", "H0 row update", "-", "H0 publish", "B3 + gather wait", "-", "sweeps"]
cases = [a.split(":") for a in sys.argv[1:]] or [("chr21_1mb", "20"), ("chr4_1mb", "20"), ("chr1_500kb", "3"), ("chr1_500kb", "20")]
for cid, nrep in cases:
    nrep = int(nrep)
    IF = load_if(cid)
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    for kind, st in (("md", [(1, 4000, 0.005, 1.0, 0.01, 1.0, 300.0)]), ("fire", [(2, 4000, 0.0, 1.0, 1.0, 0.85, 0.0)])):
        s.set_schedule(make_stages(st)); s.set_option("resident", 1); s.set_option("cluster", 1)
        s.init_replicas(nrep, 1, 0)
        s.run_steps(200)
        K = 2000
        s.run_steps(K)
        ms, _, _ = s.last_timing()
        buf = (C.c_ulonglong * 16)()
        L.c3d_debug_cluster_stamps(buf)
        a = np.array(buf[:10], dtype=np.float64) / K
        txt = " | ".join(f"{n} {v:.0f}" for n, v in zip(names[:8], a[:8]))
        print(f"{cid} {kind} nrep={nrep} parts={s.stat('cluster_parts'):.0f} rpw={s.stat('cluster_rows_per_wave'):.0f} path={s.stat('last_path'):.0f}: "
              f"{1e3 * ms / K:.2f} us/step; cycles/step: {txt} | total {a[:8].sum():.0f}; sweeps/step {a[9]:.2f}", flush=True)
