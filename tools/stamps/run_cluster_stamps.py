"""Diagnostic: time line of one step of the cluster kernel (libc3d_stamps.so): s_memrealtime stamps (100 MHz, 10 ns) of the
last 64 steps of a launch, H0 of (replica 0, part 0) and compute wave 0 of the same workgroup; medians in microseconds
since the step's start (H0 leaving barrier B1)."""
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
names = ["H0 tile units fetched, scalars done", "H0 past B2 (compute waves done)", "H0 tile sums done", "H0 rows updated and stored (B3 next)",
         "H0 tile units stored", "next step starts", "compute wave 0 starts", "compute wave 0 done"]
cases = [a.split(":") for a in sys.argv[1:]] or [("chr1_500kb", "20"), ("chr1_500kb", "8"), ("chr4_1mb", "20"), ("chr21_1mb", "20")]
for case in cases:
    cid, nrep = case[0], int(case[1])
    late = int(case[2]) if len(case) > 2 else 1       # cid:replicas[:0] = tile sums gathered with the rows (option cluster_late_tiles 0)
    try:
        IF = load_if(cid)
    except FileNotFoundError:                     # any of the 45 bundled matrices
        z = np.load(os.path.join(ROOT, "tests", "golden", "all45", f"{cid}_upper.npz"))
        n = int(z["n"]); IF = np.zeros((n, n)); iu = np.triu_indices(n); IF[iu] = z["upper"]; IF.T[iu] = z["upper"]
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    for kind, st in (("md", [(1, 4000, 0.005, 1.0, 0.01, 1.0, 300.0)]), ("fire", [(2, 4000, 0.0, 1.0, 1.0, 0.85, 0.0)])):
        s.set_schedule(make_stages(st)); s.set_option("resident", 1); s.set_option("cluster_late_tiles", late)
        s.init_replicas(nrep, 1, 0)
        s.run_steps(200)
        K = 2000
        s.run_steps(K)
        ms, _, _ = s.last_timing()
        buf = (C.c_ulonglong * 512)()
        L.c3d_debug_cluster_stamps(buf)
        a = np.array(buf[:], dtype=np.float64).reshape(64, 8)
        order = np.argsort(a[:, 0]); a = a[order]                      # by step start
        t0 = a[:-1, 0:1]
        rel = np.concatenate([a[:-1, 1:6] - t0, a[1:, 0:1] - t0, a[:-1, 6:8] - t0], axis=1) * 0.01     # us
        med = np.median(rel[2:-2], axis=0)
        print(f"{cid} {kind} nrep={nrep} late={s.stat('cluster_late_tiles'):.0f} parts={s.stat('cluster_parts'):.0f} cw={s.stat('cluster_compute_waves'):.0f} rpw={s.stat('cluster_rows_per_wave'):.0f}: "
              f"{1e3 * ms / K:.2f} us/step")
        for v, n_ in sorted(zip(med, names)):
            print(f"      {v:6.2f} us  {n_}")
