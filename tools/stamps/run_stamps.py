"""Diagnostic: per-phase cycle stamps of one workgroup of the step kernel (libc3d_stamps.so)."""
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
L.c3d_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong)]
for cid in ("chr21_1mb", "chr1_500kb"):
    IF = load_if(cid)
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    for kind, st in (("md", [(1, 400, 0.005, 1.0, 0.01, 1.0, 300.0)]), ("fire", [(2, 400, 0.0, 1.0, 1.0, 0.85, 0.0)])):
        for rpw in (2,):
            for nrep in (1, 20):
                s.set_schedule(make_stages(st)); s.set_option("rows_per_wave", rpw); s.set_option("use_graph", 0); s.set_option("replica_groups", 1)
                s.init_replicas(nrep, 1, 0)
                acc = []
                s.run_steps(50)
                for it in range(40):
                    s.run_steps(1)
                    buf = (C.c_ulonglong * 16)()
                    L.c3d_debug_stamps(buf)
                    t = np.array(buf[:7], dtype=np.int64)
                    acc.append(np.concatenate([np.diff(t[:6]), [t[0] - t[6]]]))
                a = np.median(np.array(acc), axis=0)
                print(f"{cid} {kind} rpw={rpw} nrep={nrep}: cycles loads-issue {a[0]:.0f} | scalars+stage {a[1]:.0f} | barrier {a[2]:.0f} | pair loop+reduce {a[3]:.0f} | epilogue {a[4]:.0f} | total {a[:5].sum():.0f} | kernarg wait before the first stamp {a[5]:.0f}", flush=True)
