"""Diagnostic (libc3d_stamps.so): pair sums and chain sums of every row at the last evaluation, k_step against k_cluster."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import chromosome3d_amd.lib as lib
lib.LIB_PATH = os.path.join(ROOT, "tools", "stamps", "libc3d_stamps.so")
from chromosome3d_amd import Solver, default_model, make_stages
from tests.util import load_if
import numpy as np
L = lib.load()
L.c3d_debug_forces.argtypes = [C.POINTER(C.c_float), C.c_int]
cid = sys.argv[1] if len(sys.argv) > 1 else "chr19_500kb"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
IF = load_if(cid); n = IF.shape[0]
out = {}
for name, res in (("step", 0), ("cluster", 1)):
    s = Solver(0)
    s.set_model(default_model()); s.set_if_matrix(IF)
    s.set_schedule(make_stages([(0, 400, 0.003, 0.4, 0.003, 0.9, 2000.0)]), None, 0.0, 250)
    s.set_option("resident", res); s.set_option("resident_min_ops", 1); s.set_option("use_graph", 0); s.set_option("replica_groups", 1)
    s.init_replicas(1, 82364, 0)
    s.run_steps(k)
    buf = (C.c_float * (2 * 6 * 1024))()
    L.c3d_debug_forces(buf, res)
    out[name] = np.array(buf[:], dtype=np.float32).reshape(2, 6, 1024)[res, :, :n].copy()
    s.close()
a, b = out["step"], out["cluster"]
names = ["pair x", "pair y", "pair z", "chain x", "chain y", "chain z"]
for c in range(6):
    bad = np.nonzero(a[c] != b[c])[0]
    print(names[c], "rows differing:", bad[:30].tolist(), "of", len(bad))
    for r in bad[:4]:
        print("   row", r, "step", repr(a[c][r]), "cluster", repr(b[c][r]))
