"""What c3d_create costs with the code objects loaded inside it (round 6: csrc/c3d_api.cpp "code objects"), one fresh process per mode:
python tools/preload_times.py            (GPU box)
  preload 0   no unit in c3d_create (each at the first entry that needs it)
  preload 1   the default job's four units (default)
  preload 2   all sixteen
Prints the time of the runtime's own start (c3d_device_count: hipInit), of c3d_create after it, and of a second c3d_create."""
import ctypes
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if len(sys.argv) > 1 and sys.argv[1] == "child":
    mode = float(sys.argv[2])
    L = ctypes.CDLL(os.path.join(ROOT, "chromosome3d_amd", "_lib", "libc3d.so"))
    L.c3d_set_process_option.argtypes = [ctypes.c_char_p, ctypes.c_double]
    L.c3d_get_stat.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
    assert L.c3d_set_process_option(b"preload", mode) == 0
    t = time.perf_counter()
    n = L.c3d_device_count()
    t_init = (time.perf_counter() - t) * 1e3
    out = []
    ctxs = []
    for k in range(2):
        ctx = ctypes.c_void_p()
        t = time.perf_counter()
        rc = L.c3d_create(0, ctypes.byref(ctx))
        out.append((time.perf_counter() - t) * 1e3)
        assert rc == 0
        ctxs.append(ctx)
    v = ctypes.c_double()
    L.c3d_get_stat(ctxs[0], b"units_loaded", ctypes.byref(v))
    print(f"preload {int(mode)}: devices {n}, runtime start {t_init:7.1f} ms, first c3d_create {out[0]:6.2f} ms, second {out[1]:5.2f} ms, units loaded {int(v.value)}")
    sys.exit(0)

for rep in range(2):
    for mode in (0, 1, 2):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(mode)], check=True)
