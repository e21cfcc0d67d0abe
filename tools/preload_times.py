"""Time to load each translation unit's code object (one kernel of it touched through hipFuncGetAttributes), in the order given:
python tools/preload_times.py cluster,device,score    (GPU box)"""
import ctypes, time, sys, os
L=ctypes.CDLL("/root/repo/chromosome3d_amd/_lib/libc3d.so")
L.c3d_set_process_option(b"preload", ctypes.c_double(0.0))
ctx=ctypes.c_void_p()
t=time.perf_counter(); rc=L.c3d_create(0, ctypes.byref(ctx)); print("create", rc, (time.perf_counter()-t)*1e3)
order=sys.argv[1].split(",")
names={"score":"_ZN3c3d18preload_score_unitEv","device":"_ZN3c3d19preload_device_unitEv","cluster":"_ZN3c3d20preload_cluster_unitEv"}
for o in order:
    f=getattr(L,names[o]); t=time.perf_counter(); rc=f(); print(o, rc, round((time.perf_counter()-t)*1e3,2),"ms")
