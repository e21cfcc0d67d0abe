"""Diagnostic: do G independent contexts (own streams) overlap on one GPU?  20 replicas split G ways."""
import sys, os, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import load_if
IF = load_if("chr1_500kb")
for G in (1, 2, 4, 5, 10):
    per = 20 // G
    ss = []
    for g in range(G):
        s = Solver(0); s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
        s.set_schedule(default_schedule(3000)); s.set_option("rows_per_wave", 2)
        s.init_replicas(per, 82364, g * per); s.run_steps(10**6)          # prime graphs
        s.init_replicas(per, 82364, g * per)
        ss.append(s)
    L = ss[0].schedule_length
    bar = threading.Barrier(G + 1)
    def work(s):
        bar.wait(); s.run_steps(L); bar.wait()
    th = [threading.Thread(target=work, args=(s,)) for s in ss]
    [t.start() for t in th]
    bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
    [t.join() for t in th]
    print(f"G={G:2d} streams x {per:2d} replicas: {1e6*dt/L:7.3f} us per step of all 20 replicas, {20*L/dt/1e6:6.3f} M replica-steps/s, per-ctx device ms {ss[0].last_timing()[0]:.1f}", flush=True)
    [s.close() for s in ss]
