"""Per-XCD step budgets, measured by proxy — no kernel change.

chr1_500kb x 20 on one MI355X puts 3 replicas on XCDs 0-3 and 2 on XCDs 4-7: a multi-step launch is as slow as its heavy XCDs, 56 of
256 CUs idle.  The idea on paper: let the light XCDs advance more steps per launch and rotate which replicas are light.
The proxy runs the two halves as two CONTEXTS side by side, each with its own geometry (options cluster_xcd_count / cluster_xcd_base):
    A  12 replicas on XCDs 0-3 (3 per XCD, the heavy geometry)        K_h steps
    B   8 replicas on XCDs 4-7 (2 per XCD, the geometry the planner picks for two) K_l = K_h x step time A / step time B steps
started together on two host threads, and compares the replica-steps per second of the pair with the one context of 20 replicas.
That ratio is the CEILING of what step budgets could return (no rotation overhead, no extra launches).
    python tools/xcd_budget_proxy.py        (GPU box)"""
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
from tests.util import load_if

IF = load_if("chr1_500kb")
W = 205            # start inside the hot MD stages (as bench.py's regions)


def make(nrep, count=8, base=0):
    s = Solver(0)
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
    s.set_option("cluster_xcd_base", 0)
    s.set_option("cluster_xcd_count", count)
    s.set_option("cluster_xcd_base", base)
    s.init_replicas(nrep, 82364, 0)
    return s


def geometry(s):
    return f"{int(s.stat('cluster_parts'))} parts x {int(s.stat('cluster_compute_waves'))} compute waves x {int(s.stat('cluster_rows_per_wave'))} rows"


def alone(s, nrep, K):
    out = []
    for rep in range(4):
        s.init_replicas(nrep, 82364, 0)
        s.run_steps(W)
        s.run_steps(K)
        out.append(1e3 * s.last_timing()[0] / K)
    return statistics.median(out[1:])


def together(a, b, ka, kb):
    walls = []
    for rep in range(5):
        a.init_replicas(12, 82364, 0); b.init_replicas(8, 82364, 12)
        a.run_steps(W); b.run_steps(W)
        bar = threading.Barrier(3)
        done = [0.0, 0.0]

        def work(k, s, n):
            bar.wait()
            s.run_steps(n)
            done[k] = time.perf_counter()
        th = [threading.Thread(target=work, args=(0, a, ka)), threading.Thread(target=work, args=(1, b, kb))]
        for t in th:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        walls.append((max(done) - t0, done[0] - t0, done[1] - t0))
    return sorted(walls)[len(walls) // 2]


if __name__ == "__main__":
    K = 1900
    s20 = make(20)
    t20 = alone(s20, 20, K)
    # the bench's own figure for a whole fixed-length anneal in one call
    full = []
    for rep in range(3):
        s20.init_replicas(20, 82364, 0)
        s20.run_steps(10 ** 7)
        full.append(s20.last_timing()[0])
    L = s20.schedule_length
    print(f"one context, 20 replicas on 8 XCDs ({geometry(s20)}): {t20:.3f} us per step over steps {W}..{W + K}; whole schedule ({L} steps, one call) {statistics.median(full):.2f} ms "
          f"= {20 * L / statistics.median(full) / 1e3:.3f} M replica-steps/s")
    s20.close()
    a, b = make(12, 4, 0), make(8, 4, 4)
    ta, tb = alone(a, 12, K), alone(b, 8, K)
    print(f"A alone: 12 replicas on XCDs 0-3 ({geometry(a)}): {ta:.3f} us per step;   B alone: 8 replicas on XCDs 4-7 ({geometry(b)}): {tb:.3f} us per step;   ratio {ta / tb:.3f}")
    for kb in (K, int(round(K * ta / tb))):
        kb = min(kb, L - W)
        wall, wa, wb = together(a, b, K, kb)
        rate = (12 * K + 8 * kb) / wall / 1e6
        print(f"A {K} steps + B {kb} steps side by side: wall {1e3 * wall:.3f} ms (A done at {1e3 * wa:.3f}, B at {1e3 * wb:.3f}): {rate:.3f} M replica-steps/s against {20 / t20:.3f} M in one context "
              f"({rate / (20 / t20):.3f}x); abandoned launches A {int(a.stat('resident_fallbacks'))} B {int(b.stat('resident_fallbacks'))}")
    print(f"# ceiling for the whole anneal if light replicas could take the extra steps (no rotation, no extra launches): {statistics.median(full):.2f} ms x (20 / t20) / rate of the last line above")
    a.close(); b.close()
