"""Soak test of the in-launch hand-off: every bundled matrix the resident kernel accepts is annealed on both
paths and the results are compared bit for bit — idle, and again while a second context keeps the GPU busy
with per-step launches of another chromosome (uneven load; a stale or torn record would change the bits).

    python tools/resident_soak.py [rounds=2] [replicas=20]
Needs tests/golden/_all (tools/pack_all_inputs.py).
"""
import glob, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline

ALL = os.path.join(ROOT, "tests", "golden", "_all")
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 20


def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m


def anneal(s, IF, resident, first):
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(1000), default_fire(), 1e-2, 250)
    s.set_option("resident", resident)
    s.init_replicas(nrep, 82364, first)
    s.run()
    return s.coords(), s.last_timing()


stop = False


def background():
    b = Solver(0)
    IF = load("chr5_500kb")
    b.set_model(default_model()); pipeline.IF2dist_new(b, IF)
    b.set_schedule(default_schedule(1000), default_fire(), 0.0, 250)
    b.set_option("resident", 0)
    while not stop:
        b.init_replicas(12, 1, 0)
        b.run()


s = Solver(0)
cids = sorted({os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_upper.npz")})
ref = {}
t0 = time.time()
checked = mism = fallbacks = 0
for cid in cids:
    IF = load(cid)
    ref[cid] = anneal(s, IF, 0, 0)[0]
for phase in ("idle", "loaded"):
    th = None
    if phase == "loaded":
        th = threading.Thread(target=background); th.start()
    for r in range(rounds):
        for cid in cids:
            IF = load(cid)
            x, (ms, steps, launches) = anneal(s, IF, 1, 0)
            if launches > 100:
                fallbacks += 1          # too large for the resident kernel, or it fell back
                print(f"  {phase} round {r}: {cid} (N={IF.shape[0]}) ran step by step", flush=True)
            checked += 1
            if not np.array_equal(x, ref[cid]):
                mism += 1
                print(f"MISMATCH {phase} round {r} {cid}: max |dx| = {np.abs(x - ref[cid]).max():.3e}", flush=True)
    if th:
        stop = True; th.join()
    print(f"{phase}: {checked} anneals compared so far, {mism} mismatches, {fallbacks} ran step by step", flush=True)
print(f"{len(cids)} matrices x {rounds} rounds x 2 phases, {nrep} replicas: {mism} mismatches; {time.time() - t0:.1f} s")
sys.exit(1 if mism else 0)
