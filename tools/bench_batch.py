#!/usr/bin/env python3
"""Config 4 through the native batch executor: writes the 22 bundled 500 kb matrices as text files into a scratch directory and
runs c3d_batch over them with 1-6 lanes (host threads + contexts) per GPU, small anneals paired or not, the queue largest-first or warm-started.
    python tools/bench_batch.py [pattern=_500kb]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ALL = os.path.join(ROOT, "tests", "golden", "all45")
pattern = sys.argv[1] if len(sys.argv) > 1 else "_500kb"
with tempfile.TemporaryDirectory() as td:
    ind = os.path.join(td, "in")
    os.mkdir(ind)
    t0 = time.perf_counter()
    for f in sorted(os.listdir(ALL)):
        if not f.endswith("_upper.npz") or pattern not in f:
            continue
        z = np.load(os.path.join(ALL, f))
        n = int(z["n"])
        m = np.zeros((n, n))
        iu = np.triu_indices(n)
        m[iu] = z["upper"]
        m.T[iu] = z["upper"]
        with open(os.path.join(ind, f.replace("_upper.npz", "_matrix.txt")), "w") as out:
            for row in m:
                out.write(" ".join(repr(float(v)) for v in row) + " \r\n")
    print(f"wrote {len(os.listdir(ind))} matrices in {time.perf_counter() - t0:.1f} s")
    runs = [(1, 0, "lpt"), (3, 0, "lpt")] + [(lanes, 1, order) for _ in range(3) for lanes in (3, 4, 6) for order in ("lpt", "warm")]
    for k, (lanes, pair, order) in enumerate(runs):
        out = os.path.join(td, f"out{k}")
        t0 = time.perf_counter()
        p = subprocess.run([os.path.join(ROOT, "chromosome3d_amd", "_lib", "c3d_batch"), ind, "--out", out, "--lanes", str(lanes), "--pair", str(pair), "--order", order],
                           capture_output=True, text=True)
        wall = time.perf_counter() - t0
        last = p.stdout.strip().splitlines()[-1] if p.stdout.strip() else p.stderr[-300:]
        halves = sum(" XCDs " in l for l in p.stdout.splitlines())
        print(f"lanes {lanes} pair {pair} order {order}: process wall {wall:.2f} s (incl. device init); {halves} anneals on half a device; {last}")
