#!/usr/bin/env python3
"""Cluster kernel (k_cluster) against the per-step path: bit-identity of whole anneals and microseconds per step.
    python tools/cluster_check.py [cid:nrep ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromosome3d_amd import Solver, default_model, default_schedule  # noqa: E402
from tests.util import load_if  # noqa: E402


def anneal(s, IF, nrep, resident, cluster, min_steps=300, gtol=1e-2):
    s.set_model(default_model())
    s.set_if_matrix(IF)
    s.set_schedule(default_schedule(min_steps), None, gtol, 100)
    s.set_option("resident", resident)
    s.set_option("cluster", cluster)
    s.init_replicas(nrep, 82364, 0)
    t0 = time.perf_counter()
    s.run()
    wall = time.perf_counter() - t0
    out = s.coords(), s.velocities(), s.energies(), s.last_timing(), wall, s.stat("last_path"), s.stat("resident_fallbacks")
    s.set_option("resident", -1)
    s.set_option("cluster", -1)
    return out


def main():
    cases = sys.argv[1:] or ["chr21_1mb:4", "chr21_1mb:20", "chr19_500kb:3", "chr1_500kb:3", "chr1_500kb:20", "chr4_1mb:20",
                             "chr13_1mb:9", "chr21_500kb:17"]
    s = Solver(0)
    for case in cases:
        cid, nrep = case.split(":")
        nrep = int(nrep)
        IF = load_if(cid)
        fb0 = s.stat("resident_fallbacks")
        xa, va, ea, ta, wa, pa, _ = anneal(s, IF, nrep, 0, 0)
        xb, vb, eb, tb, wb, pb, fb = anneal(s, IF, nrep, 1, 1)
        same = np.array_equal(xa, xb) and np.array_equal(va, vb) and np.array_equal(ea, eb)
        print(f"{cid:12s} n={IF.shape[0]:4d} nrep={nrep:3d}  per-step {1e3 * ta[0] / ta[1]:7.3f} us/step ({ta[2]} launches, path {pa:.0f})   "
              f"cluster {1e3 * tb[0] / tb[1]:7.3f} us/step ({tb[2]} launches, path {pb:.0f}, parts {s.stat('cluster_parts'):.0f} "
              f"cw {s.stat('cluster_compute_waves'):.0f}+{s.stat('cluster_helper_waves'):.0f} x{s.stat('cluster_wgs_per_cu'):.0f}/CU rpw {s.stat('cluster_rows_per_wave'):.0f}, fallbacks {fb - fb0:.0f})   steps {ta[1]}/{tb[1]}  bit-identical {same}", flush=True)
        if not same:
            d = np.abs(xa - xb)
            print("   max |dx| =", d.max(), " first differing replica", int(np.argmax(d.reshape(nrep, -1).max(1) > 0)))
    s.close()


if __name__ == "__main__":
    main()
