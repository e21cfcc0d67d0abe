"""Debug aid: first step at which the multi-step cluster kernel and the per-step kernel disagree."""
import sys

import numpy as np

sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, default_schedule  # noqa: E402
from tests.util import load_if  # noqa: E402


def state(cid, nrep, k, resident):
    s = Solver(0)
    s.set_model(default_model(**{kv.split("=")[0][2:]: float(kv.split("=")[1]) for kv in sys.argv[3:] if kv.startswith("m:")}))
    s.set_if_matrix(load_if(cid))
    s.set_schedule(default_schedule(300), None, 0.0, 250)
    s.set_option("resident", resident)
    s.set_option("resident_min_ops", 1)
    for kv in sys.argv[3:]:
        if kv.startswith("m:"):
            continue
        key, val = kv.split("=")
        s.set_option(key, float(val))
    s.init_replicas(nrep, 82364, 0)
    s.run_steps(k)
    return s.coords(), s.velocities()


def main():
    cid = sys.argv[1] if len(sys.argv) > 1 else "chr21_1mb"
    nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    for k in (1, 2, 3, 4, 5, 6, 8, 16, 100, 200, 201, 202, 210, 400, 1200, 1212, 1300, 2172, 2180):
        xa, va = state(cid, nrep, k, 0)
        xb, vb = state(cid, nrep, k, 1)
        dx = np.abs(xa - xb).max()
        dv = np.abs(va - vb).max()
        print(f"{cid} k={k:5d} max|dx|={dx:.3e} max|dv|={dv:.3e} nan={np.isnan(xb).any()}", flush=True)
        if dx > 0 or dv > 0:
            bad = np.argwhere(np.abs(xa - xb) > 0)
            print("  first differing x (replica, bead, comp):", bad[:5].tolist(), "of", len(bad))
            badv = np.argwhere(np.abs(va - vb) > 0)
            print("  first differing v (replica, bead, comp):", badv[:8].tolist(), "of", len(badv), "of", va.size)
            print("  beads:", sorted(set(int(b[1]) for b in badv))[:40])
            break


if __name__ == "__main__":
    main()
