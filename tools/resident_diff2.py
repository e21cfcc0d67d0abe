"""Debug aid: mix resident / per-step launches over the first three steps and compare velocities."""
import sys

import numpy as np

sys.path.insert(0, ".")
from chromosome3d_amd import Solver, default_model, default_schedule  # noqa: E402
from tests.util import load_if  # noqa: E402


def run(cid, plan):
    s = Solver(0)
    s.set_model(default_model())
    s.set_if_matrix(load_if(cid))
    s.set_schedule(default_schedule(300), None, 0.0, 250)
    s.set_option("resident_min_ops", 1)
    s.init_replicas(2, 82364, 0)
    for mode, k in plan:
        s.set_option("resident", 1 if mode == "R" else 0)
        s.run_steps(k)
    return s.coords(), s.velocities()


def main():
    cid = sys.argv[1] if len(sys.argv) > 1 else "chr1_500kb"
    ref = run(cid, [("S", 1), ("S", 1), ("S", 1)])
    for name, plan in (("S2+R1", [("S", 2), ("R", 1)]), ("R1+R1+S1", [("R", 1), ("R", 1), ("S", 1)]), ("R2+S1", [("R", 2), ("S", 1)]),
                       ("R1+R1+R1", [("R", 1), ("R", 1), ("R", 1)]), ("R1+R2", [("R", 1), ("R", 2)]), ("R3", [("R", 3)]), ("S1+R2", [("S", 1), ("R", 2)])):
        x, v = run(cid, plan)
        print(f"{name:10s} max|dx|={np.abs(x - ref[0]).max():.3e} max|dv|={np.abs(v - ref[1]).max():.3e} ndiff_v={(v != ref[1]).sum()}", flush=True)


if __name__ == "__main__":
    main()


def detail():
    cid = "chr1_500kb"
    ref = run(cid, [("S", 2), ("S", 1)])
    x, v = run(cid, [("S", 2), ("R", 1)])
    bad = np.argwhere(v != ref[1])
    print("differing (rep, bead, comp):")
    for r, i, c in bad[:40]:
        print(f"  rep {r} bead {i:3d} (tile {i // 8}, row-in-tile {i % 8}) comp {c}: {ref[1][r, i, c]!r} vs {v[r, i, c]!r}")
    beads = sorted(set((int(r), int(i)) for r, i, c in bad))
    print("beads:", len(beads), "row-in-tile histogram:", np.bincount([i % 8 for _, i in beads], minlength=8).tolist())


if len(sys.argv) > 2 and sys.argv[2] == "detail":
    detail()
