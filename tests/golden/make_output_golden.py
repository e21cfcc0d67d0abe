#!/usr/bin/env python3
"""Output-side goldens (build container only): write solver-output PDBs with OUR writer (c3d_write_pdb, host code of
libc3d.so; no GPU needed) from the coordinates of bundled reference models, then let the reference's own assess_dgsa
subs post-process them (tests/golden/make_output_golden.pl evals them from /root/reference at run time).

    python tests/golden/make_output_golden.py [/root/reference]

Committed results (tests/golden/output_side/): <id>_solver_out.pdb (input = our writer's bytes), <id>_final.pdb and
<id>_model_info.log (what the reference leaves behind), output_side_golden.json."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from chromosome3d_amd import pipeline  # noqa: E402
from tests.util import load_pdb_xyz, model_pdb  # noqa: E402

ENERGIES = {"chr21_1mb": (54337.7461, 1203.25, 3.0625), "chr22_1mb": (39387.6953, 877.5, 0.125)}   # noe, bond, repel: arbitrary but fixed


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    out = os.path.join(HERE, "output_side")
    os.makedirs(out, exist_ok=True)
    for cid, (e_noe, e_bond, e_rep) in ENERGIES.items():
        xyz = load_pdb_xyz(model_pdb(cid)).astype("float32")
        pipeline.write_pdb(os.path.join(out, f"{cid}_solver_out.pdb"), xyz, e_noe, e_bond, e_rep, title=f"{cid}_1.pdb")
    subprocess.check_call(["perl", os.path.join(HERE, "make_output_golden.pl"), ref, out])


if __name__ == "__main__":
    main()
