"""Wall-clock of the Perl driver's phases (C3D_TIMING=1) on chr1_500kb x 20 models, two runs (the first also pages perl in).
Run on the GPU box: python tools/perl_driver_phases.py"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_if, write_if_text

with tempfile.TemporaryDirectory() as td:
    mat = os.path.join(td, "chr1_500kb_matrix.txt")
    write_if_text(load_if("chr1_500kb"), mat)
    for k in range(3):
        t = time.perf_counter()
        p = subprocess.run(["perl", os.path.join(ROOT, "bin", "chromosome3D_amd.pl"), "-i", mat, "-o", os.path.join(td, f"o{k}"), "-m", "20"],
                           capture_output=True, text=True, env=dict(os.environ, C3D_TIMING="1"))
        print(f"run {k}: rc {p.returncode} wall {time.perf_counter() - t:.3f} s")
        print("".join(l + "\n" for l in p.stderr.splitlines() if l.startswith("[timing]")), end="")
