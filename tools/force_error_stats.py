"""Diagnostic: fp32 device force error vs the fp64 oracle (what tolerance do the parity tests need?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, pipeline
from oracle import oracle as O
from tests.util import load_if, oracle_model_from, random_coil
s = Solver(0)
for cid in ("chr21_1mb", "chr13_1mb", "chr1_500kb"):
    IF = load_if(cid); n = IF.shape[0]; m = default_model(); s.set_model(m)
    d10 = pipeline.IF2dist_new(s, IF); s.init_replicas(3, 1, 0)
    x = np.stack([random_coil(n, 100 + r) * sc for r, sc in zip(range(3), (1.0, 0.4, 0.15))]); s.set_coords(x)
    F, e = s.eval(1.0, 1.0, 0.85); om = oracle_model_from(m, n)
    for r in range(3):
        Fo, eo = O.energy_force(om, d10, x[r].astype(np.float64), 1.0, 1.0, 0.85)
        err = np.abs(F[r] - Fo)
        print(f"{cid} rep{r}: max|dF|/max|F| = {err.max() / np.abs(Fo).max():.2e}, max rel (|F|>1% max) = {(err / np.abs(Fo))[np.abs(Fo) > 0.01 * np.abs(Fo).max()].max():.2e}, rms rel = {np.sqrt((err**2).sum() / (Fo**2).sum()):.2e}, dE rel = {np.abs(e[r] - np.array(eo)).max() / max(eo):.1e}")
