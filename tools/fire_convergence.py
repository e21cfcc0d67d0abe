"""Convergence of the final minimisation (the stand-in for the reference's 10 x 15000-step L-BFGS, deck :1790-1803) — the stage as shipped
(two-point step sizes, FIRE after 1000 steps; FINAL_MINIMISER=0 in the environment: FIRE throughout, rounds 1-4):
largest RMS force component over 20 replicas against minimiser steps, for BASELINE configs 2 and 3 and a mid-size matrix.
    python tools/fire_convergence.py [cid ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, pipeline
from tests.util import load_if
s = Solver(0)
s.set_option("final_minimiser", int(os.environ.get("FINAL_MINIMISER", "1")))
for cid in sys.argv[1:] or ["chr21_1mb", "chr4_1mb", "chr1_500kb"]:
    IF = load_if(cid)
    s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
    MIN = 12000
    s.set_schedule(default_schedule(MIN), None, 0.0, 250)
    L = s.schedule_length
    s.init_replicas(20, 82364, 0)
    s.run_steps(L - MIN)                      # everything before the final minimisation
    print(f"== {cid} N={IF.shape[0]}: final minimisation after {L - MIN} SA steps; gtol of the shipped schedule = 1e-2 kcal/mol/A")
    done, first = 0, None
    for chunk in [50, 50, 100, 100, 200, 250, 250, 500, 500, 1000, 1000, 2000, 2000, 4000]:
        s.run_steps(chunk); done += chunk
        rms = s.stat("rms_force"); e = s.energies()[:, 0]
        rho = -pipeline.spearman_IF_models(IF, s.coords())
        if first is None and rms < 1e-2: first = done
        print(f"   {done:6d} minimiser steps: max RMS force {rms:10.4g}   E_noe median {np.median(e):14.2f}   Spearman(IF,1/d) mean {rho.mean():.5f}", flush=True)
    print(f"   gtol 1e-2 first met at <= {first} steps" if first else "   gtol 1e-2 not met")
