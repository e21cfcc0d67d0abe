"""The final minimisation stage (deck chromosome3D.pl:1790-1803) with FIRE (shipped default) and with the two-point step-size minimiser
(option final_minimiser = 1, stage kind 5), whole anneals of 20 replicas, exit at RMS force < 1e-2 tested every `check_every` steps:
SA steps, device ms, launches, the best-energy model's Spearman and the replicas' median total energy.
    python tools/final_minimiser_ab.py [pattern=_500kb] [check_every ...=250 124]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
from tools.parity_sweep import all_cids, load

if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else "_500kb"
    checks = [int(a) for a in sys.argv[2:]] or [250, 124]
    s = Solver(0)
    variants = [(fm, ce) for ce in checks for fm in (0, 1)]
    print("| matrix | N | " + " | ".join(f"{'two-point' if fm else 'FIRE'}, test every {ce}: steps / ms / launches" for fm, ce in variants) +
          " | Spearman of the best-energy model: " + " / ".join(f"{'2pt' if fm else 'FIRE'} {ce}" for fm, ce in variants) + " | median total energy, relative to the first column |\n|" + "---|" * (4 + len(variants)))
    tot = np.zeros((len(variants), 2))
    for cid in all_cids():
        if pat not in cid:
            continue
        IF = load(cid); n = IF.shape[0]; cells, rho, med = [], [], []
        for k, (fm, ce) in enumerate(variants):
            best = None
            for rep in range(3):                          # the fastest of three (the first run of a size pays its plan)
                s.set_model(default_model()); pipeline.IF2dist_new(s, IF)
                s.set_option("final_minimiser", fm)
                s.set_schedule(default_schedule(3000), default_fire(), 1e-2, ce); s.init_replicas(20, 82364, 0); s.run()
                t = s.last_timing()
                if best is None or t[0] < best[0]: best = t
            e = s.energies(); x = s.coords()
            cells.append(f"{best[1]} / {best[0]:.2f} / {best[2]}"); tot[k] += (best[1], best[0])
            rho.append(-pipeline.spearman_IF_pdb(IF, x[int(np.argmin(e[:, 0]))])); med.append(float(np.median(e.sum(axis=1))))
        print(f"| {cid} | {n} | " + " | ".join(cells) + " | " + " / ".join(f"{r:.4f}" for r in rho) + " | " + " / ".join(f"{(m - med[0]) / med[0]:+.1e}" for m in med) + " |", flush=True)
    s.set_option("final_minimiser", 0)
    print("# totals: " + "; ".join(f"{'two-point' if fm else 'FIRE'} every {ce}: {tot[k][0]:.0f} steps, {tot[k][1]:.1f} ms" for k, (fm, ce) in enumerate(variants)))
