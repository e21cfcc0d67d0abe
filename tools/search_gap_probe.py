"""Round 6 probe of the one SEARCH gap the parity work found: on chr13_1mb the reference's bundled model, relaxed under our
energy, lies 2.5 % below all 20 of our annealed replicas by total energy.  Is that basin merely rare?  Anneal many more replicas (different
seeds, 56 at a time) and count how many end at or below the relaxed bundled model's total energy; print where their Spearman lands.
    python tools/search_gap_probe.py [matrix=chr13_1mb] [replicas=560]        (GPU box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
from tests.util import load_pdb_xyz, model_pdb, relax_reference_model, REF_SPEARMAN
from tools.parity_sweep import load

if __name__ == "__main__":
    cid = sys.argv[1] if len(sys.argv) > 1 else "chr13_1mb"
    total = int(sys.argv[2]) if len(sys.argv) > 2 else 560
    IF = load(cid)
    n = IF.shape[0]
    s = Solver(0)
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    per = 56 if n <= 189 else 20
    E, RHO = [], []
    for b in range(0, total, per):
        s.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250)
        s.init_replicas(per, 82364, b)                    # replica ids b .. b + per - 1: ids 0..19 are the product's 20
        s.run()
        e = s.energies()
        E.append(e.copy())
        RHO.append(-pipeline.spearman_IF_models(IF, s.coords()))
    E = np.concatenate(E)[:total]
    RHO = np.concatenate(RHO)[:total]
    tot = E.sum(axis=1)
    ref = relax_reference_model(s, load_pdb_xyz(model_pdb(cid)), E[:20, 0])
    ref_tot = float(ref["e3"].sum())
    ref_rho = -float(pipeline.spearman_IF_pdb(IF, ref["xyz"]))
    print(f"{cid}: N = {n}, {len(tot)} anneals (replica ids 0..{len(tot) - 1}); reference's bundled model relaxed under our energy: E_noe {ref['e_noe']:.0f}, total {ref_tot:.0f}, "
          f"Spearman(IF,1/d) {ref_rho:.4f} (bundled: {-REF_SPEARMAN[cid]:.4f})")
    for name, v, r in (("E_noe", E[:, 0], ref["e_noe"]), ("total energy", tot, ref_tot)):
        first20 = v[:20]
        below = int((v <= r).sum())
        print(f"  by {name}: our first 20 span {first20.min():.0f} .. {first20.max():.0f} (the relaxed bundled model lies {100 * (first20.min() - r) / r:.2f} % below their best); "
              f"of all {len(v)}: min {v.min():.0f} ({100 * (v.min() - r) / r:+.2f} %), {below} at or below the bundled model's, "
              f"{int((v <= r * 1.005).sum())} within 0.5 %, {int((v <= r * 1.01).sum())} within 1 %")
    k = int(np.argmin(E[:, 0]))
    print(f"  best-energy replica of all {len(tot)}: id {k}, Spearman {RHO[k]:.4f} (d {RHO[k] + REF_SPEARMAN[cid]:+.4f}); of the first 20: id {int(np.argmin(E[:20, 0]))}, "
          f"Spearman {RHO[int(np.argmin(E[:20, 0]))]:.4f} (d {RHO[int(np.argmin(E[:20, 0]))] + REF_SPEARMAN[cid]:+.4f})")
    order = np.argsort(tot)[:5]
    print("  five lowest by total energy: " + ", ".join(f"id {int(i)} total {tot[i]:.0f} Spearman {RHO[i]:.4f}" for i in order))
