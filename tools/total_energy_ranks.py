"""The search-gap question of DESIGN.md 2 (item 4), asked of the quantity the minimiser minimises: the reference ranks its models by NOE
energy alone (chromosome3D.pl:796-802), and on a dozen matrices the bundled model, relaxed under our energy, has a lower E_noe than all 20
of ours.  Is its TOTAL energy (E_noe + bond/angle + repel at the final stage's weights, what FIRE and the anneal descend on) lower too?
    python tools/total_energy_ranks.py [matrix regex]"""
import glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chromosome3d_amd import Solver
from tests.util import bundled_rank, load_pdb_xyz, relax_reference_model
from tools.parity_sweep import ALL, all_cids, load, solve

if __name__ == "__main__":
    subset = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
    s = Solver(0)
    print("| matrix | N | file rank | relaxed bundled: E_noe / chain / repel / total | ours, 20 replicas: E_noe min .. max | chain min .. max | repel min .. max | total min .. max |"
          " rank by E_noe | rank by total | (total - our lowest total) / total |\n|" + "---|" * 11)
    rows = []
    for cid in all_cids():
        if subset and not subset.search(cid):
            continue
        ref = glob.glob(f"{ALL}/{cid}_rank*_a11.pdb")
        if not ref:
            continue
        IF = load(cid); Xr = load_pdb_xyz(ref[0])
        if len(Xr) != IF.shape[0]:
            continue
        x, e, _ = solve(s, IF, {})
        relax_reference_model(s, Xr, e[:, 0])
        er = s.energies()[0]
        tot, tr = e.sum(axis=1), er.sum()
        rk_noe = int(1 + (e[:, 0].astype(np.int64) < int(er[0])).sum()); rk_tot = int(1 + (tot < tr).sum())
        rows.append((cid, rk_noe, rk_tot, (tr - tot.min()) / tot.min(), er[1] - np.median(e[:, 1]), er[2] - np.median(e[:, 2])))
        print(f"| {cid} | {len(Xr)} | {bundled_rank(ref[0])} | {er[0]:.0f} / {er[1]:.0f} / {er[2]:.0f} / {tr:.0f} | {e[:, 0].min():.0f} .. {e[:, 0].max():.0f} | "
              f"{e[:, 1].min():.0f} .. {e[:, 1].max():.0f} | {e[:, 2].min():.0f} .. {e[:, 2].max():.0f} | {tot.min():.0f} .. {tot.max():.0f} | {rk_noe} | {rk_tot} | {rows[-1][3]:+.4f} |", flush=True)
    a = np.array([r[1:] for r in rows], dtype=float)
    print(f"# {len(rows)} matrices: relaxed bundled model below all 20 of ours by E_noe on {int((a[:, 0] == 1).sum())}, by TOTAL energy on {int((a[:, 1] == 1).sum())}; "
          f"above all of ours: {int((a[:, 0] == 21).sum())} / {int((a[:, 1] == 21).sum())}; median rank {np.median(a[:, 0]):.0f} / {np.median(a[:, 1]):.0f}; "
          f"rows that are rank 1 by E_noe: their rank by total {sorted(a[a[:, 0] == 1, 1].astype(int).tolist())}; "
          f"chain energy relaxed - our median: mean {a[:, 3].mean():+.0f}, repel: mean {a[:, 4].mean():+.0f}")
