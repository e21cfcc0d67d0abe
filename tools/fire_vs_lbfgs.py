"""The one hot-path stage whose algorithm differs from the reference's by choice: `minimize lbfgs nstep=15000 drop=10` x 10 (deck
chromosome3D.pl:1790-1803) is FIRE here.  From the device's own post-cooling coordinates: the device's final stage (shipped fp32 kernels,
gradient exit) against L-BFGS (scipy L-BFGS-B, 10 correction pairs, up to 10 restarts like the deck's, on the CPU restatement's fp64
energy and analytic gradient).  Per replica: total energy of both end points, their difference relative to the energy, RMS / max
difference of the pair distances, difference of Spearman(IF, 1/d), int(E_noe) of both.

    python tools/fire_vs_lbfgs.py [replicas=6] [chromosomes ...]        (committed: profiles/r05_fire_vs_lbfgs.txt)
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.optimize import minimize
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, make_stages, pipeline
from oracle import oracle as O
from tests.util import load_if, oracle_model_from

nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cids = sys.argv[2:] or ["chr21_1mb", "chr13_1mb", "chr4_1mb", "chr1_500kb"]
s = Solver(0)
print("| matrix | N | replica | f FIRE (device) | f L-BFGS | (FIRE - L-BFGS) / f | int(E_noe) FIRE / L-BFGS | pair distances: RMS / max difference (A) | Spearman FIRE / L-BFGS | FIRE steps | L-BFGS iterations (cycles) |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for cid in cids:
    IF = load_if(cid); n = IF.shape[0]
    rows = [(t.kind, t.nsteps, t.dt, t.w_all, t.w_vdw, t.repel_s, t.t_bath) for t in default_schedule(3000)]
    rows[-1] = (2,) + rows[-1][1:]              # FIRE is what is compared here (the shipped final stage is kind 5 since round 5)
    m = default_model(); s.set_model(m)
    d10 = pipeline.IF2dist_new(s, IF)
    s.set_schedule(make_stages(rows[:-1]), default_fire(), 0.0, 250); s.init_replicas(nrep, 82364, 0); s.run()
    x0 = s.coords()
    s.set_schedule(make_stages(rows[-1:]), default_fire(), 1e-2, 250); s.init_replicas(nrep, 82364, 0); s.set_coords(x0); s.run()
    xf = s.coords().astype(np.float64); fsteps = s.last_timing()[1]
    om = oracle_model_from(m, n); w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]
    def fg(u):
        F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
        return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
    i, j = np.triu_indices(n, 1)
    rel = []
    for r in range(nrep):
        u = x0[r].astype(np.float64).ravel(); nit = 0
        for cycle in range(10):
            res = minimize(fg, u, jac=True, method="L-BFGS-B", options=dict(maxiter=15000, maxfun=150000, ftol=1e-15, gtol=1e-5, maxcor=10))
            u = res.x; nit += res.nit
            if res.success: break
        xl = res.x.reshape(n, 3); ff = fg(xf[r].ravel())[0]
        ef = O.energy_force(om, d10, xf[r], w_all, w_vdw, rs)[1][0]; el = O.energy_force(om, d10, xl, w_all, w_vdw, rs)[1][0]
        dd = np.linalg.norm(xf[r][i] - xf[r][j], axis=1) - np.linalg.norm(xl[i] - xl[j], axis=1)
        rf = -pipeline.spearman_IF_pdb(IF, xf[r].astype(np.float32)); rl = -pipeline.spearman_IF_pdb(IF, xl.astype(np.float32))
        rel.append((ff - res.fun) / abs(res.fun))
        print(f"| {cid} | {n} | {r} | {ff:.3f} | {res.fun:.3f} | {rel[-1]:+.2e} | {int(ef)} / {int(el)} | {np.sqrt((dd ** 2).mean()):.4f} / {np.abs(dd).max():.3f} | {rf:.5f} / {rl:.5f} | {fsteps} | {nit} ({cycle + 1}) |", flush=True)
    rel = np.array(rel)
    print(f"| {cid} | | all {nrep} | | | mean {rel.mean():+.2e}, |max| {np.abs(rel).max():.2e}, FIRE lower on {(rel < 0).sum()} | | | | | |", flush=True)
