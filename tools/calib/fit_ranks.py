"""Round 5: the reference's own ENERGY RANKS as calibration data.  Every bundled file name carries the model's rank by CNS NOE energy in the
reference's run of 20 (rankNN; chromosome3D.pl:796-802, 822-828).  Under a good energy model the bundled model, relaxed, sits among OUR 20
annealed replicas where that rank says: z = (E_relaxed_bundled - mean E_ours) / sd E_ours should be the normal score of rank k of 20.

Objective on the 23 matrices at 1 Mb (the 22 at 500 kb are held out and printed beside it):
    J = mean (z - z_k)^2 / 1.0  +  mean dRMSD(relaxed bundled vs bundled) / 0.7  +  mean |d rho best-energy| / 0.004
(the second term is round 4's relaxation criterion measured on the device, the third the parity table's own figure — reported, weight
`w_rho`, default 0: NOT fitted unless asked).

    python tools/calib/fit_ranks.py eval '{"asym": 3.0, ...}' ...            one line per candidate, both halves
    python tools/calib/fit_ranks.py fit '{"free": {"mrswitch": [10, 1.0], ...}, "fixed": {...}, "maxit": 120}'
"""
import glob, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from scipy.stats import norm, spearmanr
from chromosome3d_amd import Solver
from tests.util import bundled_rank, load_pdb_xyz, relax_reference_model, structure_report
import parity_sweep as ps

data = {}
for cid in ps.all_cids():
    ref = glob.glob(f"{ps.ALL}/{cid}_rank*_a11.pdb")
    IF = ps.load(cid)
    data[cid] = (IF, load_pdb_xyz(ref[0]), bundled_rank(ref[0]))
TRAIN = [c for c in data if c.endswith("_1mb")]
TEST = [c for c in data if c.endswith("_500kb")]
s = Solver(0)
ZK = {k: float(norm.ppf((k - 0.375) / 20.25)) for k in range(1, 21)}       # Blom's normal scores of rank k of 20


def evaluate(kw, subset, seed=82364):
    kw = dict(kw)
    for key in ("msoexp", "noe_pot", "min_sep", "rep_sep", "ang_mode"):
        if key in kw:
            kw[key] = int(kw[key])
    out = []
    for cid in subset:
        IF, Xr, rank = data[cid]
        try:
            x, e, rows = ps.solve(s, IF, kw, 20, seed)
            rep = structure_report(IF, x, e[:, 0], Xr, rank)
            rel = relax_reference_model(s, Xr, e[:, 0])
        except Exception as ex:
            return None
        en = e[:, 0]
        z = (rel["e_noe"] - en.mean()) / max(en.std(), 1e-9)
        out.append(dict(cid=cid, z=z, zk=ZK[rank], rank_ours=rel["rank_in_ours"], rank=rank, moved=rel["moved"][1], delta=rep["delta"], dtop=rep["delta_max_top10"]))
    return out


def terms(res, w_rho=0.0):
    z = np.array([r["z"] for r in res]); zk = np.array([r["zk"] for r in res]); mv = np.array([r["moved"] for r in res])
    d = np.array([r["delta"] for r in res]); ro = np.array([r["rank_ours"] for r in res]); rk = np.array([r["rank"] for r in res])
    zc = np.clip(z, -4, 4)
    t = dict(zerr=float(((zc - zk) ** 2).mean()), moved=float(mv.mean()), drho=float(np.abs(d).mean()), within=int((np.abs(d) <= 0.01).sum()),
             rankcorr=float(spearmanr(ro, rk)[0]), top10=int((ro <= 10).sum()), near5=int((np.abs(ro - rk) <= 5).sum()), n=len(res), worst=float(np.abs(d).max()))
    t["J"] = t["zerr"] / 1.0 + t["moved"] / 0.7 + w_rho * t["drho"] / 0.004
    return t


def fmt(t):
    return (f"J={t['J']:.3f} z-err {t['zerr']:.3f} | relaxed bundled moved {t['moved']:.3f} A | rank corr {t['rankcorr']:+.3f}, ours<=10: {t['top10']}/{t['n']}, |ours-file|<=5: {t['near5']} | "
            f"|d rho| {t['drho']:.4f} max {t['worst']:.4f} within 0.01: {t['within']}/{t['n']}")


def mode_eval(cands):
    for over in cands:
        t0 = time.time()
        for name, sub in (("1mb  ", TRAIN), ("500kb", TEST)):
            res = evaluate(over, sub)
            print(f"{json.dumps(over):60s} {name}: " + (fmt(terms(res)) if res else "FAILED"), flush=True)
            if res and name.startswith("1mb"):
                print("      " + " ".join(f"{r['cid'][3:-4]}:{r['rank']}>{r['rank_ours']}" for r in res), flush=True)
        print(f"      ({time.time() - t0:.1f} s)", flush=True)


def mode_fit(spec):
    from scipy.optimize import minimize
    free = spec["free"]; fixed = spec.get("fixed", {}); maxit = spec.get("maxit", 120); w_rho = spec.get("w_rho", 0.0)
    names = list(free)
    x0 = np.array([free[k][0] for k in names], float); sc = np.array([free[k][1] for k in names], float)
    best = [1e9, None]; cnt = [0]

    def J(u):
        over = dict(fixed)
        for k, val in zip(names, x0 + u * sc):
            over[k] = float(val)
        if any(v <= 0 for k, v in over.items() if k not in ("masym",)):
            return 1e3
        res = evaluate(over, TRAIN)
        if res is None:
            return 1e3
        t = terms(res, w_rho); cnt[0] += 1
        if t["J"] < best[0]:
            best[0] = t["J"]; best[1] = over
            print(f"[{cnt[0]:4d}] " + " ".join(f"{k}={over[k]:.4g}" for k in names) + " | " + fmt(t), flush=True)
        return t["J"]
    t0 = time.time(); nn = len(names)
    minimize(J, np.zeros(nn), method="Nelder-Mead", options=dict(maxfev=maxit, xatol=2e-2, fatol=1e-3,
             initial_simplex=np.vstack([np.zeros(nn)] + [np.eye(nn)[k] * (1.0 if k % 2 == 0 else -1.0) for k in range(nn)])))
    print("final", json.dumps(best[1]), f"J {best[0]:.4f}", f"{time.time() - t0:.0f} s", flush=True)
    mode_eval([{}, best[1]])
    for seed in (1, 7):
        for name, sub in (("1mb  ", TRAIN), ("500kb", TEST)):
            for over in ({}, best[1]):
                res = evaluate(over, sub, seed)
                print(f"seed {seed} {'shipped' if not over else 'fitted '} {name}: " + (fmt(terms(res)) if res else "FAILED"), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "eval"
    if mode == "eval":
        mode_eval([json.loads(a) for a in sys.argv[2:]] or [{}])
    else:
        mode_fit(json.loads(sys.argv[2]))
