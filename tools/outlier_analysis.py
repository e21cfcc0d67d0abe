"""Where do our best-ranked models differ from the bundled reference models?  Bond statistics, radius of gyration and the
restraint residuals by target class, for a list of chromosomes (default: the parity outliers).
    python tools/outlier_analysis.py [cid ...] [json model overrides]"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline
ALL = os.path.join(ROOT, "tests", "golden", "all45")
args = [a for a in sys.argv[1:] if not a.startswith("{")]
over = json.loads([a for a in sys.argv[1:] if a.startswith("{")][0]) if any(a.startswith("{") for a in sys.argv[1:]) else {}
cids = args or ["chr21_500kb", "chr22_1mb", "chr17_1mb", "chr7_1mb", "chr22_500kb", "chr13_1mb", "chr16_1mb", "chr10_1mb", "chr20_1mb", "chr21_1mb", "chr1_500kb"]
def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m
def load_pdb(p):
    return np.array([[float(l[30:38]), float(l[38:46]), float(l[46:54])] for l in open(p) if l.startswith("ATOM")])
s = Solver(0)
for cid in cids:
    IF = load(cid); n = IF.shape[0]
    Xr = load_pdb(glob.glob(f"{ALL}/{cid}_rank*_a11.pdb")[0])
    s.set_model(default_model(**over))
    d10 = pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
    s.init_replicas(20, 82364, 0); s.run()
    x, e = s.coords(), s.energies()
    best = int(np.argsort(e[:, 0].astype(np.int64), kind="stable")[0])
    Xo = x[best].astype(np.float64)
    i, j = np.triu_indices(n, 5)
    t = d10[i, j] / 10.0; ok = t > 0
    i, j, t = i[ok], j[ok], t[ok]
    def dist(X): return np.linalg.norm(X[i] - X[j], axis=1)
    do, dr = dist(Xo), dist(Xr)
    def stats(X):
        b = np.linalg.norm(X[1:] - X[:-1], axis=1); rg = np.sqrt(((X - X.mean(0)) ** 2).sum(1).mean())
        return f"bond {b.mean():.2f}+-{b.std():.2f} max {b.max():.2f} min {b.min():.2f} Rg {rg:.1f}"
    rho_o = -pipeline.spearman_IF_pdb(IF, Xo.astype(np.float32)); rho_r = -pipeline.spearman_IF_pdb(IF, Xr.astype(np.float32))
    def enoe(d):   # X-PLOR soft-square, S = 10
        dl = d - t; return 10.0 * np.where(dl > 1, 2 * dl - 1, dl * dl).sum()
    print(f"== {cid} n={n} R={len(t)}  Spearman ours {rho_o:.4f} ref {rho_r:.4f}  E_noe(soft-square) ours {enoe(do):.0f} ref {enoe(dr):.0f}")
    print("   ours:", stats(Xo)); print("   ref :", stats(Xr))
    for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 40), (40, 1e9)):
        m = (t >= lo) & (t < hi)
        if m.sum() == 0: continue
        print(f"   targets [{lo:g},{hi:g}): {m.sum():6d} pairs  mean(d-t) ours {np.mean(do[m] - t[m]):7.2f} ref {np.mean(dr[m] - t[m]):7.2f}   "
              f"lower violations > 5 A: ours {np.mean((t[m] - do[m]) > 5):.2f} ref {np.mean((t[m] - dr[m]) > 5):.2f}   mean |i-j| {np.mean(j[m] - i[m]):.0f}")
    # beads carrying the huge targets
    big = t > 40
    if big.any():
        beads = np.bincount(np.concatenate([i[big], j[big]]), minlength=n)
        top = np.argsort(-beads)[:8]
        print("   beads with most targets > 40 A:", [(int(b), int(beads[b])) for b in top if beads[b] > 0])
