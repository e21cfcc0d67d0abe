"""Calibrate the bead-model parameters on the 1 Mb chromosomes (GPU), validate on the 500 kb ones.
Objective: mean |Spearman(ours best-ranked) - Spearman(bundled reference model)| over the training set.
    python tools/calib/fit_gpu.py [max_evals=60]
"""
import glob, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from scipy.optimize import minimize
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline

ALL = os.path.join(ROOT, "tests", "golden", "all45")
def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m
def load_pdb(p):
    return np.array([[float(l[30:38]), float(l[38:46]), float(l[46:54])] for l in open(p) if l.startswith("ATOM")])
cids = sorted(os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_upper.npz"))
data = {}
for c in cids:
    IF = load(c); ref = glob.glob(f"{ALL}/{c}_rank*_a11.pdb")
    if not ref: continue
    X = load_pdb(ref[0])
    if len(X) != IF.shape[0]: continue
    data[c] = (IF, -pipeline.spearman_IF_pdb(IF, X))
train = [c for c in data if c.endswith("_1mb")]
test = [c for c in data if c.endswith("_500kb")]
s = Solver(0)
NREP = 8
def evaluate(params, names, subset):
    kw = dict(zip(names, params))
    if "mrswitch" in kw: kw["masym"] = 2.0 * kw["mrswitch"]      # clamp-style lower tail
    d = []
    for c in subset:
        IF, ref = data[c]
        s.set_model(default_model(**kw)); pipeline.IF2dist_new(s, IF)
        s.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250)
        s.init_replicas(NREP, 82364, 0)
        try:
            s.run()
        except Exception:
            d.append(0.2); continue
        x, e = s.coords(), s.energies()
        rho = -pipeline.spearman_IF_models(IF, x)
        d.append(rho[np.argmin(e[:, 0].astype(np.int64))] - ref)
    return np.array(d)
names = ["k_bond", "k_ang", "a0", "r0_rep", "k_rep", "mrswitch"]
x0 = np.array([700.0, 80.0, 7.4, 6.75, 1.0, 11.0])
scale = np.array([200.0, 30.0, 0.3, 0.4, 0.5, 2.0])
hist = []
def obj(z):
    p = x0 + z * scale
    if (p <= 0).any(): return 1.0
    d = evaluate(p, names, train)
    f = float(np.abs(d).mean())
    hist.append((f, p.tolist()))
    print(f"eval {len(hist):3d}: mean|d|={f:.5f} within0.01={(np.abs(d) <= 0.01).sum()}/{len(d)} params={np.round(p, 3).tolist()}", flush=True)
    return f
maxev = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.time()
res = minimize(obj, np.zeros(len(x0)), method="Nelder-Mead", options={"maxfev": maxev, "xatol": 0.05, "fatol": 2e-5, "initial_simplex": np.vstack([np.zeros(len(x0)), np.eye(len(x0))])})
best = min(hist)[1]
print("best train params", dict(zip(names, np.round(best, 3))), "in", round(time.time() - t0), "s")
NREP = 20
for label, p in (("start", x0), ("fitted", np.array(best))):
    for sname, sub in (("train 1mb", train), ("held-out 500kb", test)):
        d = evaluate(p, names, sub)
        print(f"{label:7s} {sname:15s}: mean|d|={np.abs(d).mean():.4f} median={np.median(np.abs(d)):.4f} within0.01={(np.abs(d) <= 0.01).sum()}/{len(d)} within0.02={(np.abs(d) <= 0.02).sum()}/{len(d)} bias={d.mean():+.4f}", flush=True)
