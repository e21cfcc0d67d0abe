"""All-chromosome parity sweep on the GPU: Spearman(IF, 1/d) of our ranked models vs the bundled
reference model of every chromosome, plus chain statistics.

    python tools/parity_sweep.py [json model overrides] [replicas=20] [subset regex]
Needs tests/golden/all45 (tools/pack_all_inputs.py; git-ignored data, present on the GPU box through
the gpurun snapshot).  Prints a markdown table; the committed copy lives in profiles/.
"""
import glob, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline

ALL = os.path.join(ROOT, "tests", "golden", "all45")
over = json.loads(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].startswith("{") else {}
nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
subset = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
min_steps = int(over.pop("min_steps", 3000))
quiet = over.pop("quiet", 0)
embed = over.pop("embed", 0)
seed = int(over.pop("seed", 82364))

def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m
def load_pdb(p):
    return np.array([[float(l[30:38]), float(l[38:46]), float(l[46:54])] for l in open(p) if l.startswith("ATOM")])
def key(c):
    a, b = re.match(r"chr(\d+)_(\w+)", c).groups(); return (b, int(a))

s = Solver(0)
cids = sorted({os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_upper.npz")}, key=key)
rows, t_all = [], time.time()
for cid in cids:
    if subset and not subset.search(cid): continue
    IF = load(cid); n = IF.shape[0]
    ref = glob.glob(f"{ALL}/{cid}_rank*_a11.pdb")
    Xr = load_pdb(ref[0]) if ref else None
    s.set_model(default_model(**over))
    pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(min_steps), default_fire(), 0.0, 250)
    s.init_replicas(nrep, seed, 0)
    if embed:
        s.embed(50)
    s.run()
    ms = s.last_timing()[0]
    x, e = s.coords(), s.energies()
    rho = -pipeline.spearman_IF_models(IF, x)
    order = np.argsort(e[:, 0].astype(np.int64), kind="stable")
    best = order[0]
    b = np.linalg.norm(x[best, 1:] - x[best, :-1], axis=1)
    rg = np.sqrt(((x[best] - x[best].mean(0)) ** 2).sum(1).mean())
    rr = -pipeline.spearman_IF_pdb(IF, Xr) if Xr is not None and len(Xr) == n else float("nan")
    rgr = np.sqrt(((Xr - Xr.mean(0)) ** 2).sum(1).mean()) if Xr is not None else float("nan")
    rows.append((cid, n, s.num_restraints, rho[best], rho.mean(), rho.max(), rr, rho[best] - rr, b.mean(), b.std(), rg, rgr, ms))
    if not quiet:
        print("| %-12s | %4d | %6d | %.4f | %.4f | %.4f | %.4f | %+.4f | %.2f±%.2f | %.1f / %.1f | %.0f |" % rows[-1], flush=True)
d = np.array([r[7] for r in rows if not np.isnan(r[7])])
print(f"# {len(d)} chromosomes: mean |dSpearman| = {np.abs(d).mean():.4f}, median = {np.median(np.abs(d)):.4f}, "
      f"within 0.01: {(np.abs(d) <= 0.01).sum()}, within 0.02: {(np.abs(d) <= 0.02).sum()}, within 0.03: {(np.abs(d) <= 0.03).sum()}, "
      f"max = {np.abs(d).max():.4f}, bias = {d.mean():+.4f}; overrides {over}; total {time.time() - t_all:.1f} s", flush=True)
