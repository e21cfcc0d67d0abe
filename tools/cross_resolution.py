"""Cross-resolution consistency on the GPU: for every chromosome with a 500 kb and a 1 Mb matrix, solve both
(20 replicas, best-ranked model), reduce the 500 kb model to 1 Mb resolution and compare with the 1 Mb model the
way the reference's output_models/similarity.txt does (Spearman / scaled RMS of the pairwise distances).

    python tools/cross_resolution.py [replicas=20]
Needs tests/golden/all45 (tools/pack_all_inputs.py).  Prints a markdown table next to the reference's numbers
(tests/golden/similarity_reference.json).
"""
import glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline

ALL = os.path.join(ROOT, "tests", "golden", "all45")
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ref = json.load(open(os.path.join(ROOT, "tests", "golden", "similarity_reference.json")))
ref_by_chr = {re.match(r"(chr\d+)_", k).group(1): v for k, v in ref.items()}


def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m


def best_model(s, IF):
    s.set_model(default_model())
    pipeline.IF2dist_new(s, IF)
    s.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250)
    s.init_replicas(nrep, 82364, 0)
    s.run()
    return s.coords()[s.rank()[0]].astype(np.float64)


s = Solver(0)
chrs = sorted({re.match(r"(chr\d+)_", os.path.basename(p)).group(1) for p in glob.glob(f"{ALL}/*_500kb_upper.npz")}, key=lambda c: int(c[3:]))
print("| chromosome | N 500kb | N 1mb | Spearman ours | Spearman ref | RMSD ours | RMSD ref |")
print("|---|---|---|---|---|---|---|")
ours, theirs = [], []
for c in chrs:
    if not os.path.exists(f"{ALL}/{c}_1mb_upper.npz"):
        continue
    a, b = load(f"{c}_500kb"), load(f"{c}_1mb")
    xa, xb = best_model(s, a), best_model(s, b)
    rho, rmsd = pipeline.model_similarity(pipeline.reduce_model(xa), xb)
    r = ref_by_chr.get(c)
    if r:
        ours.append(rho); theirs.append(r["spearman"])
    print(f"| {c} | {a.shape[0]} | {b.shape[0]} | {rho:.4f} | {r['spearman']:.4f} | {rmsd:.2f} | {r['rmsd']:.2f} |" if r else
          f"| {c} | {a.shape[0]} | {b.shape[0]} | {rho:.4f} | - | {rmsd:.2f} | - |", flush=True)
ours, theirs = np.array(ours), np.array(theirs)
print(f"\n{len(ours)} chromosomes with a reference value: mean Spearman ours {ours.mean():.4f} / reference {theirs.mean():.4f}; "
      f"min ours {ours.min():.4f} / reference {theirs.min():.4f}")
