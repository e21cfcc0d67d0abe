"""Stand-in for the one input the reference does not ship: input/chr2_500kb_matrix.txt (.MISSING_LARGE_BLOBS:1), the
largest job of BASELINE configs[3] ("all 23 chromosomes at 500 kb", test.sh:9-12).

What the reference DOES ship for chr2 at 500 kb is its model, output_models/chr2_500kb_rank01_a11.pdb (N = 479 beads;
committed with the other bundled models under tests/golden/all45).  The stand-in is the Hi-C matrix that model implies,
built with SURVEY 8d's config-5 recipe: IF_ij = (K / d_ij)^(1/alpha) x log-normal noise (sigma 0.2, symmetrised,
numpy default_rng(20161015)), K = 11, alpha = 0.5, the diagonal set so that mean(IF^alpha) = 1 — then IF2dist_new
(chromosome3D.pl:110-162) turns it back into targets d_ij x noise.  It is a workload of the right size and statistics for
the LPT schedule and the batch tests, NOT parity evidence: the file carries `standin = 1` and every parity table skips it.

    python tools/make_chr2_standin.py      (needs only tests/golden/all45/chr2_500kb_rank01_a11.pdb)
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import load_pdb_xyz
ALL = os.path.join(ROOT, "tests", "golden", "all45")
K, ALPHA, SIGMA = 11.0, 0.5, 0.2
x = load_pdb_xyz(os.path.join(ALL, "chr2_500kb_rank01_a11.pdb"))
n = len(x)
assert n == 479
d = np.linalg.norm(x[:, None] - x[None], axis=-1)
np.fill_diagonal(d, 1.0)
rng = np.random.default_rng(20161015)
g = rng.normal(size=(n, n))
g = (g + g.T) / np.sqrt(2.0)
P = (K / np.maximum(d, 0.5)) * np.exp(ALPHA * SIGMA * g)          # IF^alpha off the diagonal
np.fill_diagonal(P, 0.0)
diag = (n * n - P.sum()) / n                                       # mean(IF^alpha) over all N^2 entries = 1
assert diag > P.max()
np.fill_diagonal(P, diag)
IF = P ** (1.0 / ALPHA)
IF = (IF + IF.T) / 2.0
np.savez_compressed(os.path.join(ALL, "chr2_500kb_upper.npz"), n=n, upper=IF[np.triu_indices(n)], standin=1)
T = K * (IF ** ALPHA).mean() / IF ** ALPHA
i, j = np.triu_indices(n, 5)
print(f"chr2_500kb stand-in: N = {n}, restraints {len(i)}, target / model distance: median {np.median(T[i, j] / d[i, j]):.3f}, "
      f"5-95 % {np.percentile(T[i, j] / d[i, j], 5):.3f}-{np.percentile(T[i, j] / d[i, j], 95):.3f}")
