"""Pack every bundled reference matrix + model into tests/golden/all45/ (git-ignored, travels with
gpurun snapshots) for the all-chromosome parity sweep.  Runs only where /root/reference exists."""
import glob, os, shutil, sys
import numpy as np
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "all45")
os.makedirs(OUT, exist_ok=True)
for p in sorted(glob.glob(f"{REF}/input/*_matrix.txt")):
    cid = os.path.basename(p)[:-len("_matrix.txt")]
    m = np.array([[float(t) for t in l.split()] for l in open(p) if l.strip()])
    assert np.array_equal(m, m.T), cid
    np.savez_compressed(f"{OUT}/{cid}_upper.npz", n=m.shape[0], upper=m[np.triu_indices(m.shape[0])])
    for q in glob.glob(f"{REF}/output_models/{cid}_rank*_a11.pdb"):
        shutil.copyfile(q, f"{OUT}/{os.path.basename(q)}")
# the one bundled model whose matrix the reference does not ship (.MISSING_LARGE_BLOBS:1): tools/make_chr2_standin.py builds on it
shutil.copyfile(f"{REF}/output_models/chr2_500kb_rank01_a11.pdb", f"{OUT}/chr2_500kb_rank01_a11.pdb")
print("packed", len(glob.glob(f"{OUT}/*.npz")), "matrices")
