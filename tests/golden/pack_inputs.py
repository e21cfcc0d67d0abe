#!/usr/bin/env python3
"""Pack bundled reference DATA (Hi-C matrices, ranked example models) into small fixtures.

TEST INFRASTRUCTURE, runs only where /root/reference is mounted.  Data files only: the
input IF matrices (reference input/*.txt) and the example output models
(reference output_models/*.pdb).  Small matrices are copied verbatim (they exercise the
text parser incl. the " \\r\\n" line ends); large symmetric ones are stored as the exact
float64 upper triangle (np.savez_compressed), from which tests re-create a text file
with repr() (round-trips every double exactly).

    python tests/golden/pack_inputs.py [/root/reference]
"""
import os
import shutil
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
VERBATIM = ["chr21_1mb", "chr22_1mb"]
PACKED = ["chr1_500kb", "chr19_500kb", "chr13_1mb", "chr20_1mb", "chr4_1mb", "chr21_500kb"]


def load_if(path):
    rows = []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if t:
                rows.append([float(x) for x in t])
    return np.array(rows, dtype=np.float64)


def main():
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    os.makedirs(os.path.join(HERE, "models"), exist_ok=True)
    for cid in VERBATIM:
        shutil.copyfile(f"{REF}/input/{cid}_matrix.txt", f"{HERE}/inputs/{cid}_matrix.txt")
    for cid in PACKED:
        m = load_if(f"{REF}/input/{cid}_matrix.txt")
        assert m.shape[0] == m.shape[1]
        assert np.array_equal(m, m.T), f"{cid} not symmetric"
        iu = np.triu_indices(m.shape[0])
        np.savez_compressed(f"{HERE}/inputs/{cid}_upper.npz", n=m.shape[0], upper=m[iu])
    for cid in VERBATIM + PACKED:
        import glob
        for p in glob.glob(f"{REF}/output_models/{cid}_rank*_a11.pdb"):
            shutil.copyfile(p, f"{HERE}/models/{os.path.basename(p)}")
    print("packed", VERBATIM + PACKED)


if __name__ == "__main__":
    main()
