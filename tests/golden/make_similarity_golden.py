"""Parse the reference's output_models/similarity.txt (a data file) into tests/golden/similarity_reference.json.

    python tests/golden/make_similarity_golden.py [/root/reference]
Entries: model id of the 500 kb model -> {"spearman": .., "rmsd": ..} (the file lists chr23 three times with the
same numbers; the last one wins).
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
out = {}
cur = None
for line in open(os.path.join(ref, "output_models", "similarity.txt")):
    line = line.strip()
    if not line:
        continue
    m = re.match(r"Spearman correlation:\s*(\S+)", line)
    if m:
        out[cur]["spearman"] = float(m.group(1))
        continue
    m = re.match(r"RMSD:\s*(\S+)", line)
    if m:
        out[cur]["rmsd"] = float(m.group(1))
        continue
    cur = line
    out[cur] = {}
here = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(here, "similarity_reference.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
print(len(out), "entries")
