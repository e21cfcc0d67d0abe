"""Mean per-dispatch PMC values of one kernel from a rocprofv3 --pmc csv directory.
    python tools/pmc_summarise.py <dir> <kernel name substring>"""
import csv, glob, os, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2]
acc, cnt = defaultdict(float), defaultdict(int)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if pat in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
for k in sorted(acc):
    print(f"{k:24s} {acc[k] / cnt[k]:16.1f}   (mean of {cnt[k]} dispatches)")
if "SQ_WAVE_CYCLES" in acc:
    wc = acc["SQ_WAVE_CYCLES"] / cnt["SQ_WAVE_CYCLES"]
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in acc:
            print(f"{k} / SQ_WAVE_CYCLES = {acc[k] / cnt[k] / wc:.3f}")
