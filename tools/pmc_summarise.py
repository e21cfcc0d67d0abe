"""Mean per-dispatch PMC values of one kernel from a rocprofv3 --pmc csv directory (or several: one per pass).
    python tools/pmc_summarise.py <dir> [<dir> ...] <kernel name substring> [--json out.json --n N --replicas M --steps-per-dispatch K]
With --json: writes the HBM traffic record bench.py reads (profiles/*hbm_traffic*.json): bytes = 2 x FETCH_SIZE (gfx950
tallies a 128-byte request of a wide coalesced read as 64 bytes, MI355X_MICROARCH.md, HBM) + WRITE_SIZE, per SA step."""
import csv, glob, json, os, sys
from collections import defaultdict
args = sys.argv[1:]
opts = {}
while "--json" in args or "--n" in args or "--replicas" in args or "--steps-per-dispatch" in args:
    for k in ("--json", "--n", "--replicas", "--steps-per-dispatch"):
        if k in args:
            i = args.index(k); opts[k] = args[i + 1]; del args[i:i + 2]
dirs, pat = args[:-1], args[-1]
acc, cnt, names = defaultdict(float), defaultdict(int), set()
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
                names.add(row["Kernel_Name"].split("(")[0].replace("void ", ""))
for k in sorted(acc):
    print(f"{k:24s} {acc[k] / cnt[k]:16.1f}   (mean of {cnt[k]} dispatches)")
if "SQ_WAVE_CYCLES" in acc:
    wc = acc["SQ_WAVE_CYCLES"] / cnt["SQ_WAVE_CYCLES"]
    for k in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
        if k in acc:
            print(f"{k} / SQ_WAVE_CYCLES = {acc[k] / cnt[k] / wc:.3f}")
if "--json" in opts:
    K = int(opts.get("--steps-per-dispatch", 1))
    fetch_kb = acc["FETCH_SIZE"] / cnt["FETCH_SIZE"]; write_kb = acc["WRITE_SIZE"] / cnt["WRITE_SIZE"]
    rec = {"kernel": sorted(names)[0] if names else pat, "n": int(opts["--n"]), "replicas": int(opts["--replicas"]),
           "sa_steps_per_dispatch": K, "FETCH_SIZE_KB_per_dispatch": round(fetch_kb, 1), "WRITE_SIZE_KB_per_dispatch": round(write_kb, 1),
           "hbm_bytes_per_sa_step": round((2.0 * fetch_kb + write_kb) * 1024.0 / K, 1),
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction)"}
    json.dump(rec, open(opts["--json"], "w"), indent=1)
    print("wrote", opts["--json"], rec["hbm_bytes_per_sa_step"], "bytes per SA step")
