"""How long should the two-point part of the final stage be before FIRE takes over (option final_minimiser_steps)?  All 45 bundled matrices
x 20 replicas x three seeds = 135 whole anneals per setting with the product's exit test (RMS force < 1e-2 every 250 steps, at most 3000
steps): steps of the final stage in all, device ms of the anneals in all, and the anneals that used the stage up without passing the test.
    python tools/final_minimiser_scan.py [hand-over lengths ...=3000 1500 1000 750 500]        (profiles/r05_final_minimiser_ab.md)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, pipeline
from tools.parity_sweep import all_cids, load

if __name__ == "__main__":
    lengths = [int(a) for a in sys.argv[1:]] or [3000, 1500, 1000, 750, 500]
    s = Solver(0)
    cids = list(all_cids())
    mats = {c: load(c) for c in cids}

    def sweep(fm, steps=1000, seeds=(82364, 11, 22)):
        s.set_option("final_minimiser", fm); s.set_option("final_minimiser_steps", steps)
        tot, ms, fails = 0, 0.0, []
        for seed in seeds:
            for cid in cids:
                s.set_model(default_model()); pipeline.IF2dist_new(s, mats[cid])
                s.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250); s.init_replicas(20, seed, 0); s.run()
                t = s.last_timing(); tot += t[1] - 2172; ms += t[0]
                if t[1] >= 5172: fails.append(f"{cid}/{seed}")
        return tot, ms, fails
    print("| final stage | its steps over 135 anneals | device ms of the 135 anneals | anneals that used up the 3000 steps |\n|---|---|---|---|")
    t, ms, f = sweep(0)
    print(f"| FIRE throughout (rounds 1-4; final_minimiser 0) | {t} | {ms:.1f} | {len(f)}: {', '.join(f)} |", flush=True)
    for n in lengths:
        t, ms, f = sweep(1, n)
        print(f"| two-point steps, FIRE after {n} | {t} | {ms:.1f} | {len(f)}{': ' + ', '.join(f) if f else ''} |", flush=True)
    s.set_option("final_minimiser", 1); s.set_option("final_minimiser_steps", 1000)
