"""All-chromosome parity sweep on the GPU against the reference's bundled models (output_models/*_a11.pdb, the only
solver evidence the reference holds): per matrix
  * Spearman(IF, 1/d) of our best-ranked and of our RANK-MATCHED replica (rankNN of the bundled file's name) vs the
    bundled model's, and where the reference's value falls inside our 20-replica distribution;
  * structure-level similarity of our model with the bundled one: Spearman of the pairwise distances and the scaled
    dRMSD, the two numbers of output_models/similarity.txt (c3d_model_similarity); for scale, the same between our two
    best replicas, and the reference's own 500 kb vs 1 Mb agreement is 0.855-0.967;
  * the chain envelope: bond mean/sd, |i-j| = 2 mean/sd, radius of gyration ratio.

    python tools/parity_sweep.py ['{json model overrides}'] [replicas=20] [subset regex]
Extra keys in the overrides: min_steps, quiet, embed (DG-embed start, chromosome3D.pl:1471-1525), start (1 = extended strand, :2413-2416), seed, dump (path of an .npz receiving every replica's coordinates).
Prints a markdown table; the committed copy lives in profiles/.
"""
import glob, json, os, re, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_model, default_schedule, default_fire, pipeline
from tests.util import bundled_rank, load_pdb_xyz, relax_reference_model, structure_report

ALL = os.path.join(ROOT, "tests", "golden", "all45")


def load(cid):
    z = np.load(f"{ALL}/{cid}_upper.npz"); n = int(z["n"]); m = np.zeros((n, n)); iu = np.triu_indices(n)
    m[iu] = z["upper"]; m.T[iu] = z["upper"]; return m


def key(c):
    a, b = re.match(r"chr(\d+)_(\w+)", c).groups(); return (b, int(a))


def all_cids():
    """the 45 matrices the reference ships (the chr2_500kb stand-in, tools/make_chr2_standin.py, is no parity evidence)"""
    return sorted({os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(f"{ALL}/*_upper.npz") if "standin" not in np.load(p).files}, key=key)


SCHED = dict(cool_mult=1, hot_mult=1)      # experiment knobs: the deck's step counts (chromosome3D.pl:1095-1097, :1741-1742) times these


FINAL = {"minimiser": 1, "gtol": 0.0, "check_every": 250}     # override keys final_minimiser (0 = FIRE throughout, as rounds 1-4), gtol, check_every


def schedule(min_steps):
    st = default_schedule(min_steps)
    for k in range(len(st)):
        if st[k].kind == 1:
            st[k].nsteps = int(st[k].nsteps * SCHED["cool_mult"])
        elif st[k].kind == 0:
            st[k].nsteps = int(st[k].nsteps * SCHED["hot_mult"])
    return st


def solve(s, IF, over, nrep=20, seed=82364, min_steps=3000, embed=0, start=0):
    s.set_option("start", start)
    s.set_option("final_minimiser", FINAL["minimiser"])
    s.set_model(default_model(**over))
    d10 = pipeline.IF2dist_new(s, IF)
    s.set_schedule(schedule(min_steps), default_fire(), FINAL["gtol"], FINAL["check_every"])
    s.init_replicas(nrep, seed, 0)
    if embed:
        s.embed(50)
    s.run()
    return s.coords(), s.energies(), pipeline.restraints_from_dist10(d10)


HEADER = ("| matrix | N | R | rho best | rho rank-matched (rank) | rho mean | rho max | rho reference | d best | d matched | d max | ref pct | "
          "dist-Spearman best / matched / own | dRMSD best / own | bond ours | bond ref | i+2 ours | i+2 ref | Rg ours/ref | ratio | "
          "satisfied ours / ref (%) | deviation sum ours / ref | ms | bundled relaxed under our energy: int(E_noe) (ours best .. worst) | its rank in our 20 | file rank | moved: dist-Spearman / dRMSD | d max top10 |\n"
          "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")


def row(cid, n, R, rep, rank, ms):
    c, cr = rep["chain"], rep["chain_ref"]
    return ("| %-12s | %4d | %6d | %.4f | %.4f (%d) | %.4f | %.4f | %.4f | %+.4f | %+.4f | %+.4f | %.2f | %.3f / %.3f / %.3f | %.2f / %.2f | "
            "%.2f±%.2f | %.2f±%.2f | %.2f±%.2f | %.2f±%.2f | %.1f / %.1f | %.3f | %.1f / %.1f | %.4g / %.4g | %.0f | %d (%d .. %d) | %d | %d | %.4f / %.2f | %+.4f |" % (
                cid, n, R, rep["rho_best"], rep["rho_matched"], rank, rep["rho_mean"], rep["rho"].max(), rep["rho_ref"], rep["delta"], rep["delta_matched"],
                rep["delta_max"], rep["ref_percentile"], rep["sim_best"][0], rep["sim_matched"][0], rep["sim_own"][0], rep["sim_best"][1], rep["sim_own"][1],
                c[0], c[1], cr[0], cr[1], c[2], c[3], cr[2], cr[3], c[4], cr[4], rep["rg_ratio"],
                100.0 * rep["assess"]["best"][0] / R, 100.0 * rep["assess"]["ref"][0] / R, rep["assess"]["best"][1], rep["assess"]["ref"][1], ms,
                int(rep["relaxed"]["e_noe"]), int(rep["e_sorted"][0]), int(rep["e_sorted"][-1]), rep["relaxed"]["rank_in_ours"], rank,
                rep["relaxed"]["moved"][0], rep["relaxed"]["moved"][1], rep["delta_max_top10"]))


def summary(reps):
    d = np.array([r["delta"] for r in reps]); dm = np.array([r["delta_matched"] for r in reps])
    rg = np.array([r["rg_ratio"] for r in reps]); sb = np.array([r["sim_best"][0] for r in reps]); so = np.array([r["sim_own"][0] for r in reps])
    bsd = np.array([r["chain"][1] - r["chain_ref"][1] for r in reps]); a2 = np.array([r["chain"][2] - r["chain_ref"][2] for r in reps])
    a2s = np.array([r["chain"][3] - r["chain_ref"][3] for r in reps]); q05 = np.array([r["chain"][5] - r["chain_ref"][5] for r in reps]); q95 = np.array([r["chain"][6] - r["chain_ref"][6] for r in reps])
    dx = np.array([r["delta_max"] for r in reps]); dc = np.array([r["delta_closest"] for r in reps])
    pct = np.array([r["ref_percentile"] for r in reps])
    sat = np.array([r["assess"]["best"][0] / r["assess"]["ref"][0] for r in reps]); dev = np.array([r["assess"]["best"][1] / r["assess"]["ref"][1] for r in reps])
    rk = np.array([r["relaxed"]["rank_in_ours"] for r in reps]); fr = np.array([r["file_rank"] for r in reps]); dt = np.array([r["delta_max_top10"] for r in reps])
    from scipy.stats import spearmanr
    rank_line = (f"# energy ranks (the rankNN of the bundled file names, chromosome3D.pl:796-802): rank of the relaxed bundled model among our 20 vs file rank: "
                 f"Spearman {spearmanr(rk, fr)[0]:+.3f}, |ours - file| <= 3: {(np.abs(rk - fr) <= 3).sum()}, <= 5: {(np.abs(rk - fr) <= 5).sum()}, "
                 f"ours <= 10 (the file ranks are all <= 10): {(rk <= 10).sum()}, ours = 21 (above all of ours): {(rk == 21).sum()}, ours = 1: {(rk == 1).sum()}, median ours {np.median(rk):.0f} / file {np.median(fr):.0f}; "
                 f"order statistics of the pick: ref pct = 1 on {(pct == 1).sum()} rows, = 0 on {(pct == 0).sum()} "
                 f"(expected of {len(pct)}: best of 20 -> {len(pct) * 0.5:.1f} / 0, best of the 10 lowest-energy -> {len(pct) / 3.0:.1f} / 0, random member -> {len(pct) / 21.0:.1f} / {len(pct) / 21.0:.1f}); "
                 f"best Spearman of OUR 10 lowest-energy replicas vs bundled: within 0.01: {(np.abs(dt) <= 0.01).sum()}, mean |d| {np.abs(dt).mean():.4f}, bias {dt.mean():+.4f}\n")
    return rank_line + (f"# {len(d)} matrices: |d best| mean {np.abs(d).mean():.4f} median {np.median(np.abs(d)):.4f} max {np.abs(d).max():.4f}, "
            f"within 0.01: {(np.abs(d) <= 0.01).sum()}, 0.02: {(np.abs(d) <= 0.02).sum()}, 0.03: {(np.abs(d) <= 0.03).sum()}, bias {d.mean():+.4f}; "
            f"rank-matched: mean {np.abs(dm).mean():.4f}, within 0.01: {(np.abs(dm) <= 0.01).sum()}, bias {dm.mean():+.4f}; "
            f"BEST-SPEARMAN replica (the bundled model is not the reference's energy-best: ranks 1..10): mean {np.abs(dx).mean():.4f} max {np.abs(dx).max():.4f}, "
            f"within 0.01: {(np.abs(dx) <= 0.01).sum()}, bias {dx.mean():+.4f}; some replica within 0.01 of the reference: {(np.abs(dc) <= 0.01).sum()}; "
            f"reference inside our replica range (0 < pct < 1): {((pct > 0) & (pct < 1)).sum()}; "
            f"dist-Spearman ours vs bundled: mean {sb.mean():.4f} min {sb.min():.4f} (ours vs ours: mean {so.mean():.4f} min {so.min():.4f}); "
            f"Rg ratio mean {rg.mean():.3f} range {rg.min():.3f}-{rg.max():.3f}, within 2%: {(np.abs(rg - 1) <= 0.02).sum()}; "
            f"bond sd ours-ref mean {bsd.mean():+.3f}; i+2 mean ours-ref {a2.mean():+.3f}, i+2 sd ours-ref mean {a2s.mean():+.3f} (within 0.25: {(np.abs(a2s) <= 0.25).sum()}), "
            f"i+2 5% / 95% quantile ours-ref {q05.mean():+.2f} / {q95.mean():+.2f}; "
            f"reference's assessment, ours / bundled: satisfied restraints ratio mean {sat.mean():.3f} range {sat.min():.3f}-{sat.max():.3f}, "
            f"deviation sum ratio mean {dev.mean():.3f} range {dev.min():.3f}-{dev.max():.3f}")


def main():
    over = json.loads(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].startswith("{") else {}
    nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    subset = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    min_steps = int(over.pop("min_steps", 3000)); quiet = over.pop("quiet", 0); embed = over.pop("embed", 0)
    FINAL["minimiser"] = int(over.pop("final_minimiser", 1)); FINAL["gtol"] = float(over.pop("gtol", 0.0)); FINAL["check_every"] = int(over.pop("check_every", 250))
    seed = int(over.pop("seed", 82364)); dump = over.pop("dump", None); start = int(over.pop("start", 0))
    SCHED["cool_mult"] = float(over.pop("cool_mult", 1)); SCHED["hot_mult"] = float(over.pop("hot_mult", 1))
    s = Solver(0)
    reps, t_all, store = [], time.time(), {}
    if not quiet:
        print(HEADER)
    for cid in all_cids():
        if subset and not subset.search(cid):
            continue
        ref = glob.glob(f"{ALL}/{cid}_rank*_a11.pdb")
        IF = load(cid)
        if not ref:
            continue
        Xr = load_pdb_xyz(ref[0])
        if len(Xr) != IF.shape[0]:
            continue
        x, e, rows = solve(s, IF, over, nrep, seed, min_steps, embed, start)
        ms = s.last_timing()[0]
        rank = bundled_rank(ref[0])
        rep = structure_report(IF, x, e[:, 0], Xr, rank, rows)
        rep["relaxed"] = relax_reference_model(s, Xr, e[:, 0])
        rep["e_sorted"] = np.sort(e[:, 0].astype(np.int64)); rep["file_rank"] = rank
        rep["cid"] = cid
        reps.append(rep)
        if dump:
            store[cid] = x; store[cid + "_e"] = e
        if not quiet:
            print(row(cid, IF.shape[0], s.num_restraints, rep, rank, ms), flush=True)
    print(summary(reps) + f"; overrides {over}; schedule multipliers {SCHED}; final stage {FINAL}; start {'DG embed' if embed else ('extended strand' if start else 'random coil')}; seed {seed}; total {time.time() - t_all:.1f} s", flush=True)
    if dump:
        np.savez_compressed(dump, **store)


if __name__ == "__main__":
    main()
