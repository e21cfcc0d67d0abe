"""North-star acceptance (BASELINE.json): Spearman(IF, 1/d) of our models within +-0.01 of the bundled reference model of the same
chromosome (spearman_IF_pdb.pl:42-70 on output_models/*_a11.pdb), 20 replicas, the full default schedule, through the C ABI — and the
STRUCTURES themselves against the bundled models: the distance-matrix Spearman / scaled dRMSD of output_models/similarity.txt
(c3d_model_similarity) between our model and the bundled one, the place of the reference's Spearman inside our 20-replica distribution,
the chain envelope of SURVEY 8a/8c (bond mean/sd, |i-j| = 2 mean AND sd, radius of gyration) and the reference's own assessment numbers.
Everything is recomputed here from the committed fixtures (tests/golden/all45: exact IF matrices + the reference's model files) with the
pinned host scorers.

Where the model parameters come from (round 4): the RELAXATION criterion — every bundled model is a minimum of the
energy CNS minimised, so under a good model it stays put when minimised again — fitted on the 23 matrices at 1 Mb only
(tools/calib/relax_fit.py, host only, profiles/r04_relax_fit.txt); the 22 matrices at 500 kb are never seen by the fit and are asserted
separately below.  No annealing result enters the fit any more (round 3 fitted the annealed best-energy Spearman to the bundled value).

Which of our 20 replicas to compare: the bundled model of a chromosome is ONE of the reference's 20, and NOT its energy-best — the file
names carry ranks 1..10 (chr22_1mb_rank08, chr4_1mb_rank10, ...), never 11..20: it was picked among the ten lowest-energy models, by all
appearance for its Spearman (spearman_IF_pdb.pl:73-76 prints the models sorted by it; order statistics: test below).  THE GATE is the
literal north star:
  * best-ENERGY replica: 43 of 45 within +-0.01 (8 seeds, final stage as shipped since round 5: profiles/r05_seed_robustness_all45.md; round 4: 43 seven times, 42 once, r04_seed_robustness_all45.md); the two outside
    are named below, STRICT xfails with hard bounds — on both our energy prefers another fold than the bundled one (by 0.75 % of E_noe on
    chr7_1mb, 5 % on chr22_1mb).  Round 5 asked the reference's own energy ranks (the rankNN of the file names): relaxed under our energy the
    bundled chr22_1mb (file rank 8) is 19th of our 20, chr7_1mb (rank 2) 11th — the model gap is real, not an artefact of which replica is
    compared; chr22_1mb's -0.034 (round 3: -0.019) is a known loss of round 4's lower side, recorded in DESIGN.md section 2 item 4.
  * information only (selected on the metric under test, hence no gate): best Spearman of our 20 — 45 of 45 — and of our ten lowest-energy
    replicas — 44 of 45, bias -0.0003."""
import glob
import os
import re

import numpy as np
import pytest

from tests.util import GOLD, bundled_rank, load_pdb_xyz, relax_reference_model, structure_report

pytestmark = pytest.mark.gpu
ALL = os.path.join(GOLD, "all45")
TOL = 0.01
# Outside +-0.01 with their best-ENERGY replica, for every seed tried (sd over 8 seeds 0.0006 and 0.0000): deterministic fold preference.
#   chr22_1mb -0.034  N = 35.  Bead 0 carries thirteen 34-45 A targets; our energy-best fold gives it room (those pairs 12 A inside their
#                     targets, beads 2-5 eight A off beads 13-15), the bundled fold keeps 2-5 on 13-15 and bead 0 twenty A inside.  Relaxed
#                     under our energy the bundled model stays where it is (distance-Spearman 0.999 to itself, Spearman 0.7391 against
#                     0.7393) at E_noe 44 334 against our best fold's 42 060; the bundled model is the reference's RANK 8: seven of its own
#                     models had a lower CNS energy too.  Our best-Spearman replica: -0.0002.
#   chr7_1mb  -0.017  two folds; the bundled one (rank 2) relaxes to E_noe 548 732, ours to 544 644 (0.75 % lower); best-Spearman replica +0.002.
EDGE = {"chr22_1mb": 0.04, "chr7_1mb": 0.025}


def _load(cid):
    z = np.load(os.path.join(ALL, f"{cid}_upper.npz"))
    n = int(z["n"])
    m = np.zeros((n, n))
    iu = np.triu_indices(n)
    m[iu] = z["upper"]
    m.T[iu] = z["upper"]
    return m


def _key(c):
    a, b = re.match(r"chr(\d+)_(\w+)", c).groups()
    return (b, int(a))


# the 45 matrices the reference ships; chr2_500kb_upper.npz is a stand-in for the one it does not (tools/make_chr2_standin.py)
CIDS = sorted({os.path.basename(p)[:-len("_upper.npz")] for p in glob.glob(os.path.join(ALL, "*_upper.npz"))
               if "standin" not in np.load(p).files}, key=_key)


def _solve(solver, cid, nrep=20, seed=82364):
    from chromosome3d_amd import default_fire, default_model, default_schedule, pipeline
    IF = _load(cid)
    ref = glob.glob(os.path.join(ALL, f"{cid}_rank*_a11.pdb"))
    assert ref, cid
    Xr = load_pdb_xyz(ref[0])
    assert len(Xr) == IF.shape[0]
    solver.set_model(default_model())
    d10 = pipeline.IF2dist_new(solver, IF)
    solver.set_schedule(default_schedule(3000), default_fire(), 0.0, 250)
    solver.init_replicas(nrep, seed, 0)
    solver.run()
    x, e3 = solver.coords(), solver.energies()
    e = e3[:, 0]
    rep = structure_report(IF, x, e, Xr, bundled_rank(ref[0]), pipeline.restraints_from_dist10(d10))
    rep["x_best"] = x[rep["best"]].astype(np.float64)
    rep["file_rank"] = bundled_rank(ref[0])
    rep["e_total"] = e3.sum(axis=1)               # what the anneal and the minimiser descend on (final stage: all weights 1)
    rep["e_chain_repel"] = e3[:, 1:].sum(axis=1)
    rep["relaxed"] = relax_reference_model(solver, Xr, e)          # replaces the solver's replicas: last
    return rep


_CACHE = {}


def _report(solver, cid):
    if cid not in _CACHE:
        _CACHE[cid] = _solve(solver, cid)
    return _CACHE[cid]


@pytest.mark.parametrize("cid", [pytest.param(c, marks=pytest.mark.xfail(reason="named fold-preference outlier, bounded below", strict=True)) if c in EDGE else c
                                 for c in CIDS])
def test_best_ranked_replica_within_the_north_star_tolerance(solver, cid):
    """+-0.01 for every bundled matrix with the best-ENERGY replica; BASELINE configs 2 and 3 (chr21_1mb x 20, chr1_500kb x 20) are two of
    them.  The two named outliers are STRICT xfails: an unexpected pass is reported as loudly as a regression."""
    r = _report(solver, cid)
    assert len(r["rho"]) == 20 and np.isfinite(r["rho"]).all()
    assert abs(r["delta"]) <= TOL, (cid, r["delta"])


def test_information_only_best_spearman_replica(solver):
    """INFORMATION, not the acceptance gate (the gate is the best-ENERGY reading above, two named strict xfails, hard bounds in
    test_all_45_bundled_matrices_spearman): picking our replica by the metric under test is biased towards passing.  It is reported
    because the bundled file was itself picked from the reference's run (ranks 1..10 in the names, never 11..20): the best Spearman among
    ALL 20 of ours, and — the reading the file names support — among our TEN lowest-energy replicas, against the bundled model's."""
    reps = {cid: _report(solver, cid) for cid in CIDS}
    dx = np.array([r["delta_max"] for r in reps.values()])
    dt = np.array([r["delta_max_top10"] for r in reps.values()])
    dc = np.array([r["delta_closest"] for r in reps.values()])
    print("best Spearman of 20: within 0.01 %d, bias %+.4f; of the 10 lowest-energy: within 0.01 %d, bias %+.4f; closest: %d" %
          ((np.abs(dx) <= TOL).sum(), dx.mean(), (np.abs(dt) <= TOL).sum(), dt.mean(), (np.abs(dc) <= TOL).sum()))
    assert (np.abs(dx) <= TOL).sum() >= 44 and (np.abs(dt) <= TOL).sum() >= 42 and (np.abs(dc) <= TOL).sum() >= 44


def test_headline_config_margin(solver):
    """chr1_500kb x 20, the configuration the metric is quoted on: |delta| <= 0.005 for the best-energy, the rank-matched (the bundled
    file is the reference's rank 3) AND the best-Spearman replica (measured +0.0013 / +0.0009 / +0.0015), the reference's value inside
    our replica distribution, the same structure (distance-Spearman 0.985), the same (i,i+2) spread (2.22 against 2.27 A)."""
    r = _report(solver, "chr1_500kb")
    assert abs(r["delta"]) <= 0.005 and abs(r["delta_matched"]) <= 0.005 and abs(r["delta_max"]) <= 0.005, (r["delta"], r["delta_matched"], r["delta_max"])
    assert 0.0 < r["ref_percentile"] < 1.0
    assert r["sim_best"][0] >= 0.975 and abs(r["rg_ratio"] - 1.0) <= 0.01 and abs(r["chain"][3] - r["chain_ref"][3]) <= 0.15


def test_all_45_bundled_matrices_spearman(solver):
    """Every bundled matrix (both resolutions), 20 replicas each, ~1.5 s of annealing on an MI355X for all of them."""
    assert len(CIDS) == 45
    reps = {cid: _report(solver, cid) for cid in CIDS}
    d = {c: r["delta"] for c, r in reps.items()}
    a = np.abs(np.array(list(d.values())))
    bad = {c: round(float(v), 4) for c, v in d.items() if abs(v) > EDGE.get(c, TOL)}
    assert not bad, bad
    assert (a <= 0.01).sum() >= 43 and np.sort(a)[-3] <= 0.01 and a.mean() <= 0.004 and np.median(a) <= 0.0028, \
        ((a <= 0.01).sum(), a.max(), a.mean(), np.median(a))
    assert abs(np.mean(list(d.values()))) <= 0.0025                      # measured -0.0017, of which -0.0011 are the two named outliers
    # the replica that has the bundled model's RANK in our run: which replica that is changes with the last bit of the arithmetic, and
    # on the chromosomes with several folds (chr7_1mb, chr13_1mb, chr22_1mb) an arbitrary replica sits up to 0.035 from the best one
    dm = np.abs(np.array([r["delta_matched"] for r in reps.values()]))
    assert (dm <= 0.01).sum() >= 40 and dm.max() <= 0.04, ((dm <= 0.01).sum(), dm.max())
    # like for like: the best Spearman of our 20 (measured: 45 within 0.01, max 0.0081, bias +0.0009)
    dx = np.array([r["delta_max"] for r in reps.values()])
    assert (np.abs(dx) <= 0.01).sum() == 45 and abs(dx.mean()) <= 0.002 and np.abs(dx).mean() <= 0.003, ((np.abs(dx) <= 0.01).sum(), dx.mean(), np.abs(dx).mean())
    # the reference's value is not an outlier of our own 20 replicas for most chromosomes (measured 32; round 3: 30)
    pct = np.array([r["ref_percentile"] for r in reps.values()])
    assert ((pct > 0) & (pct < 1)).sum() >= 28, ((pct > 0) & (pct < 1)).sum()
    # the 22 matrices at 500 kb were NOT used by the fit (tools/calib/relax_fit.py trains on 1 Mb only): 22 of 22, mean 0.0022
    held = np.abs(np.array([v for c, v in d.items() if c.endswith("_500kb")]))
    assert len(held) == 22 and (held <= 0.01).all() and held.mean() <= 0.003, (held.max(), held.mean())


def test_all_45_bundled_matrices_structure(solver):
    """Same structure, not only the same scalar: our model against the bundled model of the same chromosome.  Yardstick for
    the distance-matrix Spearman: the reference's own agreement between its 500 kb and 1 Mb models of one chromosome,
    0.855-0.967 (output_models/similarity.txt:1-75), and the agreement between two of OUR replicas (~0.98)."""
    reps = {cid: _report(solver, cid) for cid in CIDS}
    sim = np.array([r["sim_best"][0] for r in reps.values()])
    simm = np.array([r["sim_matched"][0] for r in reps.values()])
    own = np.array([r["sim_own"][0] for r in reps.values()])
    assert sim.min() >= 0.855 and simm.min() >= 0.85 and sim.mean() >= 0.968 and (sim >= 0.93).sum() >= 40, (sim.min(), simm.min(), sim.mean(), (sim >= 0.93).sum())
    assert own.mean() - sim.mean() <= 0.015                              # ours-vs-reference is within 0.015 of ours-vs-ours
    drm = np.array([r["sim_best"][1] for r in reps.values()])
    assert drm.max() <= 3.5 and drm.mean() <= 1.75, (drm.max(), drm.mean())
    # chain envelope.  Radius of gyration: ALL 45 within 1 % (measured 0.992-1.008; round 3: 39 within 2 %)
    rg = np.array([r["rg_ratio"] for r in reps.values()])
    assert np.abs(rg - 1).max() <= 0.015 and abs(rg.mean() - 1) <= 0.005, (rg.min(), rg.max(), rg.mean())
    ch = np.array([r["chain"] for r in reps.values()])
    cr = np.array([r["chain_ref"] for r in reps.values()])
    assert np.abs(ch[:, 0] - cr[:, 0]).max() <= 0.13 and np.abs((ch[:, 0] - cr[:, 0]).mean()) <= 0.06       # bond mean (measured +0.04)
    assert np.abs(ch[:, 1] - cr[:, 1]).max() <= 0.12 and np.abs((ch[:, 1] - cr[:, 1]).mean()) <= 0.03       # bond sd
    assert np.abs(ch[:, 2] - cr[:, 2]).max() <= 0.3 and np.abs((ch[:, 2] - cr[:, 2]).mean()) <= 0.08        # |i-j| = 2 mean (round 3: 0.6 / 0.2)
    # |i-j| = 2 SPREAD: round 3's harmonic k_ang 43 made every row 0.3-0.45 A too narrow; now within 0.25 on all 45 rows, mean -0.06
    assert (np.abs(ch[:, 3] - cr[:, 3]) <= 0.25).sum() >= 43 and np.abs(ch[:, 3] - cr[:, 3]).max() <= 0.3 and abs((ch[:, 3] - cr[:, 3]).mean()) <= 0.1
    assert abs((ch[:, 6] - cr[:, 6]).mean()) <= 0.15                                                          # its 95 % quantile
    assert (3.6 <= ch[:, 0]).all() and (ch[:, 0] <= 4.3).all() and (11.0 <= ch[:, 4]).all() and (ch[:, 4] <= 19.5).all()


def test_all_45_the_references_own_assessment_of_our_models(solver):
    """The two numbers the reference prints for every model it builds — restraints satisfied within the relaxation and the summed
    violation (assess_dgsa, chromosome3D.pl:447-485, :581-600; c3d_assess is pinned to its known answers 68/528, 2955.67 and
    10778/101426, 374370.87 in test_output_side / test_gpu_parity) — of OUR best-ranked model against the bundled model, on the same
    contact.tbl rows.  chr1_500kb: 10.5 % / 10.6 % satisfied, 372 700 / 374 371 summed violation.  Table: profiles/r05_parity_sweep_all45.md
    (the shipped model and schedule; the round-4 table of the same name prefix held FIRE's final stage)."""
    reps = {cid: _report(solver, cid) for cid in CIDS}
    sat = np.array([r["assess"]["best"][0] / r["assess"]["ref"][0] for r in reps.values()])
    dev = np.array([r["assess"]["best"][1] / r["assess"]["ref"][1] for r in reps.values()])
    assert 0.94 <= dev.min() and dev.max() <= 1.06 and abs(dev.mean() - 1.0) <= 0.015, (dev.min(), dev.max(), dev.mean())
    # measured: summed violation ALL 45 within 5 % (0.953-1.046; round 3: 35), 40 within 3 %; satisfied counts: 34 within 5 %, 40 within 10 %
    assert (np.abs(dev - 1.0) <= 0.05).sum() >= 44 and (np.abs(dev - 1.0) <= 0.03).sum() >= 37, ((np.abs(dev - 1.0) <= 0.05).sum(), (np.abs(dev - 1.0) <= 0.03).sum())
    assert (np.abs(sat - 1.0) <= 0.10).sum() >= 38, (np.abs(sat - 1.0) <= 0.10).sum()
    assert 0.70 <= sat.min() and sat.max() <= 1.15 and abs(sat.mean() - 1.0) <= 0.04, (sat.min(), sat.max(), sat.mean())
    ours = reps["chr1_500kb"]["assess"]
    assert ours["R"] == 101426 and ours["ref"][0] == 10778 and abs(ours["ref"][1] - 374370.87) < 0.5       # the reference's own line for its model
    assert abs(ours["best"][1] / ours["ref"][1] - 1.0) <= 0.02 and abs(ours["best"][0] / ours["ref"][0] - 1.0) <= 0.05


def test_reference_energy_ranks_against_ours(solver):
    """The one reference-held datum on ENERGY ORDERING: every bundled file name carries the model's rank by CNS NOE energy in the
    reference's own run of 20 (chr22_1mb_rank08, chr4_1mb_rank10, ...; rule chromosome3D.pl:796-802, 822-828), all of them 1..10.  Each
    bundled model is relaxed under OUR energy (final minimisation stage only, from the bundled coordinates) and its int(E_noe) ranked among
    the int(E_noe) of our 20 annealed replicas.  If our energy orders folds as CNS's does, the relaxed bundled model lands where the file
    name says, within the scatter of two independent runs of 20.  Table: profiles/r05_parity_sweep_all45.md (last columns)."""
    from scipy.stats import spearmanr
    reps = {cid: _report(solver, cid) for cid in CIDS}
    rk = np.array([r["relaxed"]["rank_in_ours"] for r in reps.values()])
    fr = np.array([r["file_rank"] for r in reps.values()])
    moved = np.array([r["relaxed"]["moved"][0] for r in reps.values()])
    gap = np.array([r["relaxed"]["rel_gap"] for r in reps.values()])
    print("relaxed bundled model: rank in ours", dict(zip(CIDS, rk.tolist())), "file", fr.tolist(), "rel gap to our best", np.round(gap, 4).tolist())
    for c in EDGE:
        print(c, "file rank", reps[c]["file_rank"], "-> rank in ours", reps[c]["relaxed"]["rank_in_ours"], "E_noe", int(reps[c]["relaxed"]["e_noe"]),
              "moved", reps[c]["relaxed"]["moved"])
    # the bundled model IS a minimum of our energy: relaxing it does not change the structure
    assert moved.min() >= 0.98 and moved.mean() >= 0.99, (moved.min(), moved.mean())        # measured 0.9868 / 0.9943
    assert ((rk >= 1) & (rk <= 21)).all()
    # Measured (seed 82364, profiles/r05_parity_sweep_all45.md): the relaxed bundled model sits among our TEN lowest energies on 35 of 45
    # rows (every file rank is <= 10), within 5 places of its file rank on 29, median place 4 (file: 4), above all 20 of ours on ONE row
    # (chr16_1mb) — and BELOW all 20 of ours on 11 rows: there the reference's search found a deeper minimum of OUR energy than our 20
    # anneals did (chr13_1mb by 3.4 %).  The rank correlation itself is weak (+0.10): our energy orders folds as CNS's does only coarsely.
    # The two north-star outliers are exactly rows where it does not: chr22_1mb file rank 8 -> 19th of ours, chr7_1mb 2 -> 11th.
    assert (rk <= 10).sum() >= 31 and (np.abs(rk - fr) <= 5).sum() >= 25 and (rk == 21).sum() <= 3 and 2 <= np.median(rk) <= 7, \
        ((rk <= 10).sum(), (np.abs(rk - fr) <= 5).sum(), (rk == 21).sum(), np.median(rk), spearmanr(rk, fr)[0])
    assert gap.min() >= -0.05 and gap.max() <= 0.08, (gap.min(), gap.max())   # measured -3.4 % (chr13_1mb) .. +6.0 % (chr11_1mb) of our best E_noe
    for c in EDGE:                                                             # the named outliers: our energy ranks the bundled fold LOW
        assert reps[c]["relaxed"]["rank_in_ours"] > reps[c]["file_rank"] + 5, (c, reps[c]["relaxed"]["rank_in_ours"])


def test_bundled_models_by_total_energy_lie_inside_our_field(solver):
    """The reference ranks by NOE energy alone (chromosome3D.pl:796-802); by THAT figure the relaxed bundled model is below all 20 of
    ours on about a dozen rows (test_reference_energy_ranks_against_ours).  By the TOTAL energy — E_noe + bond/angle + repel at the final
    stage's weights, what our anneal and minimiser descend on — it is below all of ours on 2 of 45 (chr13_1mb by 2.5 %: the one real
    search gap) and takes the median place 8 of 21: the bundled models pay for NOE energy with chain and repel energy
    (profiles/r05_total_energy_ranks.md; measured 2 / 3 above all / median 8 / mean excess of chain + repel over our medians +3 000)."""
    rk_noe, rk_tot, excess = [], [], []
    for cid in CIDS:
        r = _report(solver, cid)
        tot = float(r["relaxed"]["e3"].sum())
        rk_noe.append(r["relaxed"]["rank_in_ours"]); rk_tot.append(int(1 + (r["e_total"] < tot).sum()))
        excess.append(float(r["relaxed"]["e3"][1:].sum() - np.median(r["e_chain_repel"])))
    rk_noe, rk_tot, excess = np.array(rk_noe), np.array(rk_tot), np.array(excess)
    assert (rk_tot == 1).sum() <= 4 < (rk_noe == 1).sum(), (rk_tot, rk_noe)
    assert (rk_tot == 21).sum() <= 6 and 5 <= np.median(rk_tot) <= 12, rk_tot
    assert np.median(rk_tot) > np.median(rk_noe) and (excess > 0).sum() >= 27, (np.median(rk_tot), np.median(rk_noe), (excess > 0).sum())


def test_order_statistics_of_how_the_bundled_model_was_picked(solver):
    """`ref_percentile` = fraction of our 20 replicas whose Spearman lies below the bundled model's.  If the bundled model were the best
    of 20 draws from OUR distribution it would exceed all 20 of ours on ~1/2 of the rows; the best of the TEN lowest-energy models (what the
    file names say: ranks 1..10 only) on ~1/3; a random member on ~1/21 — and it would sit below all of ours on ~0 / ~0 / ~1/21 of them."""
    reps = {cid: _report(solver, cid) for cid in CIDS}
    pct = np.array([r["ref_percentile"] for r in reps.values()])
    top, bottom = int((pct == 1).sum()), int((pct == 0).sum())
    print("ref pct = 1 on", top, "rows, = 0 on", bottom, "of", len(pct))
    # measured 13 / 2 of 45: what "best Spearman among the ten lowest-energy models" predicts (15 +- 3.2 / 0), 2.8 sd below "best of all
    # 20" (22.5 +- 3.4), far above "a random member" (2.1).  The two rows at 0 are chr5_1mb and chr23_500kb (both within +-0.005).
    assert 8 <= top <= 20 and bottom <= 4, (top, bottom)


def test_cross_resolution_agreement_of_our_own_models(solver):
    """SURVEY 8f-2, output_models/similarity.txt:1-75: the reference's 500 kb model, reduced to 1 Mb, against its 1 Mb model of the same
    chromosome — Spearman of the pair distances 0.855-0.967, mean 0.925 over the 17 chromosomes it lists.  The same for OUR best-ranked
    models of the shipped energy model (c3d_reduce_model + c3d_model_similarity, which reproduce that file to 1e-12 on the bundled
    models: tests/test_abi_host.py).  Table: profiles/r05_cross_resolution.md."""
    import json
    from chromosome3d_amd import pipeline
    ref = json.load(open(os.path.join(GOLD, "similarity_reference.json")))
    chrs = sorted({re.match(r"(chr\d+)_", k).group(1) for k in ref}, key=lambda c: int(c[3:]))
    assert len(chrs) == 17
    ours, theirs = [], []
    for c in chrs:
        a, b = _report(solver, f"{c}_500kb"), _report(solver, f"{c}_1mb")
        rho, _ = pipeline.model_similarity(pipeline.reduce_model(a["x_best"]), b["x_best"])
        ours.append(rho)
        theirs.append([v["spearman"] for k, v in ref.items() if k.startswith(c + "_")][0])
    ours, theirs = np.array(ours), np.array(theirs)
    print("cross-resolution Spearman ours", np.round(ours, 4).tolist(), "mean", ours.mean(), "min", ours.min(), "reference mean", theirs.mean(), "min", theirs.min())
    assert abs(theirs.mean() - 0.925) < 1e-3 and abs(theirs.min() - 0.855) < 1e-3
    assert ours.mean() >= 0.92 and ours.min() >= 0.85, (ours.mean(), ours.min())


def test_k1_bit_exact_on_all_45(solver):
    """IF -> target distances (tenths of an Angstrom) on the GPU against the CPU restatement of chromosome3D.pl:110-162 for every
    bundled matrix, and the restraint counts of BASELINE.md."""
    from chromosome3d_amd import default_model, pipeline
    from oracle import oracle as O
    solver.set_model(default_model())
    total = total_500kb = 0
    for cid in CIDS:
        IF = _load(cid)
        d10 = pipeline.IF2dist_new(solver, IF)
        assert np.array_equal(d10, O.if_to_dist10(IF)), cid
        total += solver.num_restraints
        total_500kb += solver.num_restraints if cid.endswith("_500kb") else 0
    # restrained pairs (|i-j| >= 5, IF > 0): SURVEY 8d gives 717 360 for the 22 matrices at 500 kb
    assert total_500kb == 717360 and total == 920419, (total_500kb, total)


def test_config4_same_models_whatever_the_rank_count(tmp_path):
    """BASELINE configs[3]: all 23 chromosomes at 500 kb x 20 replicas through the product entry (python -m
    chromosome3d_amd.batch), once in one process and once as two ranks (gloo rendezvous, both ranks on this box's GPU):
    the per-chromosome ranking and the truncated NOE energies are identical — what a chromosome yields does not depend
    on the rank that solved it (two processes sharing one GPU also exercise the abandoned-launch fallback)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    one = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--json"], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), "-m", "chromosome3d_amd.batch", "--json"], cwd=root,
                         env=dict(env, C3D_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    a = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    b = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    # 22 shipped matrices + the chr2_500kb stand-in (.MISSING_LARGE_BLOBS:1): the 23 jobs of test.sh:9-12, the largest included
    assert a["world"] == 1 and b["world"] == 2 and len(a["chromosomes"]) == 23 and a["standins"] == ["chr2_500kb"]
    assert a["chromosomes"] == b["chromosomes"]
    assert all(len(c["order"]) == 20 for c in a["chromosomes"].values())


def test_paired_small_anneals_do_not_change_a_model():
    """Config 4 anneals its small chromosomes two at a time, one on XCDs 0-3, one on XCDs 4-7 (two contexts; batch.solve_assigned, options
    cluster_xcd_count / cluster_xcd_base): `--pair 1` (default) and `--pair 0` give the same ranking and truncated energies for all 23."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    outs = []
    for pair in ("1", "0"):
        p = subprocess.run([sys.executable, "-m", "chromosome3d_amd.batch", "--json", "--pair", pair], cwd=root, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        outs.append(json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1]))
    assert len(outs[0]["chromosomes"]) == 23 and outs[0]["chromosomes"] == outs[1]["chromosomes"]


def test_xcd_set_of_a_launch_is_bitwise_neutral(solver):
    """A context's multi-step launches on 8, 4, 2 XCDs and on the upper half: replica r lives on XCD base + r % count, workgroups elsewhere
    exit at once; the planner picks another geometry per count, the trajectories keep their bits (same canonical sums)."""
    from chromosome3d_amd import default_fire, default_model, default_schedule, pipeline
    IF = _load("chr20_500kb")
    out = []
    try:
        for count, base in ((8, 0), (4, 0), (4, 4), (2, 6), (1, 3)):
            solver.set_option("cluster_xcd_base", 0)
            solver.set_option("cluster_xcd_count", count)
            solver.set_option("cluster_xcd_base", base)
            solver.set_model(default_model())
            pipeline.IF2dist_new(solver, IF)
            solver.set_schedule(default_schedule(600), default_fire(), 0.0, 250)
            solver.init_replicas(8, 82364, 0)
            solver.run()
            assert solver.stat("cluster_ok") == 1 and solver.stat("last_path") == 2 and solver.stat("cluster_xcd_count") == count, (count, base)
            out.append((solver.coords(), solver.energies()))
        assert solver.stat("resident_fallbacks") == 0 and solver.stat("cluster_placement_mismatches") == 0
        for x, e in out[1:]:
            assert np.array_equal(x, out[0][0]) and np.array_equal(e, out[0][1])
        from chromosome3d_amd import lib
        with pytest.raises(lib.C3DError):
            solver.set_option("cluster_xcd_count", 9)
        with pytest.raises(lib.C3DError):
            solver.set_option("cluster_xcd_base", 8)
    finally:
        solver.set_option("cluster_xcd_base", 0)
        solver.set_option("cluster_xcd_count", 8)


def test_final_minimisation_converges_at_the_headline_size(solver):
    """The FIRE stand-in for the reference's <= 150 000-evaluation L-BFGS stage (deck :1790-1803) reaches the gradient
    exit (max RMS force component < 1e-2 kcal/mol/A over all 20 replicas) well inside its 3000-step budget at N = 455
    (profiles/r05_fire_convergence.txt and r05_final_stage_convergence.txt, the shipped model: Spearman is settled to 1e-5 after ~500
    steps, the largest RMS force is below 1e-2 after 1000-2000)."""
    from chromosome3d_amd import default_fire, default_model, default_schedule, pipeline
    IF = _load("chr1_500kb")
    solver.set_model(default_model())
    pipeline.IF2dist_new(solver, IF)
    solver.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250)
    solver.init_replicas(20, 82364, 0)
    solver.run()
    L = solver.schedule_length
    ms, steps, launches = solver.last_timing()
    assert 2172 + 500 <= steps <= L - 250, (steps, L)         # left through the gradient exit, not by running out of steps
    assert solver.stat("rms_force") < 1e-2


def test_final_stage_leaves_through_the_exit_test_on_every_bundled_matrix(solver):
    """The shipped final stage (kind 5: two-point step sizes, FIRE after 1000 of them) with the product's exit test (RMS force < 1e-2,
    every 250 steps), 20 replicas, all 45 matrices: every anneal leaves through the test, well before the stage's 3000 steps are used
    up, and the two-point part alone ends most of them.  (FIRE throughout — rounds 1-4, option final_minimiser = 0 — used up all 3000
    steps on 7 of 135 anneals over three seeds; this stage on none: scratch scan of round 5, profiles/r05_final_minimiser_ab.md.)"""
    from chromosome3d_amd import default_fire, default_model, default_schedule, pipeline
    steps = {}
    for fm in (1, 0):
        solver.set_option("final_minimiser", fm)
        for cid in CIDS:
            solver.set_model(default_model())
            pipeline.IF2dist_new(solver, _load(cid))
            solver.set_schedule(default_schedule(3000), default_fire(), 1e-2, 250)
            solver.init_replicas(20, 82364, 0)
            solver.run()
            steps[(fm, cid)] = solver.last_timing()[1]
            if fm:
                assert solver.stat("rms_force") < 1e-2, cid
    solver.set_option("final_minimiser", 1)
    L = solver.schedule_length
    two = np.array([steps[(1, c)] for c in CIDS]) - 2172
    fire = np.array([steps[(0, c)] for c in CIDS]) - 2172
    assert L == 5172 and (two <= 2000).all(), dict(zip(CIDS, two))
    assert (two <= 1000).sum() >= 40 and two.sum() < 0.65 * fire.sum(), (two.sum(), fire.sum())      # measured: 0.52-0.55 of FIRE's steps


@pytest.mark.parametrize("kind", [2, 5])
@pytest.mark.parametrize("cid,nrep", [("chr21_1mb", 6), ("chr1_500kb", 4)])
def test_device_fire_stage_ends_where_lbfgs_ends(solver, cid, nrep, kind):
    """(kind 2: FIRE, the final stage of rounds 1-4; kind 5: the stage as shipped since round 5 — two-point step sizes, then FIRE.)
    The one hot-path stage whose ALGORITHM differs from the reference's by choice: `minimize lbfgs nstep=15000` x 10 (deck
    chromosome3D.pl:1790-1803) is FIRE here.  From the device's own post-cooling coordinates, the device's final stage (the shipped fp32
    kernels, gradient exit) against L-BFGS (scipy L-BFGS-B, 10 correction pairs, up to 10 restarts like the deck's, on the CPU
    restatement's fp64 energy and gradient).  profiles/r05_fire_vs_lbfgs.txt (32 replicas, N = 37 .. 455): in 20 of 32 — all 8 at
    N = 37 — the two end in the SAME minimum (energy to 1e-8, every pair distance to 0.004 A); in the others in NEIGHBOURING minima (a
    bead or two seated differently: single pair distances up to 4.6 A apart, the energy 5e-5 of itself apart on average at N = 455 and
    3.6e-3 at worst, L-BFGS's the lower one more often), Spearman(IF, 1/d) within 5e-4 in every case."""
    from scipy.optimize import minimize
    from chromosome3d_amd import default_fire, default_model, default_schedule, make_stages, pipeline
    from oracle import oracle as O
    from tests.util import oracle_model_from
    IF = _load(cid)
    n = IF.shape[0]
    rows = [(s.kind, s.nsteps, s.dt, s.w_all, s.w_vdw, s.repel_s, s.t_bath) for s in default_schedule(3000)]
    assert rows[-1][0] == 5
    rows[-1] = (kind,) + rows[-1][1:]
    m = default_model()
    solver.set_model(m)
    d10 = pipeline.IF2dist_new(solver, IF)
    solver.set_schedule(make_stages(rows[:-1]), default_fire(), 0.0, 250)
    solver.init_replicas(nrep, 82364, 0)
    solver.run()
    x0 = solver.coords()
    solver.set_schedule(make_stages(rows[-1:]), default_fire(), 1e-2, 250)
    solver.init_replicas(nrep, 82364, 0)
    solver.set_coords(x0)
    solver.run()
    xf = solver.coords().astype(np.float64)
    om = oracle_model_from(m, n)
    w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]

    def fg(u):
        F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
        return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
    i, j = np.triu_indices(n, 1)
    rel, drms, drho = [], [], []
    for r in range(nrep):
        u = x0[r].astype(np.float64).ravel()
        for cycle in range(10):                  # the deck restarts its minimiser ten times (:1800-1803); scipy's line search gives up now and then
            res = minimize(fg, u, jac=True, method="L-BFGS-B", options=dict(maxiter=15000, maxfun=150000, ftol=1e-15, gtol=1e-5, maxcor=10))
            u = res.x
            if res.success:
                break
        xl = res.x.reshape(n, 3)
        rel.append((fg(xf[r].ravel())[0] - res.fun) / abs(res.fun))
        dd = np.linalg.norm(xf[r][i] - xf[r][j], axis=1) - np.linalg.norm(xl[i] - xl[j], axis=1)
        drms.append(np.sqrt((dd ** 2).mean()))
        drho.append(abs(pipeline.spearman_IF_pdb(IF, xf[r].astype(np.float32)) - pipeline.spearman_IF_pdb(IF, xl.astype(np.float32))))
    rel, drms, drho = np.array(rel), np.array(drms), np.array(drho)
    print(cid, "(f FIRE - f L-BFGS) / f", rel, "dRMSD", drms, "dSpearman", drho)
    if n < 100:        # one minimum, the same coordinates
        assert np.abs(rel).max() <= 2e-6 and drms.max() < 0.05 and drho.max() < 2e-4, (rel, drms, drho)
    else:              # the same or a neighbouring minimum: energy, structure and Spearman agree
        assert np.abs(rel).max() <= 5e-3 and drms.max() < 0.8 and drho.max() < 2e-3, (rel, drms, drho)
        assert (np.abs(rel) < 1e-6).sum() >= 1 and np.median(np.abs(rel)) < 5e-4, rel      # measured: 1 of the first 4 identical, median 3e-5
