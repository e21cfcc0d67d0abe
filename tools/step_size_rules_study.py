"""CPU-only design study (round 6; test infrastructure: it runs the restatement under oracle/): step-size rules for the two-point minimiser
that need only the four sums the kernels already carry (s.y, F.F, y.y, s.s), all with the length one evaluation late — alternating
BB1/BB2 (what ships), BB1, BB2, adaptive (ABB) and min-of-recent (ABBmin) variants, the geometric mean.  Start: the restatement's own
post-cooling coordinates.   python tools/step_size_rules_study.py [matrices,comma,separated] [replicas]   -> profiles/r06_step_size_rules_study.txt"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from multiprocessing import Pool
from chromosome3d_amd.solver import default_fire, default_model, default_schedule
from oracle import oracle as O
from tests.util import load_if, oracle_fire_from, oracle_model_from

GT = 1e-2
MAXE = 3000

def rms(g): return float(np.sqrt((g * g).mean()))

def run_rule(fg, x, rule, max_move=0.5, a0=None):
    _, g = fg(x); ne = 1
    a = a0; a_next = a; k = 0
    hist2 = []      # recent BB2 values
    while rms(g) >= GT and ne < MAXE:
        step = (-a * g).reshape(-1, 3)
        ln = np.sqrt((step ** 2).sum(axis=1)); step *= np.minimum(1.0, max_move / np.maximum(ln, 1e-30))[:, None]
        xn = x + step.ravel(); _, gn = fg(xn); ne += 1
        s_ = xn - x; y_ = gn - g; sy = s_.dot(y_); ss = s_.dot(s_); yy = y_.dot(y_)
        a = a_next
        if sy > 0:
            bb1, bb2 = ss / sy, sy / yy
            hist2.append(bb2); hist2[:] = hist2[-3:]
            if rule == "alt": a_next = bb1 if k % 2 == 0 else bb2
            elif rule == "bb1": a_next = bb1
            elif rule == "bb2": a_next = bb2
            elif rule.startswith("abb"):        # abb<kappa>: BB2 when BB2/BB1 < kappa else BB1
                kap = float(rule[3:]); a_next = bb2 if bb2 / bb1 < kap else bb1
            elif rule.startswith("abbmin"):
                pass
            elif rule.startswith("amin"):       # ABBmin (Frassoldati et al. 2008): min of the last BB2s when BB2/BB1 < tau else BB1
                tau = float(rule[4:]); a_next = min(hist2) if bb2 / bb1 < tau else bb1
            elif rule == "geo": a_next = np.sqrt(bb1 * bb2)
        else:
            a_next = a * 2.0
        a_next = min(max(a_next, 1e-7), 1e2)
        x, g = xn, gn; k += 1
    return ne

def work(arg):
    cid, r = arg
    IF = load_if(cid); n = IF.shape[0]
    m = default_model(); fire = default_fire()
    d10 = O.if_to_dist10(IF)
    rows = [(t.kind, t.nsteps, t.dt, t.w_all, t.w_vdw, t.repel_s, t.t_bath) for t in default_schedule(3000)]
    om, of = oracle_model_from(m, n), oracle_fire_from(fire)
    x, v, ev = O.run_schedule(om, d10, O.make_stages(rows[:-1]), of, 82364, r)
    w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]
    def fg(u):
        F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
        return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
    a0 = fire.dt_start * fire.dt_start * 418.4 / m.mass
    out = {}
    for rule in RULES:
        out[rule] = run_rule(fg, x.ravel().copy(), rule, a0=a0)
    return cid, n, r, out

RULES = ["alt", "bb1", "bb2", "abb0.3", "abb0.5", "abb0.8", "amin0.5", "amin0.8", "amin0.9", "geo"]
if __name__ == "__main__":
    cids = sys.argv[1].split(",") if len(sys.argv) > 1 else ["chr21_1mb", "chr13_1mb", "chr19_500kb", "chr4_1mb", "chr1_500kb"]
    nrep = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    jobs = [(c, r) for c in cids for r in range(nrep)]
    with Pool(8) as p:
        res = p.map(work, jobs)
    print("| matrix | N | r | " + " | ".join(RULES) + " |")
    tot = {k: 0 for k in RULES}; worst = {k: 0 for k in RULES}
    for cid, n, r, out in res:
        print(f"| {cid} | {n} | {r} | " + " | ".join(str(out[k]) for k in RULES) + " |")
        for k in RULES: tot[k] += out[k]; worst[k] = max(worst[k], out[k])
    print("totals", tot); print("worst", worst)
