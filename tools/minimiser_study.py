"""How many force evaluations do candidate minimisers need for the final stage (deck chromosome3D.pl:1790-1803) — from the device's own
post-cooling coordinates down to the device's exit test, RMS force < 1e-2?  FIRE as shipped (on the device, the test every 10 steps)
against, on the CPU restatement's fp64 energy and gradient: L-BFGS (m = 5, the first trial step of an iteration 1, Armijo back-tracking;
counted in evaluations), Barzilai-Borwein steps (two dot products an iteration, no energy, no line search: the cheapest thing the
multi-step kernel's one float4 of replica sums could carry; also with the step length one iteration late, as that kernel's sums arrive), and Polak-Ribiere+ conjugate gradients with back-tracking.
    python tools/minimiser_study.py [replicas=4] [chromosomes ...]        (a design study for DESIGN.md 8; nothing here ships)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from chromosome3d_amd import Solver, default_fire, default_model, default_schedule, make_stages, pipeline
from oracle import oracle as O
from tests.util import oracle_model_from
from tools.parity_sweep import load as load_if

GT = 1e-2


def rms(g):
    return float(np.sqrt((g * g).mean()))


def lbfgs(fg, x, m=5, maxeval=6000, max_move=0.5):
    f, g = fg(x); ne = 1; S, Y = [], []
    while rms(g) >= GT and ne < maxeval:
        q = g.copy(); al = []
        for s_, y_ in zip(reversed(S), reversed(Y)):
            a = s_.dot(q) / y_.dot(s_); al.append(a); q -= a * y_
        q *= (S[-1].dot(Y[-1]) / Y[-1].dot(Y[-1])) if S else 1.0 / max(np.abs(g).max(), 1e-30) * max_move
        for (s_, y_), a in zip(zip(S, Y), reversed(al)):
            q += (a - y_.dot(q) / y_.dot(s_)) * s_
        d = -q; gd = g.dot(d)
        if gd >= 0:
            d = -g; gd = -g.dot(g); S, Y = [], []
        t = 1.0
        big = np.abs(d).max()
        if big * t > 4 * max_move: t = 4 * max_move / big          # no bead moves more than 2 A in one trial
        while True:
            fn, gn = fg(x + t * d); ne += 1
            if fn <= f + 1e-4 * t * gd or t < 1e-10 or ne >= maxeval: break
            t *= 0.5
        s_ = t * d; y_ = gn - g
        if s_.dot(y_) > 1e-12 * np.sqrt(s_.dot(s_) * y_.dot(y_)):
            S.append(s_); Y.append(y_)
            if len(S) > m: S.pop(0); Y.pop(0)
        x = x + s_; f, g = fn, gn
    return x, f, ne


def bb(fg, x, maxeval=6000, max_move=0.5, alt=True):
    """Barzilai-Borwein: x+ = x - a g, a = s.s / s.y (odd iterations) or s.y / y.y (even), no energy, a cap on the largest bead move."""
    _, g = fg(x); ne = 1; a = max_move / max(np.abs(g).max(), 1e-30) * 0.1; k = 0
    while rms(g) >= GT and ne < maxeval:
        step = -a * g
        big = np.sqrt((step.reshape(-1, 3) ** 2).sum(axis=1)).max()
        if big > max_move: step *= max_move / big
        xn = x + step; _, gn = fg(xn); ne += 1
        s_ = xn - x; y_ = gn - g; sy = s_.dot(y_)
        if sy > 0:
            a = (s_.dot(s_) / sy) if (not alt or k % 2 == 0) else (sy / y_.dot(y_))
        else:
            a = a * 2.0            # negative curvature along the step: go further, the cap holds it
        x, g = xn, gn; k += 1
    return x, fg(x)[0], ne


def bb_lag(fg, x, maxeval=6000, max_move=0.5):
    """Barzilai-Borwein with the step length one iteration late (a "gradient method with retards"): the length used for the move after
    evaluation k comes from the pair (s, y) that was complete BEFORE evaluation k — what a kernel can do whose replica sums arrive one step
    after the rows that produced them (the multi-step kernel's FIRE test has that lag too).  Per-bead cap on a move, as FIRE has."""
    _, g = fg(x); ne = 1; a = max_move / max(np.abs(g).max(), 1e-30) * 0.1; a_next = a; k = 0
    while rms(g) >= GT and ne < maxeval:
        step = (-a * g).reshape(-1, 3)
        ln = np.sqrt((step ** 2).sum(axis=1)); step *= np.minimum(1.0, max_move / np.maximum(ln, 1e-30))[:, None]
        xn = x + step.ravel(); _, gn = fg(xn); ne += 1
        s_ = xn - x; y_ = gn - g; sy = s_.dot(y_)
        a = a_next                                        # the length for the NEXT move: from the pair before this one
        a_next = ((s_.dot(s_) / sy) if k % 2 == 0 else (sy / y_.dot(y_))) if sy > 0 else a * 2.0
        x, g = xn, gn; k += 1
    return x, fg(x)[0], ne


def _compact_direction(g_now, S, Y, p_s, p_y, gamma):
    """-H g for the compact (Byrd-Nocedal-Schnabel 1994) form H = gamma I + [S gamma Y] M [S' ; gamma Y'] with the projections S'g, Y'g handed in:
    p_s, p_y may be those of the CURRENT gradient (a mid-step global sum) or of the previous one (what the kernel's replica sums are)."""
    m = len(S)
    if m == 0:
        return -gamma * g_now
    Sm, Ym = np.array(S), np.array(Y)
    SY = Sm @ Ym.T                                   # s_i . y_j
    R = np.triu(SY)
    D = np.diag(np.diag(SY))
    YY = Ym @ Ym.T
    Rinv = np.linalg.inv(R)
    # H = gamma I + [S gamma Y] [[Rinv' (D + gamma YY) Rinv, -Rinv'], [-Rinv, 0]] [S' ; gamma Y']
    top = Rinv.T @ ((D + gamma * YY) @ (Rinv @ p_s)) - Rinv.T @ (gamma * p_y)
    bot = -Rinv @ p_s
    return -(gamma * g_now + Sm.T @ top + gamma * (Ym.T @ bot))


def lbfgs_fixed_step(fg, x, m=5, maxeval=6000, max_move=0.5, late=False):
    """L-BFGS without a line search and without an energy — what a kernel step can afford: the quasi-Newton direction at unit length, every
    bead's move capped at max_move (as FIRE and the two-point steps are), a pair (s, y) kept when its curvature is positive, the memory
    dropped when the direction is not a descent direction.  late = False: the 2m projections S'g, Y'g are those of THIS evaluation's gradient
    — in the multi-step kernel a second global sum per step, between the force evaluation and the move.  late = True: the projections of
    the PREVIOUS evaluation's gradient (no second sum: how the kernel's replica sums arrive) with the current gradient in the gamma I term."""
    _, g = fg(x); ne = 1; S, Y = [], []
    gamma = max_move / max(np.abs(g).max(), 1e-30) * 0.1
    g_prev = g.copy()
    while rms(g) >= GT and ne < maxeval:
        gp = g_prev if late else g
        ps = np.array([s_.dot(gp) for s_ in S]); py = np.array([y_.dot(gp) for y_ in Y])
        d = _compact_direction(g, S, Y, ps, py, gamma)
        if g.dot(d) >= 0:                                # not downhill: forget the memory
            S, Y = [], []
            d = -gamma * g
        step = d.reshape(-1, 3)
        ln = np.sqrt((step ** 2).sum(axis=1)); step = step * np.minimum(1.0, max_move / np.maximum(ln, 1e-30))[:, None]
        xn = x + step.ravel(); _, gn = fg(xn); ne += 1
        s_ = xn - x; y_ = gn - g; sy = s_.dot(y_)
        if sy > 1e-12 * np.sqrt(s_.dot(s_) * y_.dot(y_)):
            S.append(s_); Y.append(y_)
            if len(S) > m: S.pop(0); Y.pop(0)
            gamma = sy / y_.dot(y_)
        else:
            S, Y = [], []
            gamma *= 2.0
        g_prev = g; x, g = xn, gn
    return x, fg(x)[0], ne


def cg(fg, x, maxeval=6000, max_move=0.5):
    f, g = fg(x); ne = 1; d = -g; t = max_move / max(np.abs(g).max(), 1e-30) * 0.1
    while rms(g) >= GT and ne < maxeval:
        gd = g.dot(d)
        if gd >= 0: d = -g; gd = -g.dot(g)
        big = np.abs(d).max()
        t = min(t * 2.0, 4 * max_move / big)
        while True:
            fn, gn = fg(x + t * d); ne += 1
            if fn <= f + 1e-4 * t * gd or t < 1e-12 or ne >= maxeval: break
            t *= 0.5
        beta = max(0.0, gn.dot(gn - g) / g.dot(g))
        x = x + t * d; d = -gn + beta * d; f, g = fn, gn
    return x, f, ne


def study_fixed_step(nrep, cids):
    """Round 6: would L-BFGS pay in the multi-step kernel?  Evaluations to the exit test for the two forms a kernel
    step could take — no energy, no line search — against the two-point steps that ship (length one evaluation late) and the textbook L-BFGS."""
    s = Solver(0)
    print("| matrix | N | replica | two-point, length late (ships) | L-BFGS m=5, Armijo (evaluations) | L-BFGS m=5 fixed step, projections of THIS gradient | same, m=3 | "
          "L-BFGS m=5 fixed step, projections one evaluation LATE |\n|" + "---|" * 8)
    tot = np.zeros(5)
    for cid in cids:
        IF = load_if(cid); n = IF.shape[0]
        rows = [(t.kind, t.nsteps, t.dt, t.w_all, t.w_vdw, t.repel_s, t.t_bath) for t in default_schedule(3000)]
        m = default_model(); s.set_model(m); d10 = pipeline.IF2dist_new(s, IF)
        s.set_schedule(make_stages(rows[:-1]), default_fire(), 0.0, 250); s.init_replicas(nrep, 82364, 0); s.run()
        x0 = s.coords()
        om = oracle_model_from(m, n); w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]
        def fg(u):
            F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
            return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
        for r in range(nrep):
            u = x0[r].astype(np.float64).ravel()
            n_bb = bb_lag(fg, u)[2]; n_l = lbfgs(fg, u)[2]; n_f5 = lbfgs_fixed_step(fg, u, 5)[2]; n_f3 = lbfgs_fixed_step(fg, u, 3)[2]
            n_late = lbfgs_fixed_step(fg, u, 5, late=True)[2]
            tot += (n_bb, n_l, n_f5, n_f3, n_late)
            print(f"| {cid} | {n} | {r} | {n_bb} | {n_l} | {n_f5} | {n_f3} | {n_late} |", flush=True)
    print(f"# totals: two-point late {tot[0]:.0f}; L-BFGS Armijo {tot[1]:.0f} ({tot[0] / tot[1]:.2f}x fewer); fixed step m=5 {tot[2]:.0f} ({tot[0] / tot[2]:.2f}x); "
          f"m=3 {tot[3]:.0f} ({tot[0] / tot[3]:.2f}x); projections late {tot[4]:.0f} ({tot[0] / tot[4]:.2f}x)   (6000 = did not converge)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fixed":
        study_fixed_step(int(sys.argv[2]) if len(sys.argv) > 2 else 4, sys.argv[3:] or ["chr21_1mb", "chr13_1mb", "chr4_1mb", "chr10_500kb", "chr1_500kb"])
        sys.exit(0)
    nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    cids = sys.argv[2:] or ["chr21_1mb", "chr13_1mb", "chr4_1mb", "chr10_500kb", "chr1_500kb"]
    s = Solver(0)
    print("| matrix | N | replica | FIRE steps (device, test every 10) | L-BFGS m=5 evaluations | BB evaluations | BB, step length one iteration late | PR+ CG evaluations | "
          "f - f(L-BFGS), relative: FIRE / BB / CG |\n|" + "---|" * 9)
    tot = np.zeros(5)
    for cid in cids:
        IF = load_if(cid); n = IF.shape[0]
        rows = [(t.kind, t.nsteps, t.dt, t.w_all, t.w_vdw, t.repel_s, t.t_bath) for t in default_schedule(3000)]
        rows[-1] = (2,) + rows[-1][1:]              # FIRE is what is compared here (the shipped final stage is kind 5 since round 5)
        m = default_model(); s.set_model(m); d10 = pipeline.IF2dist_new(s, IF)
        s.set_schedule(make_stages(rows[:-1]), default_fire(), 0.0, 250); s.init_replicas(nrep, 82364, 0); s.run()
        x0 = s.coords()
        om = oracle_model_from(m, n); w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]
        def fg(u):
            F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
            return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
        last = list(rows[-1]); last[1] = 6000
        for r in range(nrep):
            s.set_schedule(make_stages([tuple(last)]), default_fire(), GT, 10); s.init_replicas(1, 82364, 0); s.set_coords(x0[r:r + 1]); s.run()
            fsteps = s.last_timing()[1]; ffire = fg(s.coords()[0].astype(np.float64).ravel())[0]
            u = x0[r].astype(np.float64).ravel()
            xl, fl, nl = lbfgs(fg, u); xb, fb, nb = bb(fg, u); xb1, fb1, nb1 = bb_lag(fg, u); xc, fc, nc = cg(fg, u)
            tot += (fsteps, nl, nb, nb1, nc)
            print(f"| {cid} | {n} | {r} | {fsteps} | {nl} | {nb} | {nb1} | {nc} | {(ffire - fl) / fl:+.1e} / {(fb - fl) / fl:+.1e} / {(fc - fl) / fl:+.1e} |", flush=True)
    print(f"# totals: FIRE {tot[0]:.0f}, L-BFGS {tot[1]:.0f} ({tot[0] / tot[1]:.1f}x fewer), BB {tot[2]:.0f} ({tot[0] / tot[2]:.1f}x), BB late {tot[3]:.0f} ({tot[0] / tot[3]:.1f}x), CG {tot[4]:.0f} ({tot[0] / tot[4]:.1f}x)")
