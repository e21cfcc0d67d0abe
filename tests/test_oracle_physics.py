"""Self-consistency of the oracle's solver restatement (energy model, integrators)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import load_if, random_coil


@pytest.fixture(scope="module")
def small():
    IF = load_if("chr21_1mb")
    return IF, O.if_to_dist10(IF)


@pytest.mark.parametrize("pot", [0, 1, 2, 3])
def test_force_is_minus_gradient(small, pot):
    IF, d10 = small
    n = len(IF)
    m = O.default_model(n, noe_pot=pot, k_bond=700.0, k_ang=60.0, a0=7.4, ang_mode=1, r0_rep=6.0, masym=-0.1)
    x = random_coil(n, 3).astype(np.float64) * 0.6
    w_all, w_vdw, rs = 0.7, 2.0, 0.9
    F, _ = O.energy_force(m, d10, x, w_all, w_vdw, rs)

    def etot(xx):
        _, e = O.energy_force(m, d10, xx, w_all, w_vdw, rs)
        return w_all * (e[0] + e[1]) + w_vdw * e[2]
    rng = np.random.default_rng(0)
    for _ in range(12):
        i, c = rng.integers(n), rng.integers(3)
        h = 1e-5
        xp, xm = x.copy(), x.copy()
        xp[i, c] += h
        xm[i, c] -= h
        g = (etot(xp) - etot(xm)) / (2 * h)
        assert abs(F[i, c] + g) < 1e-4 * max(1.0, abs(g))


@pytest.mark.parametrize("kw", [dict(mrswitch=10.0, masym=0.0, msoexp=2), dict(mrswitch=10.0, masym=0.0, msoexp=1), dict(mrswitch=4.0, masym=8.0, msoexp=1),
                                dict(mrswitch=6.0, masym=1.5, msoexp=2)])
def test_lower_side_forms_force_is_minus_gradient_and_continuous(small, kw):
    """noe_pot 3, the shipped lower side (square up to mrswitch, then the CNS soft form a + b / D^msoexp + masym D) and its relatives: the
    force is minus the gradient on a coil collapsed far inside its targets (hundreds of pairs beyond mrswitch), an independent numpy
    restatement of the NOE energy gives the same number, and energy and force are continuous at D = mrswitch."""
    IF, d10 = small
    n = len(IF)
    m = O.default_model(n, noe_pot=3, rswitch=0.5, asym=2.0, k_bond=500.0, k_ang=15.0, a0=5.55, ang_mode=1, r0_rep=5.25, k_rep=4.0, **kw)
    x = random_coil(n, 11).astype(np.float64) * (0.25 if kw["mrswitch"] >= 10 else 0.7)
    t = d10 / 10.0
    i, j = np.triu_indices(n, 5)
    ok = t[i, j] > 0
    D = t[i, j][ok] - np.linalg.norm(x[i] - x[j], axis=1)[ok]
    assert (D > kw["mrswitch"]).sum() > 50 and (D < kw["mrswitch"]).sum() > 20
    F, e = O.energy_force(m, d10, x, 1.0, 1.0, 0.85)
    # numpy restatement of the NOE energy alone (S = 10)
    mrs, mc, p = kw["mrswitch"], kw["masym"], kw["msoexp"]
    mb = (mc - 2 * mrs) * mrs ** (p + 1) / p
    ma = mrs * mrs - mb / mrs ** p - mc * mrs
    dl = -D
    en = np.where(dl > 0.5, 1.0 * dl - 0.25, dl * dl)                       # upper side: rswitch 0.5, slope 1.0: a = rs^2 - c rs = -0.25
    en = np.where(D > mrs, ma + mb / np.maximum(D, 1e-9) ** p + mc * D, en)
    assert abs(10.0 * en.sum() - e[0]) < 1e-8 * abs(e[0])

    def etot(xx):
        _, ee = O.energy_force(m, d10, xx, 1.0, 1.0, 0.85)
        return ee[0] + ee[1] + ee[2]
    rng = np.random.default_rng(1)
    for _ in range(12):
        a, c = rng.integers(n), rng.integers(3)
        h = 1e-5
        xp, xm = x.copy(), x.copy()
        xp[a, c] += h
        xm[a, c] -= h
        g = (etot(xp) - etot(xm)) / (2 * h)
        assert abs(F[a, c] + g) < 1e-4 * max(1.0, abs(g))
    # continuity at the switch: the soft form's value and slope equal the square's
    assert abs(ma + mb / mrs ** p + mc * mrs - mrs * mrs) < 1e-9 and abs(mc - p * mb / mrs ** (p + 1) - 2 * mrs) < 1e-9


def test_net_force_and_torque_vanish(small):
    IF, d10 = small
    n = len(IF)
    m = O.default_model(n, noe_pot=1)
    x = random_coil(n, 5).astype(np.float64)
    F, _ = O.energy_force(m, d10, x, 1, 1, 0.85)
    assert np.abs(F.sum(0)).max() < 1e-8 * np.abs(F).max() * n
    assert np.abs(np.cross(x, F).sum(0)).max() < 1e-7 * np.abs(F).max() * n * 50


def test_fire_converges_and_md_thermostat(small):
    IF, d10 = small
    n = len(IF)
    m = O.default_model(n, noe_pot=1, k_bond=700.0, k_ang=60.0, a0=7.4, ang_mode=1, r0_rep=6.0)
    fire = O.default_fire()
    x, v, ev = O.run_schedule(m, d10, O.make_stages([(2, 4000, 0, 1.0, 1.0, 0.85, 0)]), fire, 82364, 0)
    F, e = O.energy_force(m, d10, x, 1, 1, 0.85)
    assert np.sqrt((F ** 2).mean()) < 1e-3
    assert ev == 4000
    # Berendsen coupling pulls T to the bath within a few tau (1/fbeta = 0.1 ps = 33 steps)
    x2, v2, _ = O.run_schedule(m, d10, O.make_stages([(0, 600, 0.003, 0.4, 0.003, 0.9, 2000.0)]), fire, 82364, 0, x0=x)
    T = m.mass * (v2 ** 2).sum() / 418.4 / ((3 * n - 3) * 0.0019872)
    assert 1200 < T < 2800
    # hard rescale: T equals the bath exactly at the previous half step -> close now
    x3, v3, _ = O.run_schedule(m, d10, O.make_stages([(1, 50, 0.005, 1.0, 0.1, 1.0, 300.0)]), fire, 82364, 0, x0=x)
    T3 = m.mass * (v3 ** 2).sum() / 418.4 / ((3 * n - 3) * 0.0019872)
    assert 150 < T3 < 600
    assert abs(v3.sum(0)).max() < 1e-6 * np.abs(v3).max() * n + 1e-9   # COM removed


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
    import ctypes as C
    L = O.lib()

    def run(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        L.c3o_philox4x32(c, k, o)
        return [int(v) for v in o]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


@pytest.mark.parametrize("kind", [2, 5])
def test_fire_and_lbfgs_reach_the_same_minimum(kind):
    """(kind 2: FIRE, the final stage of rounds 1-4; kind 5: the stage as shipped since round 5 — two-point step sizes, then FIRE.)
    The final stage of the reference is `minimize lbfgs nstep=15000 drop=10` x 10 (deck chromosome3D.pl:1790-1803); here it is FIRE
    (one force evaluation per step, own-row data only).  A documented deviation in ALGORITHM, not in RESULT: from the same
    post-cooling state, FIRE (the restatement's, as the GPU runs it) and L-BFGS (scipy's L-BFGS-B, 10 correction pairs, on the
    restatement's energy and its analytic gradient — checked against a central difference first) end in the same minimum: total energy to
    1e-7 relative, every pair distance to 0.02 A, the same truncated NOE energy rank key, the same Spearman to 1e-4."""
    from scipy.optimize import minimize
    from chromosome3d_amd import default_fire, default_model, default_schedule, pipeline
    from tests.util import load_if, oracle_fire_from, oracle_model_from
    IF = load_if("chr21_1mb")
    n = IF.shape[0]
    d10 = O.if_to_dist10(IF)
    om, of = oracle_model_from(default_model(), n), oracle_fire_from(default_fire())
    rows = [(s.kind, s.nsteps, s.dt, s.w_all, s.w_vdw, s.repel_s, s.t_bath) for s in default_schedule(3000)]
    assert rows[-1][0] == 5
    rows[-1] = (kind,) + rows[-1][1:]
    w_all, w_vdw, rs = rows[-1][3], rows[-1][4], rows[-1][5]

    def fg(u):
        F, e = O.energy_force(om, d10, u.reshape(n, 3), w_all, w_vdw, rs)
        return w_all * (e[0] + e[1]) + w_vdw * e[2], -F.ravel()
    for r in range(3):
        x0, _, _ = O.run_schedule(om, d10, O.make_stages(rows[:-1]), of, 82364, r)
        u = x0.ravel().copy()
        g = fg(u)[1]
        for k in (5, 17, 3 * n - 2):                                    # the objective L-BFGS sees IS the energy whose force FIRE follows
            h = 1e-5
            up, um = u.copy(), u.copy()
            up[k] += h
            um[k] -= h
            assert abs((fg(up)[0] - fg(um)[0]) / (2 * h) - g[k]) <= 1e-5 * max(1.0, abs(g[k]))
        xf, _, evf = O.run_schedule(om, d10, O.make_stages(rows[-1:]), of, 82364, r, x0=x0, gtol=1e-2, check_every=250)
        res = minimize(fg, x0.ravel(), jac=True, method="L-BFGS-B", options=dict(maxiter=15000, maxfun=150000, ftol=1e-15, gtol=1e-6, maxcor=10))
        xl = res.x.reshape(n, 3)
        ff, fl = fg(xf.ravel())[0], res.fun
        assert evf < 3000 and res.nit < 15000
        assert abs(ff - fl) <= 1e-7 * abs(fl), (ff, fl)
        i, j = np.triu_indices(n, 1)
        assert np.abs(np.linalg.norm(xf[i] - xf[j], axis=1) - np.linalg.norm(xl[i] - xl[j], axis=1)).max() < 0.02
        ef = O.energy_force(om, d10, xf, w_all, w_vdw, rs)[1][0]
        el = O.energy_force(om, d10, xl, w_all, w_vdw, rs)[1][0]
        assert abs(ef - el) < 1.0                                        # the ranking key is int(E_noe): chromosome3D.pl:796-802
        assert abs(pipeline.spearman_IF_pdb(IF, xf.astype(np.float32)) - pipeline.spearman_IF_pdb(IF, xl.astype(np.float32))) < 1e-4


def test_two_point_stage_is_its_two_parts_and_reaches_the_fire_minimum():
    """Stage kind 5 (round 5, the default schedule's final stage; c3o_bb_step): two-point step sizes for the first
    `final_minimiser_steps`, then FIRE from a fresh state — the same as running the two parts as two stages; and from a half-relaxed
    coil it reaches the exit test in about half of FIRE's evaluations and ends in FIRE's minimum or a lower neighbour."""
    from chromosome3d_amd import default_fire, default_model
    from tests.util import oracle_fire_from, oracle_model_from
    IF = load_if("chr13_1mb")
    n = IF.shape[0]
    d10 = O.if_to_dist10(IF)
    om, of = oracle_model_from(default_model(), n), oracle_fire_from(default_fire())
    pre = O.make_stages([(2, 300, 0.0, 1.0, 1.0, 0.85, 0.0)])
    tot = np.zeros(2)
    try:
        for r in range(4):
            x0, _, _ = O.run_schedule(om, d10, pre, of, 82364, r)
            O.set_two_point_steps(25)
            xa, va, ea = O.run_schedule(om, d10, O.make_stages([(5, 70, 0.0, 1.0, 1.0, 0.85, 0.0)]), of, 82364, r, x0=x0)
            O.set_two_point_steps(1000)
            xb, vb, eb = O.run_schedule(om, d10, O.make_stages([(5, 25, 0.0, 1.0, 1.0, 0.85, 0.0), (2, 45, 0.0, 1.0, 1.0, 0.85, 0.0)]), of, 82364, r, x0=x0)
            assert ea == eb == 70 and np.array_equal(xa, xb) and np.array_equal(va, vb)
            xt, _, et = O.run_schedule(om, d10, O.make_stages([(5, 6000, 0.0, 1.0, 1.0, 0.85, 0.0)]), of, 82364, r, x0=x0, gtol=1e-2, check_every=10)
            xf, _, ef = O.run_schedule(om, d10, O.make_stages([(2, 6000, 0.0, 1.0, 1.0, 0.85, 0.0)]), of, 82364, r, x0=x0, gtol=1e-2, check_every=10)
            Ft, e_t = O.energy_force(om, d10, xt, 1.0, 1.0, 0.85)
            tot += (et, ef)
            # (the exit test sees the force BEFORE the last move, as on the device: after it the RMS may be a little above the bound)
            assert np.sqrt((Ft * Ft).mean()) < 5e-2 and et < 6000 and ef < 6000
            e_f = O.energy_force(om, d10, xf, 1.0, 1.0, 0.85)[1]
            assert sum(e_t) <= sum(e_f) * (1 + 1e-4), (sum(e_t), sum(e_f))      # the same minimum (measured: three of four to 1e-8) or a lower neighbour
        assert tot[0] < 0.7 * tot[1], tot                                    # measured: 1240 evaluations against 2440
    finally:
        O.set_two_point_steps(1000)
