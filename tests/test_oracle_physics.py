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
