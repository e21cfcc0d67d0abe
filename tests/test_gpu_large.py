"""BASELINE.json configs[4]: synthetic N = 2500 (R ~ 3.1 M restraints), beyond anything the reference
can run (N <= 663, chromosome3D.pl:93-94).  Size-independent properties + oracle parity of one
force evaluation at full size."""
import numpy as np
import pytest

from tests.util import oracle_model_from, random_coil, synthetic_if

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big():
    return synthetic_if(2500)


def test_k1_and_forces_at_n2500(solver, big):
    from chromosome3d_amd import default_model, pipeline
    from oracle import oracle as O
    IF, truth = big
    n = IF.shape[0]
    m = default_model()
    solver.set_model(m)
    d10 = pipeline.IF2dist_new(solver, IF)
    assert np.array_equal(d10, O.if_to_dist10(IF))
    assert solver.num_restraints == (n - 5) * (n - 4) // 2          # every pair restrained: R = 3 113 760
    solver.init_replicas(2, 82364, 0)
    x = np.stack([truth.astype(np.float32) * 1.1, random_coil(n, 3) * 0.3])
    solver.set_coords(x)
    F, e = solver.eval(1.0, 1.0, 0.85)
    om = oracle_model_from(m, n)
    for r in range(2):
        Fo, eo = O.energy_force(om, d10, x[r].astype(np.float64), 1.0, 1.0, 0.85)
        scale = np.abs(Fo).max()
        assert (np.abs(F[r] - Fo) <= 2e-5 * np.abs(Fo) + 2e-6 * scale).all()
        assert np.allclose(e[r], eo, rtol=1e-6)
    assert (np.abs(F.sum(1)).max(1) < 5e-5 * np.abs(F).sum(1).max(1)).all()     # Newton's third law


def test_anneal_recovers_synthetic_structure(solver, big):
    """Ground truth is known: after the schedule the model's pair distances correlate with the
    generating structure's (Spearman > 0.9) and Spearman(IF, d) is strongly negative."""
    from chromosome3d_amd import default_model, default_schedule, pipeline
    from scipy.stats import spearmanr
    IF, truth = big
    n = IF.shape[0]
    solver.set_model(default_model())
    pipeline.IF2dist_new(solver, IF)
    solver.set_schedule(default_schedule(1500), None, 0.0, 250)
    solver.init_replicas(2, 82364, 0)
    solver.run()
    ms, steps, launches = solver.last_timing()
    x = solver.coords()
    e = solver.energies()
    assert np.isfinite(x).all() and np.isfinite(e).all()
    rng = np.random.default_rng(0)
    i = rng.integers(0, n, 200000)
    j = rng.integers(0, n, 200000)
    keep = np.abs(i - j) >= 3
    dt = np.linalg.norm(truth[i[keep]] - truth[j[keep]], axis=1)
    for r in range(2):
        dm = np.linalg.norm(x[r, i[keep]] - x[r, j[keep]], axis=1)
        assert spearmanr(dt, dm)[0] > 0.9
        assert spearmanr(IF[i[keep], j[keep]], dm)[0] < -0.85
    print(f"N=2500 x 2 replicas: {steps} SA steps in {ms:.1f} ms = {1e3 * ms / launches:.1f} us/step")
