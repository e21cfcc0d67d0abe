"""A deterministic stand-in for chromosome3d_amd.Solver, for CPU rehearsals of the MULTI-RANK plumbing only (bench.py's line at eight
ranks, the batch driver's LPT + gather): there is no GPU in the build container and a GPU box admits at most six processes on its card.
It anneals nothing: `run` / `run_steps` only move the schedule position, coordinates are random coils keyed by (bead count, replica id),
energies a fixed function of the same — "recorded results standing in for the solver" (VERDICT round 4, item 3).  K1 comes from the CPU
oracle.  Test infrastructure: nothing under chromosome3d_amd/, bin/ or bench.py knows this file exists; tests install it by patching
`chromosome3d_amd.Solver` inside their own worker processes."""
import numpy as np


class StandinSolver:
    def __init__(self, device=0):
        self.device, self.n, self.nrep, self.first, self.pos, self.L, self._d10, self._last = device, 0, 0, 0, 0, 0, None, (0.0, 0, 0)
        self.calls = []

    def close(self):
        pass

    def set_model(self, model):
        pass

    def set_option(self, key, value):
        pass

    def set_schedule(self, stages, fire=None, gtol=0.0, check_every=250):
        self.L = int(sum(s.nsteps for s in stages))

    def set_if_matrix(self, IF, alpha=0.5, K=11.0):
        from oracle import oracle as O
        self._d10 = O.if_to_dist10(np.asarray(IF, dtype=np.float64), alpha, K)
        self.n = IF.shape[0]

    def dist10(self):
        return self._d10

    @property
    def num_restraints(self):
        d, n = self._d10, self.n
        i, j = np.triu_indices(n, 5)
        return int((d[i, j] > 0).sum())

    def init_replicas(self, nrep, seed=82364, first_replica=0):
        self.nrep, self.first, self.pos, self.seed = nrep, first_replica, 0, seed

    @property
    def schedule_length(self):
        return self.L

    @property
    def steps_done(self):
        return self.pos

    def run_steps(self, nsteps):
        did = max(0, min(int(nsteps), self.L - self.pos))
        self.pos += did
        self._last = (1e-3 * did, did, 1 if did else 0)
        return did

    def run(self):
        self.run_steps(self.L)

    def last_timing(self):
        return self._last

    def stat(self, key):
        return 0.0

    @property
    def step_kernel_name(self):
        return "standin (no kernel)"

    def _rng(self, r):
        return np.random.default_rng([self.n, self.first + r, int(self.seed) & 0xFFFF])

    def coords(self):
        x = np.empty((self.nrep, self.n, 3), dtype=np.float32)
        for r in range(self.nrep):
            d = self._rng(r).normal(size=(self.n, 3))
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            c = np.cumsum(3.9 * d, axis=0)
            x[r] = c - c.mean(0)
        return x

    def energies(self):
        e = np.zeros((self.nrep, 3))
        for r in range(self.nrep):
            e[r, 0] = 1e4 * self.n + 1e3 * self._rng(r).uniform()
        return e

    def score(self, IF=None, rng=3):
        rho = np.array([-0.5 - 0.4 * self._rng(r).uniform() for r in range(self.nrep)])
        return np.zeros(self.nrep, dtype=np.int32), np.zeros(self.nrep), (rho if IF is not None else None)

    def rank(self):
        e = self.energies()[:, 0].astype(np.int64)
        return np.lexsort((np.arange(self.nrep), e)).astype(np.int32)
