"""Thin object view of a libc3d context (one per GPU / process)."""
import ctypes as C

import numpy as np

from . import lib as _l


def default_model(**kw):
    m = _l.Model()
    _l.load().c3d_default_model(C.byref(m))
    for k, v in kw.items():
        if not hasattr(m, k):
            raise AttributeError(f"c3d_model has no field {k}")
        setattr(m, k, v)
    return m


def default_fire(**kw):
    f = _l.FireParams()
    _l.load().c3d_default_fire(C.byref(f))
    for k, v in kw.items():
        setattr(f, k, v)
    return f


def default_schedule(min_steps=3000):
    L = _l.load()
    n = L.c3d_default_schedule(None, 0, min_steps)
    arr = (_l.Stage * n)()
    L.c3d_default_schedule(arr, n, min_steps)
    return arr


def make_stages(rows):
    arr = (_l.Stage * len(rows))()
    for k, r in enumerate(rows):
        arr[k] = _l.Stage(*r)
    return arr


class Solver:
    """c3d_ctx wrapper.  Raises lib.C3DError when no gfx950 device / library is available."""

    def __init__(self, device=0):
        self._L = _l.load()
        self._h = C.c_void_p()
        _l.check(self._L.c3d_create(device, C.byref(self._h)))
        self.device = device
        self.n = 0
        self.nrep = 0

    def close(self):
        if self._h:
            self._L.c3d_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- problem ----
    def set_model(self, model):
        _l.check(self._L.c3d_set_model(self._h, C.byref(model)))

    def set_schedule(self, stages, fire=None, gtol=0.0, check_every=250):
        fire = fire if fire is not None else default_fire()
        _l.check(self._L.c3d_set_schedule(self._h, stages, len(stages), C.byref(fire), gtol, check_every))

    def set_option(self, key, value):
        _l.check(self._L.c3d_set_option(self._h, key.encode(), float(value)))

    def set_if_matrix(self, IF, alpha=0.5, K=11.0):
        IF = np.ascontiguousarray(IF, dtype=np.float64)
        assert IF.ndim == 2 and IF.shape[0] == IF.shape[1]
        _l.check(self._L.c3d_set_if_matrix(self._h, _l.dptr(IF), IF.shape[0], alpha, K))
        self.n = IF.shape[0]

    def set_restraints(self, n, ri, rj, rt10):
        ri, rj, rt10 = (np.ascontiguousarray(a, dtype=np.int32) for a in (ri, rj, rt10))
        _l.check(self._L.c3d_set_restraints(self._h, n, len(ri), _l.i32ptr(ri), _l.i32ptr(rj), _l.i32ptr(rt10)))
        self.n = n

    def dist10(self):
        d = np.empty((self.n, self.n), dtype=np.int32)
        _l.check(self._L.c3d_get_dist10(self._h, _l.i32ptr(d)))
        return d

    @property
    def num_restraints(self):
        return self._L.c3d_num_restraints(self._h)

    # ---- replicas ----
    def init_replicas(self, nrep, seed=82364, first_replica=0):
        _l.check(self._L.c3d_init_replicas(self._h, nrep, seed, first_replica))
        self.nrep = nrep

    def embed(self, iters=50):
        """A7: metric-matrix distance-geometry start for every replica."""
        _l.check(self._L.c3d_embed_replicas(self._h, iters))

    def set_coords(self, xyz):
        xyz = _l.as_f32(xyz)
        assert xyz.shape == (self.nrep, self.n, 3)
        _l.check(self._L.c3d_set_coords(self._h, _l.fptr(xyz)))

    def coords(self):
        x = np.empty((self.nrep, self.n, 3), dtype=np.float32)
        _l.check(self._L.c3d_get_coords(self._h, _l.fptr(x)))
        return x

    def velocities(self):
        v = np.empty((self.nrep, self.n, 3), dtype=np.float32)
        _l.check(self._L.c3d_get_velocities(self._h, _l.fptr(v)))
        return v

    # ---- solve ----
    def run(self):
        _l.check(self._L.c3d_run(self._h))

    def run_steps(self, nsteps):
        done = C.c_long()
        _l.check(self._L.c3d_run_steps(self._h, nsteps, C.byref(done)))
        return done.value

    def centre(self):
        _l.check(self._L.c3d_centre(self._h))

    @property
    def schedule_length(self):
        return self._L.c3d_schedule_length(self._h)

    @property
    def steps_done(self):
        return self._L.c3d_steps_done(self._h)

    def last_timing(self):
        ms, st, la = C.c_double(), C.c_long(), C.c_long()
        _l.check(self._L.c3d_last_timing(self._h, C.byref(ms), C.byref(st), C.byref(la)))
        return ms.value, st.value, la.value

    def stat(self, key):
        v = C.c_double()
        _l.check(self._L.c3d_get_stat(self._h, key.encode(), C.byref(v)))
        return v.value

    def debug_tear16(self, iterations=200000):
        """(unit reads, torn units, reads that saw a new value) of the hand-off's 16-byte store / load pair (c3d_debug_tear16)."""
        a, b, c = C.c_ulonglong(), C.c_ulonglong(), C.c_ulonglong()
        _l.check(self._L.c3d_debug_tear16(self._h, iterations, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    @property
    def step_kernel_name(self):
        return self._L.c3d_step_kernel_name(self._h).decode()

    def eval(self, w_all=1.0, w_vdw=1.0, repel_s=0.85, forces=True, energies=True):
        F = np.empty((self.nrep, self.n, 3), dtype=np.float32) if forces else None
        e = np.empty((self.nrep, 3), dtype=np.float64) if energies else None
        _l.check(self._L.c3d_eval(self._h, w_all, w_vdw, repel_s, _l.fptr(F) if forces else None,
                                  _l.dptr(e) if energies else None))
        return F, e

    def energies(self):
        e = np.empty((self.nrep, 3), dtype=np.float64)
        _l.check(self._L.c3d_get_energies(self._h, _l.dptr(e)))
        return e

    def score(self, IF=None, rng=3):
        """K6 on the device: (satisfied[M], sum_dev[M], spearman[M] or None) at the current coordinates."""
        sat = np.empty(self.nrep, dtype=np.int32)
        dev = np.empty(self.nrep, dtype=np.float64)
        rho = np.empty(self.nrep, dtype=np.float64) if IF is not None else None
        IFc = np.ascontiguousarray(IF, dtype=np.float64) if IF is not None else None
        _l.check(self._L.c3d_score_replicas(self._h, _l.dptr(IFc) if IF is not None else None, rng, _l.i32ptr(sat), _l.dptr(dev),
                                            _l.dptr(rho) if IF is not None else None))
        return sat, dev, rho

    def rank(self):
        r = np.empty(self.nrep, dtype=np.int32)
        _l.check(self._L.c3d_rank(self._h, _l.i32ptr(r)))
        return r
