"""ctypes view of oracle/c3d_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product package (chromosome3d_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libc3d_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "c3d_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class Model(C.Structure):
    _fields_ = [("n", C.c_int), ("min_sep", C.c_int), ("noe_pot", C.c_int), ("rep_sep", C.c_int), ("ang_mode", C.c_int),
                ("s_noe", C.c_double), ("rswitch", C.c_double), ("asym", C.c_double), ("masym", C.c_double), ("mrswitch", C.c_double),
                ("k_bond", C.c_double), ("b0", C.c_double), ("r0_rep", C.c_double),
                ("k_rep", C.c_double), ("k_ang", C.c_double), ("a0", C.c_double),
                ("mass", C.c_double), ("fbeta", C.c_double), ("msoexp", C.c_int)]


class Stage(C.Structure):
    _fields_ = [("kind", C.c_int), ("nsteps", C.c_int), ("dt", C.c_double), ("w_all", C.c_double),
                ("w_vdw", C.c_double), ("repel_s", C.c_double), ("t_bath", C.c_double)]


class FireParams(C.Structure):
    _fields_ = [("dt_start", C.c_double), ("dt_max", C.c_double), ("f_inc", C.c_double),
                ("f_dec", C.c_double), ("alpha_start", C.c_double), ("f_alpha", C.c_double),
                ("max_step", C.c_double), ("n_min", C.c_int)]


class FireState(C.Structure):
    _fields_ = [("dt", C.c_double), ("alpha", C.c_double), ("npos", C.c_int), ("started", C.c_int)]


class Energy(C.Structure):
    _fields_ = [("e_noe", C.c_double), ("e_bond", C.c_double), ("e_rep", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        i32p = C.POINTER(C.c_int32)
        L.c3o_parse_if_text.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(dp)]
        L.c3o_parse_if_text.restype = C.c_int
        L.c3o_free.argtypes = [C.c_void_p]
        L.c3o_if_to_dist10.argtypes = [dp, C.c_int, C.c_double, C.c_double, i32p]
        L.c3o_dist_to_rr.argtypes = [i32p, C.c_int, C.c_int, C.POINTER(ip), C.POINTER(ip), C.POINTER(i32p)]
        L.c3o_dist_to_rr.restype = C.c_int
        for f in (L.c3o_write_rr, L.c3o_write_tbl):
            f.argtypes = [C.c_char_p, ip, ip, i32p, C.c_int]
        L.c3o_write_dist.argtypes = [C.c_char_p, i32p, C.c_int]
        L.c3o_energy_force.argtypes = [C.POINTER(Model), i32p, dp, C.c_double, C.c_double, C.c_double, dp, C.POINTER(Energy)]
        L.c3o_energy_force.restype = None
        L.c3o_init_coords.argtypes = [C.POINTER(Model), C.c_uint64, C.c_uint32, dp]
        L.c3o_init_coords_extended.argtypes = [C.POINTER(Model), C.c_uint64, C.c_uint32, dp]
        L.c3o_init_velocities.argtypes = [C.POINTER(Model), C.c_uint64, C.c_uint32, C.c_double, dp]
        L.c3o_md_step.argtypes = [C.POINTER(Model), i32p, C.POINTER(Stage), dp, dp, dp]
        L.c3o_fire_step.argtypes = [C.POINTER(Model), i32p, C.POINTER(Stage), C.POINTER(FireParams), C.POINTER(FireState), dp, dp, dp, dp]
        L.c3o_run_schedule.argtypes = [C.POINTER(Model), i32p, C.POINTER(Stage), C.c_int, C.POINTER(FireParams),
                                       C.c_double, C.c_int, C.c_uint64, C.c_uint32, dp, dp]
        L.c3o_run_schedule.restype = C.c_long
        L.c3o_set_two_point_steps.argtypes = [C.c_int]
        L.c3o_set_two_point_steps.restype = None
        L.c3o_bb_step.argtypes = [C.POINTER(Model), i32p, C.POINTER(Stage), C.POINTER(FireParams), C.POINTER(FireState), dp, dp, dp, dp]
        L.c3o_assess.argtypes = [dp, C.c_int, ip, ip, i32p, C.c_double, ip, dp]
        L.c3o_spearman_if_dist.argtypes = [dp, dp, C.c_int, C.c_int]
        L.c3o_spearman_if_dist.restype = C.c_double
        L.c3o_dg_bounds.argtypes = [C.POINTER(Model), i32p, C.c_double, dp, dp]
        L.c3o_dg_smooth.argtypes = [C.c_int, dp, dp]
        L.c3o_dg_trial_d2.argtypes = [C.c_int, dp, dp, C.c_uint64, C.c_uint32, dp]
        L.c3o_dg_embed.argtypes = [C.c_int, dp, C.c_uint64, C.c_uint32, C.c_int, dp]
        for f in (L.c3o_dg_bounds, L.c3o_dg_smooth, L.c3o_dg_trial_d2, L.c3o_dg_embed):
            f.restype = None
        L.c3o_philox4x32.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i32p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def parse_if_text(text: bytes):
    out = C.POINTER(C.c_double)()
    n = lib().c3o_parse_if_text(text, len(text), C.byref(out))
    if n <= 0:
        raise ValueError("oracle: cannot parse IF matrix")
    m = np.ctypeslib.as_array(out, shape=(n, n)).copy()
    lib().c3o_free(out)
    return m


def if_to_dist10(IF, alpha=0.5, K=11.0):
    IF = np.ascontiguousarray(IF, dtype=np.float64)
    n = IF.shape[0]
    d = np.empty((n, n), dtype=np.int32)
    rc = lib().c3o_if_to_dist10(_dp(IF), n, alpha, K, _i32p(d))
    assert rc == 0
    return d


def dist_to_rr(dist10, min_sep=5):
    n = dist10.shape[0]
    pi, pj, pt = C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int32)()
    d = np.ascontiguousarray(dist10, dtype=np.int32)
    R = lib().c3o_dist_to_rr(_i32p(d), n, min_sep, C.byref(pi), C.byref(pj), C.byref(pt))
    ri = np.ctypeslib.as_array(pi, shape=(max(R, 1),))[:R].copy().astype(np.intc)
    rj = np.ctypeslib.as_array(pj, shape=(max(R, 1),))[:R].copy().astype(np.intc)
    rt = np.ctypeslib.as_array(pt, shape=(max(R, 1),))[:R].copy().astype(np.int32)
    for p in (pi, pj, pt):
        lib().c3o_free(p)
    return ri, rj, rt


def write_front_half(outdir, cid, dist10, rr):
    ri, rj, rt = rr
    L = lib()
    assert L.c3o_write_dist(os.path.join(outdir, f"{cid}.dist").encode(), _i32p(np.ascontiguousarray(dist10)), dist10.shape[0]) == 0
    assert L.c3o_write_rr(os.path.join(outdir, f"{cid}.rr").encode(), _ip(ri), _ip(rj), _i32p(rt), len(ri)) == 0
    assert L.c3o_write_tbl(os.path.join(outdir, "contact.tbl").encode(), _ip(ri), _ip(rj), _i32p(rt), len(ri)) == 0


def default_model(n, **kw):
    m = Model(n=n, min_sep=5, noe_pot=0, rep_sep=2, ang_mode=0, s_noe=10.0, rswitch=1.0, asym=2.0, masym=0.1, mrswitch=1.0,
              k_bond=100.0, b0=3.8, r0_rep=4.0, k_rep=1.0, k_ang=0.0, a0=5.0, mass=100.0, fbeta=10.0, msoexp=1)
    for k, v in kw.items():
        setattr(m, k, v)
    return m


def default_fire(**kw):
    f = FireParams(dt_start=0.002, dt_max=0.02, f_inc=1.1, f_dec=0.5, alpha_start=0.1,
                   f_alpha=0.99, max_step=0.5, n_min=5)
    for k, v in kw.items():
        setattr(f, k, v)
    return f


def energy_force(model, tgt10, x, w_all=1.0, w_vdw=1.0, repel_s=1.0):
    x = np.ascontiguousarray(x, dtype=np.float64)
    F = np.zeros_like(x)
    e = Energy()
    t = np.ascontiguousarray(tgt10, dtype=np.int32)
    lib().c3o_energy_force(C.byref(model), _i32p(t), _dp(x), w_all, w_vdw, repel_s, _dp(F), C.byref(e))
    return F, (e.e_noe, e.e_bond, e.e_rep)


def init_coords(model, seed, replica, start=0):
    """start 0: random coil; 1: the extended strand of extn.inp (chromosome3D.pl:2413-2416) at bead level"""
    x = np.zeros((model.n, 3))
    (lib().c3o_init_coords_extended if start == 1 else lib().c3o_init_coords)(C.byref(model), seed, replica, _dp(x))
    return x


def init_velocities(model, seed, replica, temp=0.5):
    v = np.zeros((model.n, 3))
    lib().c3o_init_velocities(C.byref(model), seed, replica, temp, _dp(v))
    return v


def make_stages(rows):
    arr = (Stage * len(rows))()
    for k, r in enumerate(rows):
        arr[k] = Stage(*r)
    return arr


def set_two_point_steps(k):
    """Stage kind 5: evaluations of the two-point step-size minimiser before FIRE takes the stage over (device option final_minimiser_steps)."""
    lib().c3o_set_two_point_steps(int(k))


def run_schedule(model, tgt10, stages, fire, seed, replica, x0=None, gtol=0.0, check_every=0):
    x = init_coords(model, seed, replica) if x0 is None else np.array(x0, dtype=np.float64, order="C")
    v = np.zeros_like(x)
    t = np.ascontiguousarray(tgt10, dtype=np.int32)
    evals = lib().c3o_run_schedule(C.byref(model), _i32p(t), stages, len(stages), C.byref(fire), gtol,
                                   check_every, seed, replica, _dp(x), _dp(v))
    return x, v, evals


def assess(x, rr, relax=0.5):
    ri, rj, rt = rr
    x = np.ascontiguousarray(x, dtype=np.float64)
    sat = C.c_int()
    dev = C.c_double()
    lib().c3o_assess(_dp(x), len(ri), _ip(ri), _ip(rj), _i32p(rt), relax, C.byref(sat), C.byref(dev))
    return sat.value, dev.value


def spearman_if_dist(IF, x, rng=3):
    IF = np.ascontiguousarray(IF, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    return lib().c3o_spearman_if_dist(_dp(IF), _dp(x), IF.shape[0], rng)


def dg_bounds(model, tgt10, lower_default):
    n = model.n
    U = np.empty((n, n)); Lo = np.empty((n, n))
    t = np.ascontiguousarray(tgt10, dtype=np.int32)
    lib().c3o_dg_bounds(C.byref(model), _i32p(t), lower_default, _dp(U), _dp(Lo))
    return U, Lo


def dg_smooth(U, Lo):
    U = np.array(U, dtype=np.float64, order="C"); Lo = np.array(Lo, dtype=np.float64, order="C")
    lib().c3o_dg_smooth(U.shape[0], _dp(U), _dp(Lo))
    return U, Lo


def dg_trial_d2(U, Lo, seed, replica):
    n = U.shape[0]
    D2 = np.empty((n, n))
    lib().c3o_dg_trial_d2(n, _dp(np.ascontiguousarray(U)), _dp(np.ascontiguousarray(Lo)), seed, replica, _dp(D2))
    return D2


def dg_embed(D2, seed, replica, iters=50):
    n = D2.shape[0]
    x = np.zeros((n, 3))
    lib().c3o_dg_embed(n, _dp(np.ascontiguousarray(D2)), seed, replica, iters, _dp(x))
    return x
