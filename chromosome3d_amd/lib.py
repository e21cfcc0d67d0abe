"""ctypes binding of libc3d.so (the C ABI declared in include/c3d.h).

This module is plumbing: it loads the in-tree shared library built from
chromosome3d_amd/csrc (hand-written HIP for gfx950) and exposes its entry points.  There is
no Python or CPU fallback for the solver: if the library is missing, or no gfx950 device is
visible when a context is created, the call raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libc3d.so")


class C3DError(RuntimeError):
    pass


class Model(C.Structure):
    _fields_ = [("min_sep", C.c_int32), ("noe_pot", C.c_int32), ("rep_sep", C.c_int32), ("ang_mode", C.c_int32),
                ("s_noe", C.c_float), ("rswitch", C.c_float), ("asym", C.c_float),
                ("k_bond", C.c_float), ("b0", C.c_float), ("k_ang", C.c_float), ("a0", C.c_float),
                ("r0_rep", C.c_float), ("k_rep", C.c_float), ("mass", C.c_float), ("fbeta", C.c_float), ("masym", C.c_float), ("mrswitch", C.c_float),
                ("msoexp", C.c_int32)]


class Stage(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nsteps", C.c_int32), ("dt", C.c_float), ("w_all", C.c_float),
                ("w_vdw", C.c_float), ("repel_s", C.c_float), ("t_bath", C.c_float)]


class FireParams(C.Structure):
    _fields_ = [("dt_start", C.c_float), ("dt_max", C.c_float), ("f_inc", C.c_float), ("f_dec", C.c_float),
                ("alpha_start", C.c_float), ("f_alpha", C.c_float), ("max_step", C.c_float), ("n_min", C.c_int32)]


# name -> (restype, argtypes); every symbol include/c3d.h declares
_vp, _i, _l, _d, _f = C.c_void_p, C.c_int, C.c_long, C.c_double, C.c_float
_i32p, _fp, _dp = C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_double)
SIGNATURES = {
    "c3d_last_error": (C.c_char_p, []),
    "c3d_version": (C.c_char_p, []),
    "c3d_device_count": (_i, []),
    "c3d_create": (_i, [_i, C.POINTER(_vp)]),
    "c3d_destroy": (None, [_vp]),
    "c3d_default_model": (None, [C.POINTER(Model)]),
    "c3d_default_fire": (None, [C.POINTER(FireParams)]),
    "c3d_default_schedule": (_i, [C.POINTER(Stage), _i, _i]),
    "c3d_set_if_matrix": (_i, [_vp, _dp, _i, _d, _d]),
    "c3d_set_restraints": (_i, [_vp, _i, _i, _i32p, _i32p, _i32p]),
    "c3d_get_dist10": (_i, [_vp, _i32p]),
    "c3d_num_beads": (_i, [_vp]),
    "c3d_num_restraints": (_i, [_vp]),
    "c3d_set_model": (_i, [_vp, C.POINTER(Model)]),
    "c3d_set_schedule": (_i, [_vp, C.POINTER(Stage), _i, C.POINTER(FireParams), _f, _i]),
    "c3d_set_option": (_i, [_vp, C.c_char_p, _d]),
    "c3d_set_process_option": (_i, [C.c_char_p, _d]),
    "c3d_init_replicas": (_i, [_vp, _i, C.c_uint64, C.c_uint32]),
    "c3d_embed_replicas": (_i, [_vp, _i]),
    "c3d_set_coords": (_i, [_vp, _fp]),
    "c3d_get_coords": (_i, [_vp, _fp]),
    "c3d_get_velocities": (_i, [_vp, _fp]),
    "c3d_run": (_i, [_vp]),
    "c3d_run_steps": (_i, [_vp, _l, C.POINTER(_l)]),
    "c3d_schedule_length": (_l, [_vp]),
    "c3d_steps_done": (_l, [_vp]),
    "c3d_centre": (_i, [_vp]),
    "c3d_last_timing": (_i, [_vp, _dp, C.POINTER(_l), C.POINTER(_l)]),
    "c3d_get_stat": (_i, [_vp, C.c_char_p, _dp]),
    "c3d_step_kernel_name": (C.c_char_p, [_vp]),
    "c3d_debug_tear16": (_i, [_vp, _i, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "c3d_eval": (_i, [_vp, _f, _f, _f, _fp, _dp]),
    "c3d_get_energies": (_i, [_vp, _dp]),
    "c3d_score_replicas": (_i, [_vp, _dp, _i, _i32p, _dp, _dp]),
    "c3d_rank": (_i, [_vp, _i32p]),
    "c3d_reduce_model": (_i, [_dp, _i, _dp]),
    "c3d_model_similarity": (_i, [_dp, _dp, _i, _dp, _dp]),
    "c3d_parse_if_file": (_i, [C.c_char_p, C.POINTER(_dp), C.POINTER(_i)]),
    "c3d_free": (None, [_vp]),
    "c3d_write_front_half": (_i, [_i32p, _i, _i, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_i)]),
    "c3d_read_tbl": (_i, [C.c_char_p, C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_i)]),
    "c3d_write_pdb": (_i, [C.c_char_p, _fp, _i, _d, _d, _d, C.c_char_p]),
    "c3d_set_residue_sequence": (_i, [C.c_char_p]),
    "c3d_shape_pdb": (_i, [C.c_char_p, C.c_char_p, C.c_char_p]),
    "c3d_read_pdb_ca": (_i, [C.c_char_p, C.POINTER(_fp), C.POINTER(_i)]),
    "c3d_assess": (_i, [_fp, _i, _i, _i32p, _i32p, _i32p, _d, C.POINTER(_i), _dp]),
    "c3d_write_violations": (_i, [_fp, _i, _i, _i32p, _i32p, _i32p, _d, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(_i), _dp]),
    "c3d_spearman_if_dist": (_i, [_dp, _fp, _i, _i, _dp]),
    "c3d_spearman_if_dist_batch": (_i, [_dp, _fp, _i, _i, _i, _dp]),
}

_lib = None


def load():
    """Load libc3d.so; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise C3DError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(make -C chromosome3d_amd/csrc). There is no CPU fallback.")
        # a device exception should reach stderr in the runtime's own words: without this ROCr first pipes a GPU core dump to the helper
        # of /proc/sys/kernel/core_pattern and, where that does not exist, dies on the broken pipe before it has named the exception
        # (the CLIs do the same in main(); a user's own setting wins; no effect if the process initialised HIP before this import)
        os.environ.setdefault("HSA_DISABLE_COREDUMP_ON_EXCEPTION", "1")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise C3DError(f"libc3d error {rc}: {load().c3d_last_error().decode(errors='replace')}")


def fptr(a):
    return a.ctypes.data_as(_fp)


def dptr(a):
    return a.ctypes.data_as(_dp)


def i32ptr(a):
    return a.ctypes.data_as(_i32p)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
