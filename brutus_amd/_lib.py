"""ctypes binding of the C ABI in include/brutus_amd.h (libbrutus_amd.so).

The shared library is built in-tree by `__graft_entry__.build()` (hipcc,
--offload-arch=gfx950).  There is NO CPU fallback: if the library is missing or
no GPU is visible, every compute entry point raises.
"""
import ctypes as C
import os

__all__ = ["lib", "Params", "PostParams", "check", "LIB_PATH", "BrutusError", "NVALS",
           "MAX_BATCH", "MAX_FILT", "MAX_FILT_FIT"]

# BRUTUS_AMD_LIB: another build of the same library (A/B kernel timing)
LIB_PATH = os.environ.get("BRUTUS_AMD_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "libbrutus_amd.so")
NVALS = 11
ABI_VERSION = 4
MAX_BATCH = 256
MAX_FILT = 64          # bands per call (full-grid pipeline); include/brutus_amd.h
MAX_FILT_FIT = 32      # bands the hot path (brutus_fit_batch) takes at once

_lib = None


class BrutusError(RuntimeError):
    pass


class Params(C.Structure):
    """struct brutus_params (include/brutus_amd.h)."""
    _fields_ = [("avlim", C.c_double * 2), ("av_gauss", C.c_double * 2),
                ("rvlim", C.c_double * 2), ("rv_gauss", C.c_double * 2),
                ("ltol", C.c_double), ("ltol_subthresh", C.c_double),
                ("init_thresh", C.c_double), ("wt_thresh", C.c_double),
                ("dim_prior", C.c_int32), ("max_iter", C.c_int32)]


_vp, _i64, _i32, _sz, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_double
_u64 = C.c_uint64


class PostParams(C.Structure):
    """struct brutus_post_params (include/brutus_amd.h)."""
    _fields_ = [("nmc", C.c_int32), ("ndraws", C.c_int32),
                ("return_distreds", C.c_int32), ("has_feh", C.c_int32),
                ("has_loga", C.c_int32), ("per_object", C.c_int32),
                ("wt_thresh", C.c_double), ("avlim", C.c_double * 2),
                ("rvlim", C.c_double * 2), ("nsel_max", C.c_int64),
                ("object0", C.c_int64), ("seed", C.c_uint64),
                ("normal_base", C.c_uint64), ("uniform_base", C.c_uint64),
                ("R_solar", C.c_double), ("Z_solar", C.c_double),
                ("R_thin", C.c_double), ("Z_thin", C.c_double),
                ("Rs_thin", C.c_double), ("R_thick", C.c_double),
                ("Z_thick", C.c_double), ("f_thick", C.c_double),
                ("Rs_thick", C.c_double), ("Rs_halo", C.c_double),
                ("q_halo_ctr", C.c_double), ("q_halo_inf", C.c_double),
                ("r_q_halo", C.c_double), ("eta_halo", C.c_double),
                ("f_halo", C.c_double), ("feh_mean", C.c_double * 3),
                ("feh_sigma", C.c_double * 3), ("age_mean", C.c_double * 3),
                ("age_sigma", C.c_double * 3), ("age_lnnorm", C.c_double * 3),
                ("min_age", C.c_double), ("max_age", C.c_double),
                ("frame_mat", C.c_double * 9), ("frame_off", C.c_double * 3)]


# name -> (restype, argtypes); mirrors include/brutus_amd.h (product ABI) and
# include/brutus_amd_debug.h (test hooks / measurement aids, see DEBUG_NAMES) one to one
DEBUG_NAMES = ("brutus_calibrate_traffic", "brutus_calibrate_copy16", "brutus_calibrate_issue", "brutus_debug_exp10", "brutus_debug_math", "brutus_debug_mt_stream", "brutus_debug_rng", "brutus_debug_galprior", "brutus_debug_galprior_mc", "brutus_debug_galprior_sl", "brutus_debug_zig_table", "brutus_debug_copy", "brutus_debug_sizeof_star32", "brutus_debug_fit_stats", "brutus_debug_pre32_time")
SIGNATURES = {
    "brutus_abi_version": (C.c_int, []),
    "brutus_last_error": (C.c_char_p, []),
    "brutus_padded_filters": (C.c_int, [_i32]),
    "brutus_grid_soa_bytes": (_sz, [_i64, _i32]),
    "brutus_grid_relayout": (C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    "brutus_workspace_bytes": (_sz, [_i64, _i32, _i32]),
    "brutus_loglike_batch": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp,
                                       _vp, _i32, C.POINTER(Params), _vp, _sz,
                                       _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp, _vp]),
    "brutus_fit_batch": (C.c_int, [_vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp,
                                   _i32, C.POINTER(Params), _vp, _sz, _i64, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "brutus_last_timing": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_char_p),
                                     C.POINTER(C.c_float), C.c_int]),
    "brutus_enable_timing": (None, [C.c_int]),
    "brutus_calibrate_traffic": (C.c_int, [_vp, _vp, _i64, _vp]),
    "brutus_calibrate_copy16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "brutus_calibrate_issue": (C.c_int, [_i32, _i32, _i32, _vp, _i64, _vp]),
    "brutus_debug_exp10": (C.c_int, [_vp, _vp, _i64, _vp]),
    "brutus_debug_math": (C.c_int, [_i32, _vp, _vp, _i64, _vp]),
    "brutus_post_workspace_bytes": (_sz, [_i32, _i64, _i32]),
    "brutus_post_batch": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _vp, C.POINTER(PostParams), _vp, _sz, _vp, _vp, _vp,
                                    _vp, _vp, _vp]),
    "brutus_post_batch_numpy": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, C.POINTER(PostParams), _vp, _sz, _vp, _vp, _vp,
                                          _vp, _i32, _vp, _vp, _sz, _vp]),
    "brutus_post_batch_numpy_phase": (C.c_int, [_i32, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                                _vp, _vp, C.POINTER(PostParams), _vp, _sz, _vp,
                                                _vp, _vp, _vp, _i32, _vp, _vp, _sz, _i32, _vp]),
    "brutus_debug_mt_stream": (C.c_int, [_i32, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "brutus_set_mt_jump": (C.c_int, [_vp, _i32, _i64, _i64]),
    "brutus_post_set_dust": (C.c_int, [_vp, _vp, _i32, _dbl, _dbl, _dbl, _dbl]),
    "brutus_post_set_after_jump": (C.c_int, [_vp, _vp]),
    "brutus_debug_rng": (C.c_int, [_u64, _u64, _i64, _vp, _vp, _vp]),
    "brutus_debug_zig_table": (C.c_int, [_vp, _vp, _i32]),
    "brutus_debug_fit_stats": (C.c_int, [_vp, _vp]),
    "brutus_debug_galprior": (C.c_int, [C.POINTER(PostParams), _i32, _vp, _vp, _vp, _vp,
                                        _vp, _vp]),
    "brutus_debug_galprior_mc": (C.c_int, [C.POINTER(PostParams), _i32, _vp, _vp, _vp, _vp,
                                           _vp, _vp]),
    "brutus_debug_galprior_sl": (C.c_int, [C.POINTER(PostParams), _i32, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp]),
    "brutus_debug_copy": (C.c_int, [_vp, _sz, _i64, _i32, _i32, _i32, _vp, _sz, _vp]),
    "brutus_debug_sizeof_star32": (C.c_int, []),
    "brutus_debug_pre32_time": (C.c_int, [_vp, _sz, _vp, _i64, _i32, _i32, C.POINTER(Params), _i32, _i32,
                                          C.POINTER(C.c_float), _vp]),
    "brutus_cluster_points": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "brutus_cluster_points_grid": (C.c_int, [_i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "brutus_offsets_weights": (C.c_int, [_i32, _i32, _i32, _i64] + [_vp] * 12 + [_i32, _vp, _vp, _vp]),
    "brutus_offsets_workspace_bytes": (C.c_size_t, [_i32, _i32]),
    "brutus_offsets_bootstrap": (C.c_int, [_i32] * 6 + [_vp] * 7 + [C.c_size_t, _vp, _vp]),
    "brutus_cluster_workspace_bytes": (_sz, [_i32]),
    "brutus_cluster_lnl": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _i32, _vp, _sz, _vp, _vp]),
    "brutus_cluster_chunks": (C.c_int, []),
    "brutus_cluster_lnl_part": (C.c_int, [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _i32, _vp, _sz, _i32, _i32, _vp]),
    "brutus_cluster_lnl_part_mags": (C.c_int, [_i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp,
                                               _vp, _vp, _vp, _i32, _vp, _sz, _i32, _i32, _vp]),
    "brutus_cluster_lnl_merge": (C.c_int, [_i32, _i32, _vp, _sz, _vp, _vp]),
    "brutus_cluster_mix": (C.c_int, [_i32, _vp, _vp, C.c_double, C.c_double, _vp, _vp, _vp]),
}


def lib():
    """Load (once) and return the shared library; raise loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrutusError(
            "brutus_amd: HIP library %s not found. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "There is no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries its own libamdhip64, and a process must end
    # up with ONE HIP runtime -- the one that owns the device pointers we are
    # handed.  Loading ours first would bind /opt/rocm's copy and the second
    # runtime then finds "no ROCm-capable device".
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    if L.brutus_abi_version() != ABI_VERSION:
        raise BrutusError("brutus_amd: ABI version mismatch")
    # jump-ahead polynomials of MT19937 (data, see tools/gen_mt_jump.py): lets many
    # workgroups walk one numpy random stream; without the file one workgroup does
    jp = os.path.join(os.path.dirname(os.path.abspath(__file__)), "mt_jump.npz")
    if os.path.exists(jp):
        import numpy as np
        z = np.load(jp)
        polys = np.ascontiguousarray(z["polys"], dtype=np.uint32)
        check_rc = L.brutus_set_mt_jump(polys.ctypes.data, int(polys.shape[0]),
                                        int(z["strides"][0]), int(z["strides"][1]))
        if check_rc != 0:
            raise BrutusError("brutus_amd: mt_jump.npz does not match the library")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        msg = lib().brutus_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise BrutusError("brutus_amd error %d: %s" % (rc, msg))
