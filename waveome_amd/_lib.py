"""ctypes binding of libwaveome_amd.so (C-ABI declared in include/waveome_amd.h).

The product path has no CPU fallback: if the shared library is missing this module
raises at import of the symbol table (build it with `python __graft_entry__.py` or
`make -C waveome_amd/csrc`)."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (WAVEOME_AMD_LIB: a diagnostic build of the same sources - tools/libwv_stamps.so, tools/libwv_probes.so)
LIB_PATH = os.environ.get("WAVEOME_AMD_LIB") or os.path.join(_HERE, "libwaveome_amd.so")

WV_OK = 0
# factor types
K_SE, K_M12, K_M32, K_M52, K_PERIODIC, K_LIN, K_POLY, K_CONST, K_CAT, K_EMPTY = range(10)
T_IDENTITY, T_SOFTPLUS, T_SOFTPLUS_SHIFT = 0, 1, 2
LB_CONV_GTOL, LB_FG, LB_CONV_FTOL, LB_STOP_MAXITER, LB_STOP_MAXFUN, LB_ABNORMAL = 0, 1, 2, 3, 4, 5
LB_MESSAGES = {
    LB_CONV_GTOL: "CONVERGENCE: NORM_OF_PROJECTED_GRADIENT_<=_PGTOL",
    LB_CONV_FTOL: "CONVERGENCE: REL_REDUCTION_OF_F_<=_FACTR*EPSMCH",
    LB_STOP_MAXITER: "STOP: TOTAL NO. of ITERATIONS REACHED LIMIT",
    LB_STOP_MAXFUN: "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT",
    LB_ABNORMAL: "ABNORMAL_TERMINATION_IN_LNSRCH",
}


class Factor(C.Structure):
    _fields_ = [("type", C.c_int32), ("dim_mask", C.c_uint32), ("param", C.c_int32 * 3),
                ("degree", C.c_int32)]


class Param(C.Structure):
    _fields_ = [("init", C.c_double), ("prior_scale", C.c_double), ("transform", C.c_int32),
                ("trainable", C.c_int32)]


class KernelSpec(C.Structure):
    _fields_ = [("n_dims", C.c_int32), ("n_params", C.c_int32), ("params", C.POINTER(Param)),
                ("n_terms", C.c_int32), ("term_start", C.POINTER(C.c_int32)),
                ("factors", C.POINTER(Factor)), ("noise_param", C.c_int32),
                ("mean_param", C.c_int32)]


class ProblemCaps(C.Structure):
    _fields_ = [("max_specs", C.c_int32), ("max_params", C.c_int32), ("max_feat", C.c_int32),
                ("max_factors", C.c_int32), ("max_red", C.c_int32), ("nan_is_missing", C.c_int32)]


class LbfgsOpts(C.Structure):
    _fields_ = [("m", C.c_int32), ("maxiter", C.c_int32), ("maxfun", C.c_int32),
                ("maxls", C.c_int32), ("ftol", C.c_double), ("gtol", C.c_double)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# every symbol include/waveome_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "wv_abi_version": (C.c_int32, []),
    "wv_last_error": (C.c_char_p, []),
    "wv_lbfgs_default_opts": (None, [C.POINTER(LbfgsOpts)]),
    "wv_spec_check": (C.c_int32, [C.POINTER(KernelSpec), _ip]),
    "wv_spec_block_column": (C.c_int32, [C.POINTER(KernelSpec), _ip]),
    "wv_block_spans": (C.c_int32, [_dp, C.c_int64, C.c_int32, C.c_int32, _ip]),
    "wv_spec_constrain": (C.c_int32, [C.POINTER(KernelSpec), _dp, _dp]),
    "wv_spec_unconstrain": (C.c_int32, [C.POINTER(KernelSpec), _dp, _dp]),
    "wv_spec_log_prior": (C.c_int32, [C.POINTER(KernelSpec), _dp, _dp, _dp]),
    "wv_lbfgs_create": (C.c_int32, [C.c_int32, C.POINTER(LbfgsOpts), _dp, C.POINTER(_vp)]),
    "wv_lbfgs_x": (_dp, [_vp]),
    "wv_lbfgs_feed": (C.c_int32, [_vp, C.c_double, _dp]),
    "wv_lbfgs_n_iter": (C.c_int32, [_vp]),
    "wv_lbfgs_n_eval": (C.c_int32, [_vp]),
    "wv_lbfgs_f": (C.c_double, [_vp]),
    "wv_lbfgs_destroy": (None, [_vp]),
    "wv_device_count": (C.c_int32, [_ip]),
    "wv_ctx_create": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "wv_ctx_destroy": (None, [_vp]),
    "wv_ctx_synchronize": (C.c_int32, [_vp]),
    "wv_problem_create": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int32,
                                      C.POINTER(KernelSpec), C.c_int32, C.c_int32, C.c_int32,
                                      C.POINTER(_vp)]),
    "wv_problem_create_multi": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int32,
                                            C.POINTER(C.POINTER(KernelSpec)), C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.POINTER(_vp)]),
    "wv_problem_create_ex": (C.c_int32, [_vp, _vp, C.c_int64, C.c_int32, _vp, C.c_int32,
                                         C.POINTER(C.POINTER(KernelSpec)), C.c_int32, C.POINTER(ProblemCaps),
                                         C.c_int32, C.c_int32, C.c_int32, C.POINTER(_vp)]),
    "wv_problem_add_spec": (C.c_int32, [_vp, C.POINTER(KernelSpec), _ip]),
    "wv_queue_create": (C.c_int32, [_vp, C.POINTER(LbfgsOpts), C.POINTER(_vp)]),
    "wv_queue_submit": (C.c_int32, [_vp, C.c_int32, _ip, _ip, _dp, C.c_int32, _dp, C.POINTER(C.c_int64)]),
    "wv_queue_advance": (C.c_int32, [_vp, _ip, _ip]),
    "wv_queue_fetch": (C.c_int32, [_vp, C.c_int32, C.c_int32, C.POINTER(C.c_int64), _dp, _dp, _dp, _ip, _ip, _ip,
                                   _ip]),
    "wv_queue_destroy": (None, [_vp]),
    "wv_problem_destroy": (None, [_vp]),
    "wv_problem_workspace_bytes": (C.c_int64, [_vp]),
    "wv_gram": (C.c_int32, [_vp, _dp, _dp]),
    "wv_lml": (C.c_int32, [_vp, _ip, C.c_int32, _dp, _dp, _ip]),
    "wv_value_grad": (C.c_int32, [_vp, _ip, C.c_int32, _dp, _dp, _dp, _ip]),
    "wv_fit": (C.c_int32, [_vp, _ip, C.c_int32, C.POINTER(LbfgsOpts), _dp, _dp, _dp, _ip, _ip, _ip]),
    "wv_fit_penalized": (C.c_int32, [_vp, _ip, C.c_int32, C.POINTER(LbfgsOpts), _dp, _dp, _dp, _dp, _ip, _ip, _ip]),
    "wv_fit_multi": (C.c_int32, [_vp, _ip, _ip, C.c_int32, C.POINTER(LbfgsOpts), _dp, C.c_int32, _dp, _dp, _dp, _ip,
                                 _ip, _ip]),
    "wv_predict": (C.c_int32, [_vp, C.c_int32, _dp, _dp, C.c_int64, C.c_int32, _dp, _dp]),
    "wv_whitened_means": (C.c_int32, [_vp, C.c_int32, _dp, C.POINTER(C.c_uint64), C.c_int32, C.c_double, _dp]),
    "wv_whitened_state": (C.c_int32, [_vp, C.c_int32, _dp, C.c_double, _dp, _dp]),
    "wv_stage_times": (C.c_int32, [_vp, C.c_int32, C.POINTER(C.c_char_p), _dp,
                                   C.POINTER(C.c_int64), _dp, _dp]),
    "wv_stage_times_reset": (None, [_vp]),
    "wv_stage_timing_enable": (None, [_vp, C.c_int32]),
    # include/waveome_amd_debug.h
    "wv_dbg_gemm": (C.c_int32, [_vp, _dp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_int32]),
    "wv_dbg_stage": (C.c_int32, [_vp, C.c_int32, _dp, C.c_int32, _dp]),
    "wv_dbg_npad": (C.c_int64, [_vp]),
}

# include/waveome_amd_probes.h: rate / timing probes, present only in the `make probes` build
# (tools/libwv_probes.so through WAVEOME_AMD_LIB); bound when the loaded library exports them
PROBE_SIGNATURES = {
    "wv_dbg_gemm_bench": (C.c_int32, [_vp] + [C.c_int32] * 8 + [_dp, _dp]),
    "wv_dbg_mfma_f64_rate": (C.c_int32, [_vp, C.c_int32, C.c_int32, _dp, _dp]),
    "wv_dbg_valu_f64_rate": (C.c_int32, [_vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]),
    "wv_dbg_mfma4_probe": (C.c_int32, [_vp, _ip]),
    "wv_dbg_copy_rate": (C.c_int32, [_vp, C.c_int64, C.c_int32, _dp]),
    "wv_dbg_gemm_stamps": (C.c_int32, [C.POINTER(C.c_uint64)]),
    "wv_dbg_exp_rate": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, _dp]),
    "wv_dbg_exp_check": (C.c_int32, [C.c_int32, _dp, C.c_int64, C.POINTER(C.c_int64), _dp]),
}

_lib = None


class WaveomeAmdError(RuntimeError):
    pass


def load():
    """Load libwaveome_amd.so and bind every declared symbol.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WaveomeAmdError(
            f"{LIB_PATH} is missing: the HIP extension was not built. There is no CPU fallback; "
            "run `python __graft_entry__.py` (or `make -C waveome_amd/csrc`).")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in PROBE_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if lib.wv_abi_version() != 2:
        raise WaveomeAmdError("libwaveome_amd.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != WV_OK:
        msg = load().wv_last_error()
        raise WaveomeAmdError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def as_dp(a):
    return a.ctypes.data_as(_dp)


def as_ip(a):
    return a.ctypes.data_as(_ip)
