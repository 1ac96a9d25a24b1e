"""ctypes binding of include/mcmcdate_mvn.h (libmcmcdate_mvn.so).

The product has no CPU path: if the shared library is missing or no GPU is present, the calls
below raise -- they never fall back to anything else.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCD_LIB_PATH: load another build of the same C ABI (diagnostic / tuning builds under tools/)
LIB_PATH = os.environ.get("MCD_LIB_PATH") or os.path.join(_HERE, "libmcmcdate_mvn.so")

MCD_OK = 0
MCD_ERR_INVALID_ARG = -1
MCD_ERR_NOT_SPD = -2
MCD_ERR_HIP = -3
MCD_ERR_ROOT_NOT_BIFURCATING = -4
MCD_ERR_NO_DEVICE = -5
MCD_ERR_UNSUPPORTED = -6
MCD_MAT_SIGMA = 0
MCD_MAT_SIGMA_INV = 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol declared in include/mcmcdate_mvn.h
SYMBOLS = {
    "mcd_device_count": (C.c_int, []),
    "mcd_version": (C.c_char_p, []),
    "mcd_last_error": (C.c_char_p, []),
    "mcd_set_option": (C.c_int, [C.c_char_p, C.c_char_p]),
    "mcd_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mcd_set_logpdf_form": (C.c_int, [C.c_int]),
    "mcd_mvn_set_form": (C.c_int, [_vp, C.c_int]),
    "mcd_mvn_release_stream": (C.c_int, [_vp, _vp]),
    "mcd_mvn_create": (C.c_int, [C.POINTER(_vp), C.c_int, _dp, _dp, C.c_int, C.c_double, C.c_int]),
    "mcd_mvn_destroy": (None, [_vp]),
    "mcd_mvn_dim": (C.c_int, [_vp]),
    "mcd_mvn_device": (C.c_int, [_vp]),
    "mcd_mvn_logdet": (C.c_double, [_vp]),
    "mcd_mvn_get_factor": (C.c_int, [_vp, _dp]),
    "mcd_mvn_logpdf": (C.c_int, [_vp, _dp, _dp]),
    "mcd_mvn_logpdf_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp]),
    "mcd_mvn_grad_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int64]),
    "mcd_tree_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, _ip]),
    "mcd_tree_destroy": (None, [_vp]),
    "mcd_tree_n_nodes": (C.c_int, [_vp]),
    "mcd_tree_loglik_batch": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]),
    "mcd_tree_grad_batch": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcd_prior_create": (C.c_int, [C.POINTER(_vp), C.c_int, _ip, C.c_double, C.c_int,
                                   C.c_int, _ip, _ip, _dp, _dp, _ip, _dp, _dp,
                                   C.c_int, _ip, _ip, _dp,
                                   C.c_int, _ip, _ip, _dp, C.c_int]),
    "mcd_prior_destroy": (None, [_vp]),
    "mcd_prior_logprior_batch": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp, _vp]),
    "mcd_prior_grad_batch": (C.c_int, [_vp] * 8 + [C.c_int64, C.c_int64, C.c_int, _vp] + [_vp] * 8),
    "mcd_sparse_create": (C.c_int, [C.POINTER(_vp), C.c_int, _dp, C.c_int64, _ip, _ip, _dp, C.c_double, C.c_int]),
    "mcd_sparse_destroy": (None, [_vp]),
    "mcd_sparse_dim": (C.c_int, [_vp]),
    "mcd_sparse_nnz": (C.c_int64, [_vp]),
    "mcd_sparse_release_stream": (C.c_int, [_vp, _vp]),
    "mcd_sparse_logpdf_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp]),
    "mcd_sparse_grad_batch": (C.c_int, [_vp, _vp, C.c_int64, C.c_int64, C.c_int, _vp, _vp, _vp, C.c_int64]),
    "mcd_sparse_tree_create": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, _ip]),
    "mcd_sparse_tree_destroy": (None, [_vp]),
    "mcd_sparse_tree_loglik_batch": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp, _vp, C.c_int64, C.c_int, _vp, _vp, _vp]),
    "mcd_hmc_create": (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_int, C.c_int64]),
    "mcd_hmc_destroy": (None, [_vp]),
    "mcd_hmc_dim": (C.c_int, [_vp]),
    "mcd_hmc_set_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64]),
    "mcd_hmc_get_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64]),
    "mcd_hmc_get_position": (C.c_int, [_vp, _dp, _dp, _dp]),
    "mcd_hmc_leapfrog": (C.c_int, [_vp, _dp, _dp, _dp, _dp, C.c_int]),
    "mcd_hmc_step_from": (C.c_int, [_vp, _dp, _dp, _dp, C.c_int, _dp, _dp, _dp, _dp]),
    "mcd_shard_unique_id": (C.c_int, [C.c_char_p]),
    "mcd_shard_comm_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "mcd_shard_comm_destroy": (None, [_vp]),
    "mcd_shard_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "mcd_shard_allgather": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "mcd_mh_posterior_device": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "mcd_hmc_nuts": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_uint64, C.c_int64, C.c_uint64, _dp, _ip]),
    "mcd_hmc_nuts_run": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_uint64, C.c_int64, C.c_uint64, _dp, _dp, _dp]),
    "mcd_hmc_nuts_warmup": (C.c_int, [_vp, C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_uint64, C.c_int64, C.c_uint64, _dp]),
    "mcd_mh_create": (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, _dp, _dp, C.c_int64, C.c_uint64]),
    "mcd_mh_create_sparse": (C.c_int, [C.POINTER(_vp), _vp, _vp, C.c_int, _ip, _ip, _ip, _ip, _ip, _ip, _dp, _dp, C.c_int64, C.c_uint64]),
    "mcd_mh_destroy": (None, [_vp]),
    "mcd_mh_set_chain_offset": (C.c_int, [_vp, C.c_int64]),
    "mcd_mh_set_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64]),
    "mcd_mh_get_state": (C.c_int, [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int64]),
    "mcd_mh_get_posterior": (C.c_int, [_vp, _dp]),
    "mcd_mh_run": (C.c_int, [_vp, _ip, C.c_int64, C.c_int32, C.c_int, _dp, C.POINTER(C.c_int8)]),
    "mcd_mh_tune": (C.c_int, [_vp]),
    "mcd_mh_get_tuning": (C.c_int, [_vp, _dp, _ip, _ip]),
    "mcd_mh_set_tuning": (C.c_int, [_vp, _dp]),
    "mcd_mh_reset_counters": (C.c_int, [_vp]),
    "mcd_mh_set_temperatures": (C.c_int, [_vp, _dp]),
    "mcd_mh_last_path": (C.c_int, [_vp]),
    "mcd_mh_last_dynamic_lds": (C.c_int64, [_vp]),
    "mcd_mh_mc3_init": (C.c_int, [_vp, C.c_int, _dp, C.c_int64, C.c_uint64]),
    "mcd_mh_mc3_swap": (C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int64]),
    "mcd_mh_mc3_get": (C.c_int, [_vp, _ip, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _dp]),
    "mcd_mh_get_age_sums": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_int64)]),
    "mcd_mh_reset_age_sums": (C.c_int, [_vp]),
}


class McdError(RuntimeError):
    """Structural fault reported by the C ABI (the reference raises `error` for these)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[{code}] {message}")
        self.code = code


class NotPositiveDefinite(McdError):
    pass


class RootNotBifurcating(McdError):
    pass


class NoDevice(McdError):
    pass


_lib = None


def lib():
    """The loaded shared library.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  mcmc-date_amd has no CPU fallback."
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def set_option(name: str, value=None):
    """A test / tuning knob of the library (include/mcmcdate_mvn.h: mcd_set_option): value = an integer (or its string), None = back to the
    default.  The names are the environment variables of rounds 1-3 ("MCD_MH_SEGMENTS", "MCD_SPLIT", ...); the environment itself is read
    once, when the library is loaded."""
    check(lib().mcd_set_option(name.encode(), None if value is None else str(value).encode()))


def get_option(name: str):
    """The knob's value, or None when it is at its default."""
    s, v = C.c_int(0), C.c_int(0)
    check(lib().mcd_get_option(name.encode(), C.byref(s), C.byref(v)))
    return int(v.value) if s.value else None


def check(rc: int):
    if rc == MCD_OK:
        return
    msg = lib().mcd_last_error().decode()
    cls = {MCD_ERR_NOT_SPD: NotPositiveDefinite, MCD_ERR_ROOT_NOT_BIFURCATING: RootNotBifurcating,
           MCD_ERR_NO_DEVICE: NoDevice}.get(rc, McdError)
    raise cls(rc, msg)
