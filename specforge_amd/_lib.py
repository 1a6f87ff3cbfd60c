"""ctypes binding of libsfhip.so (the C-ABI declared in include/specforge_amd.h).

The product path loads ``specforge_amd/libsfhip.so`` -- the hipcc/gfx950 build -- and
nothing else: if it is missing, every op raises.  There is no CPU fallback.
``_inject_library_for_tests`` exists so the test-suite can run the *same kernel
sources* under the SIMT interpreter build (tests/emu); production code never calls it.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libsfhip.so")

_lib = None
_emulated = False

P = c_void_p
_SIGS = {
    "sf_abi_version": (c_int, []),
    "sf_is_emulated": (c_int, []),
    "sf_last_error": (c_char_p, []),
    "sf_gemm_nt": (c_int, [P, c_long, P, c_long, P, c_int, c_long, c_int, c_int, c_int, c_float, c_float, P, c_long, P]),
    "sf_gemm_nt_ws": (c_int, [P, c_long, P, c_long, P, c_int, c_long, c_int, c_int, c_int, P, c_long, P, c_long, P]),
    "sf_gemm_tn": (c_int, [P, c_long, P, c_long, P, c_int, c_long, c_int, c_int, c_int, c_float, c_float, P, c_long, c_int, P]),
    "sf_gemm_nt_rowadd": (c_int, [P, c_long, P, c_long, P, c_int, c_long, c_int, c_int, c_int, c_float, P, c_long, c_int,
                                  c_int, c_int, P, c_long, P]),
    "sf_ce_fused": (c_int, [P, c_int, c_long, c_int, c_int, P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_int,
                            P, P, P, P, P, P]),
    "sf_ce_lk_grad": (c_int, [P, c_int, c_long, c_int, c_int, P, c_int, c_int, c_int, P, P, P, c_int, c_float, c_float,
                              c_float, c_float, P, P, P]),
    "sf_reduce_sum": (c_int, [P, c_long, c_int, P, c_float, P]),
    "sf_eagle3_metrics": (c_int, [P, P, P, c_int, c_int, c_int, c_int, P, P]),
    "sf_teacher_reduce": (c_int, [P, c_int, c_long, c_int, c_int, c_int, P, P, P, c_int, c_int, P, P, P, P, P, P]),
    "sf_gemm_nt_teacher": (c_int, [P, c_long, P, c_long, c_int, c_int, c_int, c_int, P, c_long, P, c_long, P, P, P]),
    "sf_teacher_reduce_perm": (c_int, [P, c_int, c_long, c_int, c_int, c_int, c_int, P, P, P, c_int, c_long, P, c_int, c_int, P, P, P, P,
                                       P, P, P, P]),
    "sf_gemm_nt_teacher_reduces": (c_int, [c_int, c_int, c_int, c_int]),
    "sf_ce_fused_zt": (c_int, [P, c_int, c_long, c_int, c_int, P, c_long, P, P, c_int, c_int, c_int, P, P, P, P, P, P, c_float, c_int, P, P,
                               P, P, P, P]),
    "sf_rmsnorm_fwd": (c_int, [P, c_int, c_long, P, c_int, c_int, c_int, P, c_float, c_int, c_int, P, c_long, P, P]),
    "sf_rmsnorm_fwd2": (c_int, [P, c_int, c_long, P, P, c_long, P, P, P, c_long, P, c_float, c_int, c_int, P]),
    "sf_rmsnorm_bwd2": (c_int, [P, c_long, P, P, c_int, P, c_long, P, P, c_int, c_int, P, c_long, P, c_int, c_int, P, c_long, P, c_long, P, P]),
    "sf_rmsnorm_bwd_workspace_floats": (c_long, [c_int, c_int]),
    "sf_rmsnorm_bwd": (c_int, [P, c_int, c_long, P, c_long, P, c_int, c_int, c_int, P, P, c_int, c_int, P, c_long, P,
                               c_long, P, c_int, P, P]),
    "sf_colsum_accum": (c_int, [P, c_int, c_int, P, c_int, P]),
    "sf_rows_expand": (c_int, [P, c_int, c_long, P, P, c_long, c_long, c_int, P]),
    "sf_rope": (c_int, [P, c_int, c_long, c_int, c_int, c_int, P, P, P, c_int, c_int, c_int, P]),
    "sf_swiglu_fwd": (c_int, [P, c_int, c_long, c_long, c_int, P, c_long, P]),
    "sf_swiglu_bwd": (c_int, [P, c_int, c_long, P, c_long, c_long, c_int, P, c_long, P]),
    "sf_gemm_nt_swiglu_fwd": (c_int, [P, c_long, P, c_long, c_int, c_int, c_int, P, c_long, P, c_long, P]),
    "sf_gemm_nt_swiglu_bwd": (c_int, [P, c_long, P, c_long, c_int, c_int, c_int, P, c_long, P, c_long, P, c_long, P]),
    "sf_transpose": (c_int, [P, c_int, c_long, c_long, c_long, P, c_long, c_long, c_long, c_int, c_int, c_int, c_int, P]),
    "sf_axpy_f32": (c_int, [c_long, c_float, P, P, c_int, P]),
    "sf_add_bf16": (c_int, [c_long, P, P, P, P]),
    "sf_shift_accum": (c_int, [P, c_long, P, c_long, c_int, c_int, c_int, c_int, c_int, P]),
    "sf_shift_sum_split": (c_int, [P, c_long, c_int, c_int, c_int, c_int, c_int, P, P, c_long, P]),
    "sf_split_bf16": (c_int, [P, c_long, P, P, c_long, c_long, c_int, P]),
    "sf_cast_from_f32": (c_int, [P, c_long, P, c_int, c_long, c_long, c_int, c_float, P]),
    "sf_attn_fwd": (c_int, [P, c_long, P, c_long, P, P, P, c_int, P, P, c_long, P, c_int, c_int, c_int, c_int, c_int,
                            c_float, P]),
    "sf_attn_bwd_pre": (c_int, [P, c_long, P, c_long, P, c_long, P, P, P, P, c_long, c_long, c_int, P, P, P, c_int,
                                c_int, c_int, c_int, c_int, c_float, P, P, c_long, P]),
    "sf_attn_bwd_diag": (c_int, [P, c_long, P, c_long, P, c_long, P, P, P, c_int, P, P, c_long, c_int, P, P, c_long, c_int, P, P, P, c_long,
                                 P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_float, P]),
    "sf_attn_bwd_dq": (c_int, [P, c_long, P, c_long, P, c_long, P, c_long, P, P, P, P, P, c_long, c_int, c_int,
                               c_int, c_int, c_int, c_float, P]),
    "sf_attn_bwd_dkv_workspace_floats": (c_long, [c_int, c_int, c_int, c_int, c_int]),
    "sf_attn_bwd_dkv": (c_int, [P, c_long, P, c_long, P, c_long, P, c_long, P, P, P, P, P, c_long, c_int, c_int,
                                c_int, c_int, c_int, c_float, P, c_long, P]),
    "sf_grad_norm_workspace_floats": (c_long, []),
    "sf_grad_norm": (c_int, [P, c_int, c_long, c_float, P, P, P]),
    "sf_adamw_step": (c_int, [P, c_int, P, P, P, P, c_long, P, c_float, c_float, c_float, c_float, c_float, c_float,
                              c_int, c_float, P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


def _bind(lib):
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load_library(path: str = LIB_PATH):
    """dlopen + bind a libsfhip build; raises if it is missing or lacks a symbol."""
    if not os.path.exists(path):
        raise RuntimeError(
            f"specforge_amd: native library not found at {path}. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback."
        )
    return _bind(ctypes.CDLL(path))


def lib():
    global _lib, _emulated
    if _lib is None:
        _lib = load_library(LIB_PATH)
        _emulated = bool(_lib.sf_is_emulated())
        if _emulated:
            raise RuntimeError("specforge_amd/libsfhip.so is an emulator build; refusing to use it as the product library")
    return _lib


def is_emulated() -> bool:
    return _emulated


def _inject_library_for_tests(path):
    """TEST ONLY: route the C-ABI to another build (the SIMT interpreter)."""
    global _lib, _emulated
    if path is None:
        _lib, _emulated = None, False
        return None
    _lib = load_library(path)
    _emulated = bool(_lib.sf_is_emulated())
    return _lib


class SfError(RuntimeError):
    pass


def check(status: int, what: str = ""):
    if status != 0:
        msg = lib().sf_last_error()
        raise SfError(f"{what} failed (status {status}): {msg.decode() if msg else '?'}")
