import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session")
def emu_lib_path():
    """Build (once) the SIMT-interpreter flavour of the kernels: tests/emu/libsfhip_emu.so."""
    from specforge_amd import build

    return build.build_emu()


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request, emu_lib_path):
    """'emu': kernels run under the SIMT interpreter on CPU tensors (index-logic check, no GPU).
    'gpu': the product library libsfhip.so on cuda:0 -- the parity tests proper."""
    from specforge_amd import _lib

    if request.param == "emu":
        _lib._inject_library_for_tests(emu_lib_path)
        yield "cpu"
        _lib._inject_library_for_tests(None)
    else:
        import torch

        assert torch.cuda.is_available(), "-m gpu tests need a GPU"
        _lib._inject_library_for_tests(None)
        _lib.lib()  # raises if libsfhip.so is missing: no fallback
        yield "cuda"
