"""bench.py on the TOOLS build (tools/experiments/libsfhip_ablate.so), so that SF_* knobs apply to the whole training step:
same-box A/B of a kernel variant inside the real step.   SF_GEMM_PERSIST=0 python tools/bench_ab.py --steps 10 --warmup 3 ..."""
import os
import sys

import torch  # noqa: F401  (first: the library must bind to the HIP runtime torch loads)

sys.path.insert(0, ".")
from specforge_amd import _lib  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
import bench  # noqa: E402

bench.main()
