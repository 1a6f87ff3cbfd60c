"""bench.py on another PRODUCT build of the library (e.g. one built from an earlier commit: tools/experiments/libsfhip_old.so), so that two
commits can be compared inside the real step on one box.   python tools/bench_lib.py <path.so> --steps 10 --warmup 3 ...   (GPU box)"""
import os
import sys

import torch  # noqa: F401

sys.path.insert(0, ".")
from specforge_amd import _lib  # noqa: E402

_lib._inject_library_for_tests(sys.argv.pop(1))
_lib._emulated = False
import bench  # noqa: E402

bench.main()
