"""4-wave persistent kernel vs the 8-wave ping-pong kernel on grids below one tile per CU (DeepSeek-V3 dims at batch 1: 2048 tokens x
H 7168 = 8 x 28 = 224 tiles on 256 CUs).  Run once per setting on the TOOLS build (the knob is read once per process):

    SF_GEMM_W4=0 python tools/w4_small_ab.py      (ping-pong kernel wherever the 256-tile kernels apply)
    SF_GEMM_W4=1 python tools/w4_small_ab.py      (4-wave kernel wherever the 256-tile kernels apply)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
from specforge_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(2048, 7168, 7168), (2048, 7168, 40960), (2048, 7168, 32000), (2048, 7168, 9216), (2048, 9216, 7168), (2048, 7168, 21504),
          (3072, 4096, 4096), (3072, 4096, 14336), (3072, 4096, 32000), (2560, 4096, 4096), (3584, 4096, 14336), (2304, 4096, 32000),
          (1536, 7168, 7168), (2048, 5120, 4096)]
for M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm_nt(a, b, c)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ops.gemm_nt(a, b, c)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(json.dumps(dict(M=M, N=N, K=K, tiles=((M + 255) // 256) * ((N + 255) // 256), ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1),
                          w4=os.environ.get("SF_GEMM_W4", "default"))), flush=True)
