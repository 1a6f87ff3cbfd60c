"""Under-filled NT GEMMs (fewer 256 x 256 tiles than CUs: the bs 1 recipes -- cfg 4 at bs 1 x 4096 has N_tokens = 4096, H = 2048:
down / o / lm_head-dgrad are 16 x 8 = 128 tiles on 256 CUs): 256-tile kernels vs the 128 x 128 kernel (4 x the workgroups).
Run once per setting on the TOOLS build (the knob is read once per process):

    python tools/small_tile_ab.py            > gpurun_out/small_tile_256.jsonl
    SF_GEMM_TILE=128 python tools/small_tile_ab.py > gpurun_out/small_tile_128.jsonl
    SF_NT_WS=1 python tools/small_tile_ab.py > gpurun_out/small_tile_splitk.jsonl      (product library, sf_gemm_nt_ws: split-K where it qualifies)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib  # noqa: E402

if os.environ.get("SF_GEMM_TILE"):
    _lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
    _lib._emulated = False
from specforge_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [(4096, 2048, 12288), (4096, 2048, 4096), (4096, 2048, 32000), (4096, 2048, 5120), (4096, 2048, 6144),
          (2048, 7168, 40960), (2048, 7168, 7168), (2048, 7168, 32000), (2048, 4096, 4096), (4096, 4096, 4096), (8192, 2048, 12288),
          (6144, 2048, 12288), (5120, 2048, 12288)]
for M, N, K in SHAPES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev, generator=g).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ws = torch.empty(4 * M * N, device=dev) if os.environ.get("SF_NT_WS") else None
    for _ in range(5):
        ops.gemm_nt(a, b, c, workspace=ws)
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n):
        ops.gemm_nt(a, b, c, workspace=ws)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    print(json.dumps(dict(M=M, N=N, K=K, tiles256=((M + 255) // 256) * ((N + 255) // 256), ms=round(ms, 4),
                          tflops=round(2.0 * M * N * K / ms / 1e9, 1), tile=os.environ.get("SF_GEMM_TILE", "product dispatch" + (" + workspace (split-K)" if ws is not None else "")))), flush=True)
