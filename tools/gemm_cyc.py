"""Cycles per K-tile of the NT GEMM main loop, read from the kernel itself (TOOLS build: SF_GEMM_CYC=1 makes wave 0 of every
workgroup store its s_memtime delta over the steady-state loop -- nkt - 2 iterations -- into C at its tile origin).
Clock-independent, unlike TFLOP/s under a power cap.    python tools/gemm_cyc.py [VAR=VAL[,VAR=VAL]] ...   (GPU box)
128 MFMA 16x16x32 per K-tile per wave = 2048 cycles when the matrix pipe never waits (16 cycles each)."""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
dev = "cuda"
variants = sys.argv[1:] or ["base"]
SHAPES = [(16384, 4096, 14336), (16384, 32000, 4096), (16384, 4096, 4096)]


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_GEMM_"):
            del os.environ[k]
    os.environ["SF_GEMM_CYC"] = "1"
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val


for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    row = dict(shape=[M, N, K])
    for v in variants:
        setenv(v)
        for _ in range(3):
            ops.gemm_nt(a, b, c)
        torch.cuda.synchronize()
        cyc = c.view(torch.float32)[::256, ::128].float().cpu()      # tile origins
        per = cyc / (K // 64 - 2)
        row[v] = dict(mean=round(float(per.mean()), 1), min=round(float(per.min()), 1), max=round(float(per.max()), 1))
    print(json.dumps(row), flush=True)
