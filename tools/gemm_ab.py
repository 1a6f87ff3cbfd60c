"""Within-process A/B of NT GEMM schedule variants of the TOOLS build (tools/experiments/libsfhip_ablate.so) on the step's
shapes, interleaved rounds, random bf16 operands; prints medians.   python tools/gemm_ab.py VAR=VAL[,VAR=VAL] ...  (GPU box)
Each argument is one variant = a set of SF_* knobs; "base" = no knobs."""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, build, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
dev = "cuda"
variants = sys.argv[1:] or ["base"]
SHAPES = [(16384, 4096, 4096), (16384, 4096, 14336), (16384, 28672, 4096), (16384, 32000, 4096), (16384, 4096, 32000),
          (16384, 14336, 4096), (8192, 8192, 8192), (16384, 4096, 6144), (16384, 4096, 12288), (4096, 128256, 4096),
          (16384, 4096, 28672)]
ROUNDS = 5


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_GEMM_"):
            del os.environ[k]
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val


def timed(fn, iters=3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (M, N, K) in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ref = None
    res = {v: [] for v in variants}
    res["lib"] = []
    for r in range(ROUNDS + 1):
        for v in variants:
            setenv(v)
            t = timed(lambda: ops.gemm_nt(a, b, c))
            if r:
                res[v].append(t)
            if r == 0:
                if ref is None:
                    ref = c.clone()
                elif not torch.equal(ref, c):
                    print(json.dumps(dict(shape=[M, N, K], variant=v, MISMATCH=float((ref.float() - c.float()).abs().max()))))
        t = timed(lambda: torch.matmul(a, b.t(), out=c))
        if r:
            res["lib"].append(t)
    fl = 2.0 * M * N * K
    print(json.dumps(dict(shape=[M, N, K], **{v: round(fl / statistics.median(ts) / 1e9, 1) for v, ts in res.items()})), flush=True)
