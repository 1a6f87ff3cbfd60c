"""Within-process A/B of TN (weight-gradient) GEMM variants of the TOOLS build on the step's shapes; see tools/gemm_ab.py.
python tools/gemm_tn_ab.py SF_GEMM_TN_PLAN=-1 SF_GEMM_TN_PLAN=0 SF_GEMM_TN_PLAN=1"""
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_lib._inject_library_for_tests(os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._emulated = False
dev = "cuda"
variants = sys.argv[1:] or ["base"]
K = 7 * 16384
SHAPES = [(32000, 4096, K), (28672, 4096, K), (4096, 14336, K), (6144, 4096, K), (4096, 4096, K), (4096, 12288, 16384)]
ROUNDS = 4


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_GEMM_"):
            del os.environ[k]
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val


def timed(fn, iters=2):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for (M, N, Kk) in SHAPES:
    a = torch.randn(Kk, M, device=dev).to(torch.bfloat16)
    b = torch.randn(Kk, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(2 * M * N + 4096, device=dev)
    ref = None
    res = {v: [] for v in variants}
    for r in range(ROUNDS + 1):
        for v in variants:
            setenv(v)
            t = timed(lambda: ops.gemm_tn(a, b, c, workspace=ws))
            if r:
                res[v].append(t)
            elif ref is None:
                ref = c.clone()
            elif not torch.equal(ref, c):
                print(json.dumps(dict(shape=[M, N, Kk], variant=v, MISMATCH=float((ref.float() - c.float()).abs().max()))))
    fl = 2.0 * M * N * Kk
    print(json.dumps(dict(shape=[M, N, Kk], **{v: round(fl / statistics.median(ts) / 1e9, 1) for v, ts in res.items()})), flush=True)
    del a, b, c, ws
