"""Same-process A/B of the weight-gradient (TN) GEMM's launcher knobs on the TOOLS build (tools/experiments/libsfhip_ablate.so: SF_* knobs are
getenv lookups there) on the step's weight-gradient shapes: pace-keeping interval (SF_GEMM_TN_SYNC), tile-group height (SF_GEMM_GM), which
operand is released first (SF_GEMM_TN_PLAN).  Interleaved rounds, random bf16 operands, outputs compared bit for bit with the base's.
    python tools/tn_ab.py [VAR=VAL[,VAR=VAL] ...]        (GPU box)"""
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
variants = ["base"] + (sys.argv[1:] or ["SF_GEMM_TN_SYNC=64", "SF_GEMM_TN_SYNC=256", "SF_GEMM_TN_SYNC=512", "SF_GEMM_GM=2", "SF_GEMM_GM=8", "SF_GEMM_TN_PLAN=0", "SF_GEMM_TN_PLAN=1"])
# (name, M = out features, N = in features, K = token rows)
SHAPES = [("lm_head wgrad", 32000, 4096, 114688), ("gate|up wgrad", 28672, 4096, 114688), ("down wgrad", 4096, 14336, 114688), ("qkv wgrad", 6144, 4096, 114688),
          ("o wgrad", 4096, 4096, 114688)]
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


for name, M, N, K in SHAPES:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ws = torch.empty(2 * M * N + 8192, device=dev, dtype=torch.float32)
    res = {v: [] for v in variants}
    ref, bad = None, []
    for r in range(ROUNDS + 1):
        for v in variants:
            setenv(v)
            t = timed(lambda: ops.gemm_tn(a, b, c, workspace=ws))
            if r:
                res[v].append(t)
            elif ref is None:
                ref = c.clone()
            elif not torch.equal(ref, c):
                bad.append(v)
    fl = 2.0 * M * N * K
    med = {v: statistics.median(ts) for v, ts in res.items()}
    row = dict(name=name, shape=[M, N, K], tflops={v: round(fl / t / 1e9, 1) for v, t in med.items()}, ms={v: round(t, 3) for v, t in med.items()})
    if bad:
        row["MISMATCH"] = bad
    print(json.dumps(row), flush=True)
    del a, b, c, ws
