"""Same-process A/B of the NT GEMM's bf16 epilogue forms on the TOOLS build (tools/experiments/libsfhip_ablate.so: SF_* knobs are getenv
lookups there): register-transposed 16-byte stores (SF_GEMM_EPI_DIRECT=1, the product's form since round 6) against the LDS-staged whole
lines (=0; a residual then takes the general store, as it did before).  Interleaved rounds, random bf16 operands; every variant's output
is compared bit for bit with the first one's.   python tools/epi_ab.py [--shapes headline|batch1|all]      (GPU box)"""
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
args = sys.argv[1:]
which = "all"
if "--shapes" in args:
    i = args.index("--shapes")
    which = args[i + 1]
    del args[i:i + 2]
VARIANTS = ["SF_GEMM_EPI_DIRECT=0", "SF_GEMM_EPI_DIRECT=1"]
ROUNDS = 7
# (form, M, N, K, batch rows S for the row-addend form)
HEADLINE = [("plain", 16384, 4096, 4096, 0), ("residual", 16384, 4096, 4096, 0), ("residual", 16384, 4096, 14336, 0), ("plain", 16384, 32000, 4096, 0),
            ("plain", 16384, 4096, 32000, 0), ("plain", 16384, 4096, 28672, 0), ("plain", 16384, 4096, 6144, 0), ("rowadd", 16384, 6144, 4096, 2048),
            ("swiglu_bwd", 16384, 14336, 4096, 0)]
BATCH1 = [("plain", 4096, 32000, 2048, 0), ("residual", 4096, 2048, 4096, 0), ("residual", 4096, 2048, 12288, 0), ("rowadd", 4096, 5120, 2048, 4096),
          ("swiglu_bwd", 4096, 12288, 2048, 0), ("residual", 4096, 4096, 4096, 0), ("residual", 4096, 4096, 14336, 0), ("plain", 4096, 32000, 4096, 0),
          ("rowadd", 4096, 6144, 4096, 4096), ("swiglu_bwd", 4096, 14336, 4096, 0)]
SHAPES = HEADLINE if which == "headline" else BATCH1 if which == "batch1" else HEADLINE + BATCH1


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_GEMM_"):
            del os.environ[k]
    for kv in v.split(","):
        k, val = kv.split("=")
        os.environ[k] = val


def timed(fn, iters=4):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for form, M, N, K, S in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    if form in ("plain", "residual"):
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        r_ = torch.randn(M, N, device=dev).to(torch.bfloat16) if form == "residual" else None
        fn, out = (lambda: ops.gemm_nt(a, b, c, residual=r_)), c
    elif form == "rowadd":
        T = 7
        add = torch.randn(M // S * (S + T), N, device=dev)
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn, out = (lambda: ops.gemm_nt_rowadd(a, b, c, add, S=S, Spad=S + T, off=3)), c
    else:
        gu = torch.randn(M, 2 * N, device=dev).to(torch.bfloat16)
        dgu = torch.empty(M, 2 * N, device=dev, dtype=torch.bfloat16)
        dact = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn, out = (lambda: ops.gemm_nt_swiglu_bwd(a, b, gu, dgu, dact)), dgu
    res = {v: [] for v in VARIANTS}
    ref = None
    bad = []
    for r in range(ROUNDS + 1):
        for v in VARIANTS:
            setenv(v)
            t = timed(fn)
            if r:
                res[v].append(t)
            else:
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    bad.append(v)
    fl = 2.0 * M * N * K
    med = {v: statistics.median(ts) for v, ts in res.items()}
    row = dict(form=form, shape=[M, N, K], tiles256=((M + 255) // 256) * ((N + 255) // 256),
               tflops={v.split("=")[1]: round(fl / t / 1e9, 1) for v, t in med.items()},
               ms={v.split("=")[1]: round(t, 4) for v, t in med.items()}, direct_over_staged=round(med[VARIANTS[0]] / med[VARIANTS[1]], 4))
    if bad:
        row["MISMATCH"] = bad
    print(json.dumps(row), flush=True)
