"""Same-process A/B of the pair-resident 256 x 128 GEMM (csrc/sf_gemm_p2.hip) against the product dispatch on the TOOLS build
(tools/experiments/libsfhip_ablate.so: SF_* knobs are getenv lookups there), interleaved rounds, random bf16 operands; every variant's
output is compared bit for bit with the base's.   python tools/p2_ab.py [--shapes headline|batch1|all] [variant ...]      (GPU box)
A variant is a comma-separated set of knobs, e.g. SF_P2_STAGGER_MODE=2,SF_P2_STAGGER_PCT=50; SF_GEMM_P2=7 is implied."""
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
variants = args or ["SF_P2_STAGGER_MODE=1", "SF_P2_STAGGER_MODE=0"]
ROUNDS = 5
# (form, M, N, K, batch rows S for the row-addend form)
HEADLINE = [("plain", 16384, 4096, 4096, 0), ("plain", 16384, 32000, 4096, 0), ("plain", 16384, 4096, 14336, 0), ("plain", 16384, 4096, 32000, 0),
            ("plain", 16384, 4096, 28672, 0), ("rowadd", 16384, 6144, 4096, 2048), ("swiglu_bwd", 16384, 14336, 4096, 0)]
BATCH1 = [("plain", 4096, 32000, 2048, 0), ("plain", 4096, 2048, 4096, 0), ("plain", 4096, 2048, 12288, 0), ("plain", 4096, 2048, 32000, 0),
          ("plain", 4096, 2048, 24576, 0), ("rowadd", 4096, 5120, 2048, 4096), ("swiglu_bwd", 4096, 12288, 2048, 0),
          ("plain", 4096, 4096, 4096, 0), ("plain", 4096, 32000, 4096, 0), ("plain", 4096, 4096, 14336, 0), ("rowadd", 4096, 6144, 4096, 4096),
          ("swiglu_bwd", 4096, 14336, 4096, 0), ("plain", 2048, 7168, 7168, 0), ("rowadd", 2048, 9216, 7168, 2048), ("swiglu_bwd", 2048, 40960, 7168, 0)]
SHAPES = HEADLINE if which == "headline" else BATCH1 if which == "batch1" else HEADLINE + BATCH1


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_GEMM_") or k.startswith("SF_P2_"):
            del os.environ[k]
    if v != "base":
        os.environ["SF_GEMM_P2"] = "7"
        for kv in v.split(","):
            if kv:
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


for form, M, N, K, S in SHAPES:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    if form == "plain":
        c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn, out = (lambda: ops.gemm_nt(a, b, c)), c
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
    res = {v: [] for v in ["base"] + variants}
    ref = None
    bad = []
    for r in range(ROUNDS + 1):
        for v in ["base"] + variants:
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
    row = dict(form=form, shape=[M, N, K], tiles256=((M + 255) // 256) * ((N + 255) // 256),
               **{v: round(fl / statistics.median(ts) / 1e9, 1) for v, ts in res.items()},
               ms={v: round(statistics.median(ts), 4) for v, ts in res.items()})
    if bad:
        row["MISMATCH"] = bad
    print(json.dumps(row), flush=True)
