"""Within-process A/B of the TTT attention kernels (tools build, tools/experiments/libsfhip_ablate.so) at the headline
shape (B 8, S 2048, nh 32, nkv 8, hd 128), interleaved rounds, random bf16 operands; prints one JSON line per
(variant, kernel, ndiag) with the median time.

    python tools/attn_bench.py [VAR=VAL[,VAR=VAL] ...]          (GPU box)   "base" = no knobs

MFMA work per launch: fwd 4, dQ 6, dK/dV 8 (x B*nh*hd*S^2/2); `frac` = achieved / 2.5 PFLOP/s."""
import json
import math
import os
import statistics
import sys

import torch

sys.path.insert(0, ".")
from specforge_amd import _lib, ops  # noqa: E402

_LIB = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--lib=")), os.path.join("tools", "experiments", "libsfhip_ablate.so"))
_lib._inject_library_for_tests(_LIB)       # (--lib=...: another build of the tools library, e.g. other compiler flags)
_lib._emulated = False
dev = "cuda"
args = [a for a in sys.argv[1:] if not a.startswith("--")]
variants = args or ["base"]
ROUNDS = 5
shape = dict(B=8, S=2048, nh=32, nkv=8, hd=128)
for a in sys.argv[1:]:
    if a.startswith("--shape="):
        B, S, nh, nkv, hd = (int(x) for x in a.split("=")[1].split(","))
        shape = dict(B=B, S=S, nh=nh, nkv=nkv, hd=hd)
NDIAG = [0, 6]


def setenv(v):
    for k in list(os.environ):
        if k.startswith("SF_ATTN_"):
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


def main():
    torch.manual_seed(0)
    B, S, nh, nkv, hd = (shape[k] for k in ("B", "S", "nh", "nkv", "hd"))
    N = B * S
    unit = B * nh * hd * S * S / 2.0
    qkv = [torch.randn(N, (nh + 2 * nkv) * hd, device=dev).to(torch.bfloat16) for _ in range(max(NDIAG) + 1)]
    q = qkv[-1][:, :nh * hd]
    kv = [t[:, nh * hd:(nh + nkv) * hd] for t in qkv]
    vv = [t[:, (nh + nkv) * hd:] for t in qkv]
    o = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B, nh, S, device=dev)
    do = torch.randn(N, nh * hd, device=dev).to(torch.bfloat16)
    delta = torch.empty(B, nh, S, device=dev)
    dq_init = torch.zeros(N, nh * hd, device=dev)
    dk = [torch.zeros(N, nkv * hd, device=dev) for _ in range(max(NDIAG) + 1)]
    dv = [torch.zeros(N, nkv * hd, device=dev) for _ in range(max(NDIAG) + 1)]
    dq = torch.empty(N, nh * hd, device=dev, dtype=torch.bfloat16)
    kw = dict(scale=1 / math.sqrt(hd), **shape)
    nws = ops.attn_bwd_dkv_workspace_floats(B, S, nh, nkv, hd)      # (the engine's head-split workspace: small B * nkv)
    dkv_ws = torch.empty(nws, device=dev) if nws else None

    def kernels(nd):
        return {
            "fwd": (lambda: ops.attn_fwd(q, kv[0], vv[0], kv[1:nd + 1], vv[1:nd + 1], None, o, lse, **kw), 4.0),
            "pre": (lambda: ops.attn_bwd_pre(q, o, do, kv[1:nd + 1], vv[1:nd + 1], dk[1:nd + 1], dv[1:nd + 1], lse, delta,
                                             dq_init if nd else None, **kw), 0.0),
            "dq": (lambda: ops.attn_bwd_dq(q, do, kv[0], vv[0], None, lse, delta, dq_init if nd else None, dq, **kw), 6.0),
            "dkv": (lambda: ops.attn_bwd_dkv(q, do, kv[0], vv[0], None, lse, delta, dk[0], dv[0], workspace=dkv_ws, **kw), 8.0),
        }

    res = {}
    for rnd in range(ROUNDS + 1):           # round 0 = warm-up
        for v in variants:
            setenv(v)
            for nd in NDIAG:
                for name, (fn, units) in kernels(nd).items():
                    if nd and name == "dkv":
                        continue             # the diagonal branches only touch fwd, pre and (through dq_init) dq
                    try:
                        ms = timed(fn)
                    except Exception as ex:  # a variant that does not exist in this build
                        ms = float("nan")
                        if rnd == 0:
                            print(json.dumps(dict(variant=v, kernel=name, error=str(ex)[:200])), flush=True)
                    if rnd:
                        res.setdefault((v, name, nd), []).append(ms)
    for (v, name, nd), xs in res.items():
        ms = statistics.median(xs)
        units = dict(fwd=4.0, dq=6.0, dkv=8.0, pre=0.0)[name]
        tf = units * unit / ms / 1e9 if units else None
        print(json.dumps(dict(variant=v, kernel=name, ndiag=nd, ms=round(ms, 4), min_ms=round(min(xs), 4),
                              tflops=None if tf is None else round(tf, 1), frac=None if tf is None else round(tf / 2500.0, 3),
                              **shape)), flush=True)


if __name__ == "__main__":
    main()
