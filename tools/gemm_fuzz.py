"""Randomised sweep of the GEMM entry points over the shapes where the launchers' dispatch changes (128 x 128 tiles / 8-wave ping-pong /
4-wave persistent kernel / split-K / peeled last round / shifted last row tile / fused SwiGLU epilogues / TN split-K and pace-keeping):
M, N, K drawn so that the 256 x 256 tile count lands below, at and above 128, 256 and multiples of 256, with ragged edges.

    python tools/gemm_fuzz.py [--cases 250] [--seed 0]            (GPU box; one JSON line per failure, a summary line last)

Reference: torch's fp32 matmul of the bf16-rounded operands on the same GPU (a different GEMM: rocBLAS fp32), tolerances of
tests/test_kernels.py (2e-2 * sqrt(K) absolute for bf16 outputs, 1e-3 * sqrt(K) for fp32).  The fused SwiGLU forms are compared with the
library's own unfused sequence bit for bit, like their tests.
"""
import argparse
import json
import math
import os
import random
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from specforge_amd import ops  # noqa: E402

DEV = "cuda"
SHRINK = 1          # --emu: dimensions divided by this (the interpreter runs a 256 x 256 x 512 tile in about a second)


def rnd(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV) * scale).to(dtype)


def dim(rng, lo=8, hi=9000):
    """a dimension near a tile-count threshold more often than not"""
    r = rng.random()
    if r < 0.5:
        t = rng.choice([1, 2, 3, 4, 5, 8, 11, 12, 16, 17, 23, 24, 32, 33])
        v = 256 * t + rng.choice([-248, -128, -8, 0, 0, 8, 64, 136])
    elif r < 0.7:
        v = rng.choice([8, 16, 48, 64, 120, 128, 192, 200, 256, 264])
    else:
        v = rng.randint(lo, hi)
    return max(lo, min(hi, v // SHRINK // 8 * 8))


def kdim(rng):
    r = rng.random()
    if r < 0.5:
        return max(64, 64 * rng.choice([1, 2, 3, 4, 7, 8, 9, 16, 24, 32, 33, 48, 64]) // SHRINK // 64 * 64)
    return max(8, rng.randint(8, 512) // SHRINK) * 8


def close(got, ref, K, f32, what):
    tol = 1e-3 if f32 else 2e-2
    err = (got.float() - ref).abs()
    bound = tol * ref.abs() + tol * math.sqrt(K)
    bad = err > bound
    if bool(bad.any()):
        i = int(torch.argmax((err - bound).flatten()))
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements off; worst at flat index {i}: got {float(got.flatten()[i])}, "
                             f"ref {float(ref.flatten()[i])}")


def case_nt(rng, seed):
    M, N, K = dim(rng), dim(rng), kdim(rng)
    while M * N > 40e6:
        M = max(8, M // 2 // 8 * 8)
    f32 = rng.random() < 0.3
    a, b = rnd((M, K), seed), rnd((N, K), seed + 1)
    ref = a.float() @ b.float().t()
    mode = rng.choice(["plain", "ws", "residual", "alpha_beta", "ws_residual"]) if not f32 else rng.choice(["plain", "ws", "alpha_beta"])
    # output as a strided view of a wider buffer (the fused q|k|v block) half of the time
    pad = rng.choice([0, 8, 24])
    wide = torch.full((M, N + 2 * pad), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    out = wide[:, pad:pad + N] if pad else wide
    ws = torch.empty(8 * ((M + 255) // 256 * 256) * N + 4096, dtype=torch.float32, device=DEV) if "ws" in mode else None
    desc = dict(kind="nt", M=M, N=N, K=K, f32=f32, mode=mode, pad=pad)
    if mode in ("residual", "ws_residual"):
        res = rnd((M, N), seed + 2)
        ops.gemm_nt(a, b, out, residual=res, workspace=ws)
        ref = ref + res.float()
        if ws is None:      # round(round(a . b^T) + residual): bit for bit the plain output followed by a bf16 add, whichever epilogue a tile takes
            plain = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            ops.gemm_nt(a, b, plain)
            assert torch.equal(out, plain + res), (desc, "residual form != plain output + residual")
    elif mode == "alpha_beta":
        ops.gemm_nt(a, b, out, alpha=0.5, beta=2.0)
        ref = 0.5 * ref + 14.0
    else:
        ops.gemm_nt(a, b, out, workspace=ws)
    close(out, ref, K, f32, desc)
    if pad:
        assert float((wide[:, :pad].float() - 7.0).abs().max()) == 0.0 and float((wide[:, pad + N:].float() - 7.0).abs().max()) == 0.0, (desc, "wrote outside its columns")
    return desc


def case_rowadd(rng, seed):
    S = rng.choice([8, 16, 24, 64, 100, 256, 512, 1000])
    Bn = rng.randint(1, max(1, min(12, 6000 // S)))
    T = rng.randint(1, 7)
    M, N, K = Bn * S, dim(rng, 8, 7000), kdim(rng)
    f32 = rng.random() < 0.3
    a, b = rnd((M, K), seed), rnd((N, K), seed + 1)
    Spad = S + T
    add = rnd((Bn * Spad, N), seed + 2, scale=3.0, dtype=torch.float32)
    off = rng.randint(0, T)
    rows = (torch.arange(M, device=DEV) // S) * Spad + torch.arange(M, device=DEV) % S + off
    ref = a.float() @ b.float().t() + add[rows]
    out = torch.full((M, N), 7.0, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    use_ws = rng.random() < 0.5
    ws = torch.empty(8 * ((M + 255) // 256 * 256) * N + 4096, dtype=torch.float32, device=DEV) if use_ws else None
    desc = dict(kind="rowadd", M=M, N=N, K=K, S=S, T=T, off=off, f32=f32, ws=use_ws)
    ops.gemm_nt_rowadd(a, b, out, add, S=S, Spad=Spad, off=off, workspace=ws)
    close(out, ref, K, f32, desc)
    return desc


def case_tn(rng, seed):
    M, N = dim(rng, 8, 6000), dim(rng, 8, 6000)
    K = 64 * rng.choice([1, 2, 3, 8, 16, 64, 65, 128, 256, 300, 512])
    while M * N > 30e6:
        M = max(8, M // 2 // 8 * 8)
    f32 = rng.random() < 0.4
    a, b = rnd((K, M), seed), rnd((K, N), seed + 1)
    ref = a.float().t() @ b.float()
    use_ws = rng.random() < 0.6
    ws = torch.empty(2 * M * N + 4096, dtype=torch.float32, device=DEV) if use_ws else None
    ab = rng.random() < 0.4
    pad = rng.choice([0, 8])
    wide = torch.ones(M, N + 2 * pad, dtype=torch.float32 if f32 else torch.bfloat16, device=DEV)
    out = wide[:, pad:pad + N] if pad else wide
    desc = dict(kind="tn", M=M, N=N, K=K, f32=f32, ws=use_ws, alpha_beta=ab, pad=pad)
    if ab:
        ops.gemm_tn(a, b, out, alpha=0.5, beta=2.0, workspace=ws)
        ref = 0.5 * ref + 2.0
    else:
        ops.gemm_tn(a, b, out, workspace=ws)
    close(out, ref, K, f32, desc)
    if pad:
        assert float((wide[:, :pad].float() - 1.0).abs().max()) == 0.0 and float((wide[:, pad + N:].float() - 1.0).abs().max()) == 0.0, (desc, "wrote outside its columns")
    return desc


def case_swiglu(rng, seed):
    M = dim(rng, 8, 6000)
    I = rng.choice([128, 192, 256, 384, 512, 1000, 1024, 1536, 2048, 2176, 3072])
    K = kdim(rng)
    x, wgu = rnd((M, K), seed, 0.5), rnd((2 * I, K), seed + 1, 0.1)
    gu, act = torch.empty(M, 2 * I, dtype=torch.bfloat16, device=DEV), torch.empty(M, I, dtype=torch.bfloat16, device=DEV)
    desc = dict(kind="swiglu", M=M, I=I, K=K)
    ops.gemm_nt_swiglu_fwd(x, wgu, gu, act)
    gu2, act2 = torch.empty_like(gu), torch.empty_like(act)
    ops.gemm_nt(x, wgu, gu2)
    ops.swiglu_fwd(gu2, act2)
    assert torch.equal(gu, gu2) and torch.equal(act, act2), (desc, "fused forward != gemm_nt + swiglu_fwd")
    close(gu, x.float() @ wgu.float().t(), K, False, desc)
    # backward form: d(gate|up) from dy [M, K2] @ w_down[I... ] -- contraction over the model width
    K2 = kdim(rng)
    dy, wd = rnd((M, K2), seed + 2, 0.5), rnd((I, K2), seed + 3, 0.1)
    dgu, dact = torch.empty_like(gu), torch.empty_like(act)
    ops.gemm_nt_swiglu_bwd(dy, wd, gu, dgu, dact)
    dact2, dgu2 = torch.empty_like(act), torch.empty_like(gu)
    ops.gemm_nt(dy, wd, dact2)
    ops.swiglu_bwd(dact2, gu, dgu2)
    desc["K2"] = K2
    assert torch.equal(dgu, dgu2), (desc, "fused backward != gemm_nt + swiglu_bwd")
    return desc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=250)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--emu", action="store_true", help="dry run under the SIMT interpreter on the CPU, dimensions / 12")
    args = ap.parse_args()
    if args.emu:
        global DEV, SHRINK
        from specforge_amd import _lib, build
        _lib._inject_library_for_tests(build.build_emu())
        DEV, SHRINK = "cpu", 12
    rng = random.Random(args.seed)
    kinds = [case_nt, case_nt, case_nt, case_rowadd, case_tn, case_tn, case_swiglu]
    t0 = time.time()
    fails, count = 0, {}
    for i in range(args.cases):
        fn = rng.choice(kinds)
        count[fn.__name__] = count.get(fn.__name__, 0) + 1
        try:
            fn(rng, 100 * i + 1)
            if DEV == "cuda":
                torch.cuda.synchronize()
        except Exception as e:
            fails += 1
            tb = traceback.format_exc().strip().splitlines()
            print(json.dumps(dict(i=i, kind=fn.__name__, error=f"{type(e).__name__}: {e}"[:900], where=[x.strip()[:140] for x in tb[-5:-1]])), flush=True)
            if DEV == "cuda":
                torch.cuda.synchronize()
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, failures=fails, by_kind=count, seconds=round(time.time() - t0, 1))), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
