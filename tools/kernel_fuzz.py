"""Randomised sweep of the HBM-bound kernels against their references: fused CE (loss, in-place gradient, argmax / accuracy /
acceptance artefacts), RMSNorm forward / backward (+ residual gradient, weight-gradient column sums), the teacher reduction on permuted
logits and the teacher-head GEMM with its reduction epilogue -- the bodies of tests/test_kernels.py over seeded random shapes
(row counts that are not multiples of anything, vocabularies that are multiples of 8 only, every TTT offset, hidden sizes 8 ... 8192).

    python tools/kernel_fuzz.py [--cases 300] [--seed 0]          (GPU box; one JSON line per failure, a summary line last)
    python tools/kernel_fuzz.py --emu --cases 20                  (this container: the SIMT interpreter, small shapes)
"""
import argparse
import json
import os
import random
import sys
import time
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_kernels as TK  # noqa: E402


def draw(rng, emu):
    kind = rng.choice(["ce", "ce", "norm", "norm", "teacher_reduce", "teacher_gemm"])
    f32 = rng.random() < 0.35
    dtype, = (torch.float32,) if f32 else (torch.bfloat16,)
    if kind == "ce":
        B, S = rng.choice([1, 2, 3]), rng.randint(5, 40 if emu else 300)
        V = 8 * rng.randint(2, 64 if emu else 5000)
        T = rng.randint(1, 7)
        return kind, dict(dtype=dtype, tol=1e-4 if f32 else 2e-2, B=B, S=S, V=V, T=T, off=rng.randint(0, T))
    if kind == "norm":
        H = 8 * rng.randint(1, 1024)
        R = rng.randint(1, 20 if emu else 600)
        return kind, dict(dtype=dtype, tol=1e-5 if f32 else 2e-2, R=R, H=H)
    if kind == "teacher_reduce":
        Vd = 8 * rng.randint(6, 40 if emu else 400)          # (the test plants ties at draft columns 2, 3, 7 and rest columns 0 ... 40)
        Vt = Vd + 8 * rng.randint(6, 60 if emu else 2000)
        return kind, dict(dtype=dtype, Vt=Vt, Vd=Vd)
    # (the test asserts WHICH kernel the interpreter build dispatches to: there only shapes its 4-wave kernel takes, M >= 192 and K >= 512)
    M = rng.randint(192, 300) if emu else rng.randint(8, 2200)
    Vd = 8 * rng.randint(4, 40 if emu else 300)
    Vt = Vd + (0 if rng.random() < 0.1 else 8 * rng.randint(40, 100 if emu else 3000))
    K = 64 * (rng.randint(8, 9) if emu else rng.randint(1, 16))
    return kind, dict(M=M, Vt=Vt, Vd=Vd, K=K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--emu", action="store_true")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    from specforge_amd import _lib
    if args.emu:
        from specforge_amd import build
        _lib._inject_library_for_tests(build.build_emu())
        dev = "cpu"
    else:
        _lib.lib()
        dev = "cuda"
    fns = dict(ce=TK.test_ce_fused, norm=TK.test_rmsnorm_fwd_bwd, teacher_reduce=TK.test_teacher_reduce_perm,
               teacher_gemm=TK.test_gemm_nt_teacher_matches_the_stored_logits)
    t0 = time.time()
    fails, count = 0, {}
    for i in range(args.cases):
        kind, kw = draw(rng, args.emu)
        count[kind] = count.get(kind, 0) + 1
        try:
            fns[kind](dev, **kw)
        except Exception as e:
            fails += 1
            tb = traceback.format_exc().strip().splitlines()
            print(json.dumps(dict(i=i, kind=kind, case={k: str(v) for k, v in kw.items()}, error=f"{type(e).__name__}: {e}"[:700],
                                  where=[x.strip()[:160] for x in tb[-5:-1]])), flush=True)
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, failures=fails, by_kind=count, seconds=round(time.time() - t0, 1))), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
