"""Randomised parity sweep of the TTT attention kernels (forward, dQ, dK/dV, diagonal backward) against the oracle on the GPU box:
the body of tests/test_attention.py::test_ttt_attention_fwd_bwd over seeded random shapes -- batch, sequence length (any value, tile
multiples or not), ragged valid lengths, head counts / GQA ratios, 1 ... 7 TTT steps, head_dim 64 / 128 / 256.

    python tools/attn_fuzz.py [--cases 300] [--seed 0] [--max-seq 700]          (GPU box; one JSON line per failure, a summary line last)

The fixed cases of the test suite are the regression set; this is the search for shapes nobody thought of (found by its first runs, if
anything: see profiles/README.md).  The oracle runs on the GPU as well (fp32 [B, nh, S, S + k] scores), so a case takes ~50 ms.
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
import test_attention as TA  # noqa: E402


def draw(rng, max_seq):
    hd = rng.choice([64, 128, 128, 256, 256])
    nkv = rng.choice([1, 1, 2, 2, 3, 4, 8])
    rep = rng.choice([1, 2, 4, 7, 8]) if nkv <= 2 else rng.choice([1, 2, 4])
    nh = nkv * rep
    B = rng.choice([1, 1, 2, 3])
    kind = rng.random()
    if kind < 0.15:
        S = rng.randint(1, 40)                       # shorter than one tile
    elif kind < 0.35:
        S = rng.choice([32, 64, 128, 256, 512]) + rng.choice([-1, 0, 0, 1])     # around the tile edges
    else:
        S = rng.randint(41, max_seq)
    while B * nh * S * (S + 8) * 4 * 6 > 3e9:        # oracle scores + autograd copies: keep a case under ~3 GB
        S = max(1, S // 2)
    lengths = [S if rng.random() < 0.4 else rng.randint(1, S) for _ in range(B)]
    if rng.random() < 0.5:
        lengths[0] = S                               # the collator pads to the longest sample
    nsteps = rng.choice([1, 2, 3, 4, 5, 6, 7, 7])
    return hd, B, S, nh, nkv, lengths, nsteps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-seq", type=int, default=700)
    ap.add_argument("--emu", action="store_true", help="dry run of this script under the SIMT interpreter on the CPU (slow: keep --max-seq small)")
    args = ap.parse_args()
    rng = random.Random(args.seed)
    from specforge_amd import _lib
    if args.emu:
        from specforge_amd import build
        _lib._inject_library_for_tests(build.build_emu())
        dev = "cpu"
    else:
        orig = TA._oracle
        TA._oracle = lambda *a, **k: orig(*a, **{**k, "device": "cuda"})      # the restatement itself, on the GPU
        _lib.lib()
        dev = "cuda"
    t0 = time.time()
    fails, by_hd = 0, {}
    for i in range(args.cases):
        hd, B, S, nh, nkv, lengths, nsteps = draw(rng, args.max_seq)
        case = dict(i=i, hd=hd, B=B, S=S, nh=nh, nkv=nkv, lengths=lengths, nsteps=nsteps)
        by_hd[hd] = by_hd.get(hd, 0) + 1
        try:
            TA.test_ttt_attention_fwd_bwd(dev, hd, B, S, nh, nkv, lengths, nsteps)
            if dev == "cuda":
                torch.cuda.synchronize()
        except Exception as e:      # AssertionError from the comparison, RuntimeError from a launcher
            fails += 1
            case["error"] = f"{type(e).__name__}: {e}"[:600]
            case["where"] = traceback.format_exc().strip().splitlines()[-3][:200]
            print(json.dumps(case), flush=True)
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, max_seq=args.max_seq, failures=fails, by_head_dim=by_hd,
                          seconds=round(time.time() - t0, 1))), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
