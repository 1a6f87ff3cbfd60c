"""Randomised end-to-end parity sweep: one micro-step (teacher, 1 ... 7 TTT steps forward and backward) of randomly drawn SMALL draft
configurations through the product path (strategy -> engine -> C-ABI kernels) against the oracle -- the body of
tests/test_configs.py::test_odd_dimension_relations_match_oracle over seeded random dimension relations:
hidden / intermediate sizes that are or are not multiples of the tile sizes, head_dim 64 / 128 / 256 with nh * hd == H or != H, GQA ratios
1 ... 8, draft vocabulary == or < target vocabulary, target hidden != draft hidden, fc_norm / norm_output, every RoPE variant the reference
has (default, linear, dynamic, llama3, yarn), sequences longer than max_position_embeddings + 20 (rotary-cache growth), ragged batches,
sparse loss masks, ttt_length 1 ... 7.

    python tools/engine_fuzz.py [--cases 120] [--seed 0]              (GPU box; one JSON line per failure, a summary line last)
    python tools/engine_fuzz.py --emu --cases 6                       (this container: the SIMT interpreter)
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
import test_configs as TC  # noqa: E402


def draw(rng, emu):
    hd = rng.choice([64, 64, 128, 128, 256])
    nkv = rng.choice([1, 1, 2, 2, 4])
    nh = nkv * rng.choice([1, 2, 4, 7, 8] if nkv == 1 else [1, 2, 4])
    H = rng.choice([64, 128, 192, 256, 320, 384]) if rng.random() < 0.7 else nh * hd
    H = min(H, 512)
    Ht = H if rng.random() < 0.6 else rng.choice([64, 128, 192, 256])
    I = rng.choice([128, 168, 192, 256, 320, 384, 520])
    Vt = rng.choice([256, 384, 512, 640, 1000, 1536])
    Vd = Vt if rng.random() < 0.15 else rng.choice([v for v in (64, 128, 192, 256, 384, 504) if v < Vt])
    B = rng.choice([1, 1, 2, 3])
    S = rng.randint(2, 48 if emu else 260)
    lengths = [S if rng.random() < 0.4 else rng.randint(2, S) for _ in range(B)]
    if rng.random() < 0.6:
        lengths[0] = S
    c = dict(H=H, Ht=Ht, I=I, nh=nh, nkv=nkv, hd=hd, Vt=Vt, Vd=Vd, B=B, S=S, ttt=rng.choice([1, 2, 3, 4, 5, 7, 7]), lengths=lengths,
             max_pos=rng.choice([32, 128, 128, 512]), fc_norm=rng.random() < 0.3, norm_output=rng.random() < 0.85,
             mask_keep=rng.choice([1.0, 1.0, 0.7, 0.3]))
    r = rng.random()
    if r < 0.12:
        c["rope_scaling"] = dict(rope_type="linear", factor=2.0)
    elif r < 0.24:
        c["rope_scaling"] = dict(rope_type="dynamic", factor=2.0)
    elif r < 0.36:
        c["rope_scaling"] = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=16)
    elif r < 0.48:
        c["rope_scaling"] = dict(rope_type="yarn", factor=4.0, beta_fast=32, beta_slow=1, mscale=1.0, mscale_all_dim=0.5,
                                 original_max_position_embeddings=32)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=120)
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
    t0 = time.time()
    fails = 0
    for i in range(args.cases):
        c = draw(rng, args.emu)
        try:
            TC.run_small_case(dev, c, seed=10 * i + 1)
        except Exception as e:
            fails += 1
            tb = traceback.format_exc().strip().splitlines()
            print(json.dumps(dict(i=i, case=c, error=f"{type(e).__name__}: {e}"[:700], where=[x.strip()[:160] for x in tb[-6:-1]])), flush=True)
    print(json.dumps(dict(summary=True, cases=args.cases, seed=args.seed, failures=fails, seconds=round(time.time() - t0, 1))), flush=True)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
