"""Soak run: hundreds of optimizer steps at the headline dimensions over batches whose shape changes EVERY step (batch 1 ... 8, any sequence
length up to 2048, ragged valid lengths, sparse loss masks with host-known row counts on every other step, gradient accumulation windows of
1 ... 3 micro-steps), watching what a long training run would trip over and a 50-step benchmark cannot see: device memory that keeps
growing (arena re-sizing, cached per-shape state), host memory that keeps growing (pinned buffers, ctypes temporaries), a loss or gradient
norm that stops being finite, a device-side input check that fires.

    python tools/soak.py [--steps 400] [--config llama3-8b] [--seed 0]          (GPU box; one JSON line per 50 steps, a verdict line last)
"""
import argparse
import json
import math
import os
import random
import sys
import time

import psutil
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from specforge_amd import _lib  # noqa: E402
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch, loss_mask_suffix_counts  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402


def judge(samples, steps, bad):
    """numbers finite throughout; the memory FLOOR of the second half of the run not above the first half's.  (A sample includes what the
    engine keeps of the last batch until the next one arrives -- compacted teacher inputs, artefacts: up to 0.7 GB at DeepSeek-V3 dims,
    proportional to that batch's size -- so samples scatter by that much; a leak lifts the floor, the scatter does not.)"""
    h1 = [s for s in samples if s["step"] <= steps // 2] or samples[:1]
    h2 = [s for s in samples if s["step"] > steps // 2] or samples[-1:]
    keys = ("gpu_allocated_gb", "gpu_reserved_gb", "host_rss_gb")
    floor = {k: round(min(s[k] for s in h2) - min(s[k] for s in h1), 3) for k in keys}
    band = {k: round(max(s[k] for s in samples) - min(s[k] for s in samples), 3) for k in keys}
    ok = not bad and floor["gpu_allocated_gb"] <= 0.1 and floor["host_rss_gb"] <= 0.1
    return dict(verdict="ok" if ok else "FAILED", floor_second_half_minus_first_half=floor, scatter_over_the_run=band)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--config", default="llama3-8b")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--max-seq", type=int, default=2048)
    ap.add_argument("--emu", action="store_true", help="dry run of this script: tiny dims under the SIMT interpreter on the CPU")
    args = ap.parse_args()
    if args.emu:
        from specforge_amd import build
        _lib._inject_library_for_tests(build.build_emu())
        dev, cfg = torch.device("cpu"), dict(bench.SMALL)
    else:
        _lib.lib()
        dev, cfg = torch.device("cuda", 0), dict(bench.CONFIGS[args.config][0])
    torch.manual_seed(args.seed)
    model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = OnlineEagle3Model(model, length=7).train()
    head = TargetHead((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
    strat = Eagle3TrainStrategy(eagle, target_head=head)
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=100_000))
    backend.prepare_model(eagle)
    eagle.engine.reserve(8, args.max_seq)
    rng = random.Random(args.seed)
    proc = psutil.Process()

    def draw(i):
        B = rng.choice([1, 1, 2, 3, 4, 8])
        S = rng.choice([args.max_seq, rng.randint(16, args.max_seq), rng.randint(16, args.max_seq)])
        if i < 40:      # the first optimizer steps see the LARGEST shape (dense and sparse): whatever is sized by shape has its final size early
            B, S = 8, args.max_seq
        b = bench.make_batch(cfg, B, S, dev, 1000 + i)
        lens = [S] + [rng.randint(max(2, S // 3), S) for _ in range(B - 1)]
        valid = (torch.arange(S)[None, :] < torch.tensor(lens)[:, None]).long()
        lm = valid.clone()
        meta = {"target_repr": "hidden_state"}
        if i % 2:       # chat-style mask: a prompt prefix of every sample carries no loss; the loader knows the per-step row counts
            for r, L in enumerate(lens):
                lm[r, :rng.randint(0, max(1, L // 2))] = 0
            meta["loss_mask_suffix_counts"] = loss_mask_suffix_counts(lm)
        b["attention_mask"], b["loss_mask"] = valid.to(dev), lm.to(dev)
        return TrainBatch(b, meta), B * S

    samples, bad = [], []
    t0, tokens, step = time.time(), 0, 0
    while step < args.steps:
        accum = rng.choice([1, 1, 1, 2, 3])
        for m in range(accum):
            tb, n = draw(10 * step + m)
            out = strat.forward_loss(tb)
            backend.backward(out.loss / accum, is_boundary=(m == accum - 1))
            tokens += n
        gn = backend.step()
        step += 1
        sample = step % (5 if args.emu else 50) == 0 or step == args.steps
        if step % 10 == 0 or sample:
            loss, gnv = float(out.loss.detach()), float(gn) if gn is not None else float("nan")
            if not (math.isfinite(loss) and math.isfinite(gnv)):
                bad.append(dict(step=step, loss=loss, grad_norm=gnv))
        if sample:
            del tb, out          # (the batch itself is up to 0.5 GB: what is measured is what the trainer keeps)
            if dev.type == "cuda":
                torch.cuda.synchronize()
            rec = dict(step=step, loss=loss, grad_norm=gnv,
                       gpu_allocated_gb=round(torch.cuda.memory_allocated() / 1e9, 3), gpu_reserved_gb=round(torch.cuda.memory_reserved() / 1e9, 3),
                       host_rss_gb=round(proc.memory_info().rss / 1e9, 3), elapsed_s=round(time.time() - t0, 1), tokens=tokens)
            samples.append(rec)
            print(json.dumps(rec), flush=True)
    verdict = judge(samples, args.steps, bad)
    print(json.dumps(dict(verdict, steps=args.steps, config=args.config, non_finite=bad, tokens_per_s=round(tokens / (time.time() - t0)))), flush=True)
    sys.exit(0 if verdict["verdict"] == "ok" else 1)


if __name__ == "__main__":
    main()
