#!/usr/bin/env python
"""Offline EAGLE3 draft training on MI355X with the specforge_amd path, end to end:

    pre-captured hidden-state files (the reference's format, scripts/prepare_hidden_states.py:446-480)
      -> HiddenStateIngest (normalise, right-pad, DistributedSampler-identical shards, pinned double buffer)
      -> Eagle3TrainStrategy.forward_loss (teacher soft targets, 7-step TTT unroll, CE / LK loss, metrics)
      -> TrainerCore.train_step (accumulation, DP all-reduce on boundaries, clip + AdamW)
      -> checkpoint in the reference's draft state-dict layout (export --to sglang compatible)

One process per GPU:
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_offline_eagle3.py \\
        --draft-config configs/llama3-8B-eagle3.json --features /data/hidden_states --target-head lm_head.pt \\
        --vocab-mapping vocab_mapping.pt --embedding embed_tokens.pt --batch-size 8 --max-len 2048 --epochs 1

`--synthetic N` writes N random samples of the right shapes instead of reading --features (smoke runs, no data needed).
"""
import argparse
import glob
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

import torch
import torch.distributed as dist

from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead
from specforge_amd.ingest import HiddenStateIngest
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend, TrainerCore


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--draft-config", required=True, help="HF-style draft config json (reference configs/*.json)")
    ap.add_argument("--features", help="directory of *.ckpt hidden-state samples")
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--target-head", help="torch file with the target lm_head weight [Vt, Ht]")
    ap.add_argument("--vocab-mapping", help="torch file with t2d (bool [Vt]) and d2t (int64 [Vd])")
    ap.add_argument("--embedding", help="torch file with the target embedding table [Vt, H]")
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--max-len", type=int, default=2048)
    ap.add_argument("--ttt-length", type=int, default=7)
    ap.add_argument("--accumulation-steps", type=int, default=1)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--max-grad-norm", type=float, default=0.5)
    ap.add_argument("--lk-loss-type", choices=["alpha", "lambda"], default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--output", default="eagle3_draft.pt")
    args = ap.parse_args()

    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    cfg = DraftConfig.from_hf(json.load(open(args.draft_config)))
    torch.manual_seed(args.seed)
    model = LlamaForCausalLMEagle3(cfg, device=dev)
    if args.vocab_mapping:
        vm = torch.load(args.vocab_mapping)
        model.load_vocab_mapping_tensors(vm["t2d"], vm["d2t"])
    else:  # synthetic mapping of the fixture kind (tests/test_runtime/_fixtures.py)
        ids = torch.randperm(cfg.vocab_size, generator=torch.Generator().manual_seed(0))[:cfg.draft_vocab_size].sort().values
        t2d = torch.zeros(cfg.vocab_size, dtype=torch.bool)
        t2d[ids] = True
        model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg.draft_vocab_size))
    if args.embedding:
        model.embed_tokens.weight.data.copy_(torch.load(args.embedding).to(torch.bfloat16))
    head_w = torch.load(args.target_head) if args.target_head else torch.randn(cfg.vocab_size, cfg.target_hidden_size) * 0.02
    head = TargetHead(head_w.to(torch.bfloat16).to(dev))

    tmp = None
    if args.synthetic:
        tmp = tempfile.TemporaryDirectory()
        g = torch.Generator().manual_seed(args.seed)
        for i in range(args.synthetic):
            L = args.max_len
            torch.save({"input_ids": torch.randint(0, cfg.vocab_size, (L,), generator=g), "loss_mask": torch.ones(L, dtype=torch.long),
                        "hidden_state": torch.randn(1, L, cfg.target_hidden_size, generator=g).to(torch.bfloat16),
                        "aux_hidden_state": torch.randn(1, L, 3 * cfg.target_hidden_size, generator=g).to(torch.bfloat16)},
                       os.path.join(tmp.name, f"{i:06d}.ckpt"))
        files = sorted(glob.glob(os.path.join(tmp.name, "*.ckpt")))
    else:
        files = sorted(glob.glob(os.path.join(args.features, "**", "*.ckpt"), recursive=True))
    if not files:
        raise SystemExit("no feature files")

    ingest = HiddenStateIngest(files, batch_size=args.batch_size, max_len=args.max_len, target_hidden_size=cfg.target_hidden_size,
                               device=dev, dp_rank=rank, dp_size=world, seed=args.seed)
    steps_total = max(1, ingest.batches_per_epoch() * args.epochs // args.accumulation_steps)
    eagle = OnlineEagle3Model(model, length=args.ttt_length, lk_loss_type=args.lk_loss_type).train()
    strategy = Eagle3TrainStrategy(eagle, target_head=head)
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=args.lr, max_grad_norm=args.max_grad_norm,
                                                                             total_steps=steps_total))
    backend.prepare_model(eagle)
    core = TrainerCore(strategy, backend, accumulation_steps=args.accumulation_steps)

    t0, tokens = time.perf_counter(), 0
    for epoch in range(args.epochs):
        for batch in ingest.epoch(epoch):
            res = core.train_step(batch)
            tokens += batch.tensors["input_ids"].numel() * world
            if res.stepped and rank == 0 and core.global_step % 10 == 0:
                torch.cuda.synchronize()
                acc = float(torch.stack(res.metrics["acces"]).mean())
                print(f"step {core.global_step}/{steps_total} loss {float(res.loss):.4f} acc {acc:.3f} "
                      f"grad_norm {float(res.grad_norm):.3f} lr {backend.optimizer.get_learning_rate():.2e} "
                      f"{tokens / (time.perf_counter() - t0):.0f} tok/s", flush=True)
    if rank == 0:
        state = backend.state_dict()
        torch.save(strategy.checkpoint_state_filter(state["model"]), args.output)   # reference draft checkpoint keys
        print("saved", args.output, "after", core.global_step, "optimizer steps")
    if tmp is not None:
        tmp.cleanup()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
