"""Same-process, same-box A/B of a boolean engine attribute inside the real optimizer step (alternating legs, medians):

    python tools/engine_flag_ab.py --flag norm_colsum_batched [--config llama3-8b] [--batch 8 --seq 2048] [--steps 10] [--rounds 5]     (GPU box)
    python tools/engine_flag_ab.py --flag teacher_rows --values 16384,4096          (an integer attribute: "on" = the first value)

One model, one batch resident in HBM, the product path (strategy -> autograd node -> backend.step()); the flag is flipped between legs.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flag", required=True)
    ap.add_argument("--config", default="llama3-8b")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq", type=int, default=None)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--values", default=None, help="A,B: two integer values of a non-boolean engine attribute (e.g. --flag teacher_rows --values 16384,4096)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg, B0, S0, _ = bench.CONFIGS[args.config]
    cfg = dict(cfg)
    B, S = args.batch or B0, args.seq or S0
    cfg["max_position_embeddings"] = max(cfg["max_position_embeddings"], S + 16)
    torch.manual_seed(0)
    model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = OnlineEagle3Model(model, length=7).train()
    head = TargetHead((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
    strat = Eagle3TrainStrategy(eagle, target_head=head)
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000_000, warmup_ratio=0.0))
    backend.prepare_model(eagle)
    eng = eagle.engine
    if args.values:
        on, off = (int(x) for x in args.values.split(","))
        assert isinstance(getattr(eng, args.flag), int), f"engine.{args.flag} is not an integer attribute"
    else:
        on, off = True, False
        assert isinstance(getattr(eng, args.flag), bool), f"engine.{args.flag} is not a boolean attribute"
    tb = TrainBatch(bench.make_batch(cfg, B, S, dev, 100), {"target_repr": "hidden_state"})

    def step():
        out = strat.forward_loss(tb)
        backend.backward(out.loss, is_boundary=True)
        backend.step()

    def timed(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / n

    for v in (on, off):
        setattr(eng, args.flag, v)
        timed(3)
    acc = {on: [], off: []}
    for r in range(args.rounds):
        for v in ((on, off) if r % 2 == 0 else (off, on)):      # alternate which leg goes first
            setattr(eng, args.flag, v)
            timed(2)
            acc[v].append(timed(args.steps))
    med = {v: sorted(x)[len(x) // 2] for v, x in acc.items()}
    print(json.dumps(dict(flag=args.flag, values=[on, off], config=args.config, batch=B, seq=S, steps_per_leg=args.steps, rounds=args.rounds, on_ms=med[on], off_ms=med[off],
                          on_minus_off_ms=med[on] - med[off], on_all=[round(x, 2) for x in acc[on]], off_all=[round(x, 2) for x in acc[off]])), flush=True)


if __name__ == "__main__":
    main()
