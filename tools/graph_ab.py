"""What a captured hipGraph of the whole optimizer step would return (VERDICT r4 next #4a): eager launches vs graph replay of the SAME
kernel sequence, one process, fixed batch shapes, HBM-resident synthetic batches.

    python tools/graph_ab.py [--config llama3-8b] [--shapes 1x512,1x1024,1x2048,1x4096,8x2048] [--steps 20]      (GPU box)

Legs per shape (ms per optimizer step, alternating rounds):
  strategy   the product path: Eagle3TrainStrategy.forward_loss -> loss.backward() (autograd node, upstream gradient read back through
             pinned memory at the end of the sweep) -> backend.step()
  raw        the same kernels without autograd: engine.forward + engine.backward(1.0) + optimizer.step() (what a graph would capture)
  graph      torch.cuda.CUDAGraph replay of `raw` (lr / bias-correction step are launch constants of the capture: a product version
             would read them from device memory; the measurement does not depend on their values)
The mask is all ones (dense lm_head path: loss-row compaction takes per-batch row counts, i.e. a different launch sequence per batch).
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
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch, loss_mask_suffix_counts  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="llama3-8b")
    ap.add_argument("--shapes", default="1x512,1x1024,1x2048,1x4096,8x2048")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CONFIGS[args.config][0])
    shapes = [tuple(int(x) for x in s.split("x")) for s in args.shapes.split(",")]
    cfg["max_position_embeddings"] = max(cfg["max_position_embeddings"], max(S for _, S in shapes) + 16)
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
    eng, opt = eagle.engine, backend.optimizer
    Bm, Sm = max(b for b, _ in shapes), max(s for _, s in shapes)
    eng.reserve(Bm, Sm)

    for B, S in shapes:
        lm = torch.ones(B, S, dtype=torch.int64)
        tb = TrainBatch(bench.make_batch(cfg, B, S, dev, 100, lm), {"target_repr": "hidden_state", "loss_mask_suffix_counts": loss_mask_suffix_counts(lm)})
        t = tb.tensors
        ids_s, tgt_s, lm_s = TargetHead.preprocess(t["input_ids"], t["target"], t["loss_mask"])
        kw = dict(input_ids=ids_s, attention_mask=t["attention_mask"], loss_mask=lm_s, hidden_states=t["hidden_state"], target_hidden=tgt_s,
                  target_head_weight=head.fc.weight.data, train=True)

        def strategy_step():
            out = strat.forward_loss(tb)
            backend.backward(out.loss, is_boundary=True)
            backend.step()

        def raw_step():
            eng.forward(**kw)
            eng.backward(1.0)
            opt.step()

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return 1e3 * (time.perf_counter() - t0) / n

        for _ in range(3):
            strategy_step()
        raw_step()
        res = {"config": args.config, "batch": B, "seq": S}
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    raw_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                raw_step()
            graph_ok = True
        except Exception as e:      # a capture-breaking call in the step: report it, keep the eager legs
            graph_ok = False
            res["graph_error"] = f"{type(e).__name__}: {e}"[:400]
            torch.cuda.synchronize()
        legs = {"strategy": strategy_step, "raw": raw_step}
        if graph_ok:
            legs["graph"] = graph.replay
        acc = {k: [] for k in legs}
        for _ in range(args.rounds):
            for k, fn in legs.items():
                timed(fn, 2)
                acc[k].append(timed(fn, args.steps))
        for k, v in acc.items():
            res[k + "_ms"] = sorted(v)[len(v) // 2]
        if graph_ok:
            res["graph_vs_strategy"] = res["graph_ms"] / res["strategy_ms"]
            res["final_loss_finite"] = bool(torch.isfinite(eng.flat.data.float()).all())
            del graph
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
