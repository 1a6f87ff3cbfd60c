"""The training step on RAGGED batches (what real data looks like: the collator pads a batch to its own longest sample, so B*S is
arbitrary and changes every step) next to the aligned headline shape, one process.  Each ragged batch has per-sample lengths below
its padded S (right-padded attention / loss masks).  tokens/s counts B * S of each batch, as the headline metric does.   (GPU box)"""
import json
import random
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from specforge_amd import _lib  # noqa: E402
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402

_lib.lib()
dev = torch.device("cuda", 0)
cfg, B, T = bench.LLAMA3_8B, (1 if "--bs1" in sys.argv else 8), 7      # --bs1: the reference recipes' batch size, a new length every step
torch.manual_seed(0)
model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
t2d[ids] = True
model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
eagle = OnlineEagle3Model(model, length=T).train()
head = TargetHead((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
strat = Eagle3TrainStrategy(eagle, target_head=head)
backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000))
backend.prepare_model(eagle)
eagle.engine.reserve(B, 2048)


def batch(S, seed, ragged):
    b = bench.make_batch(cfg, B, S, dev, seed)
    if ragged:
        rnd = random.Random(seed)
        lens = [S] + [rnd.randint(int(0.6 * S), S) for _ in range(B - 1)]      # the longest sample defines S
        m = (torch.arange(S, device=dev)[None, :] < torch.tensor(lens, device=dev)[:, None]).long()
        b["attention_mask"], b["loss_mask"] = m, m.clone()
        if "--no-compact" not in sys.argv:     # the per-step loss-row counts a loader computes from its host copy of the mask (loss-row compaction)
            from specforge_amd.eagle3 import loss_mask_suffix_counts
            return TrainBatch(b, {"target_repr": "hidden_state", "loss_mask_suffix_counts": loss_mask_suffix_counts(m.cpu())})
    return TrainBatch(b, {"target_repr": "hidden_state"})


def run(seqs, ragged, steps):
    bs = [batch(S, 100 + i, ragged) for i, S in enumerate(seqs)]
    for i in range(2):           # warm-up (also sizes / visits the shapes once)
        out = strat.forward_loss(bs[i % len(bs)]); backend.backward(out.loss, is_boundary=True); backend.step()
    torch.cuda.synchronize()
    t0, tok = time.perf_counter(), 0
    for i in range(steps):
        tb = bs[i % len(bs)]
        out = strat.forward_loss(tb); backend.backward(out.loss, is_boundary=True); backend.step()
        tok += B * tb.tensors["input_ids"].shape[1] if hasattr(tb, "tensors") else B * seqs[i % len(seqs)]
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    return dict(seqs=seqs, ragged=ragged, steps=steps, ms_per_step=1e3 * el / steps, tokens_per_s=tok / el, final_loss=float(out.loss))


if "--bs1" in sys.argv:
    # one sample per step: the same 8 lengths visited in turn (a new shape every step) vs each length repeated (no shape switch)
    L = [1000, 1432, 771, 1999, 1203, 888, 1640, 1111]
    res = [run(L, False, 32)] + [run([s], False, 4) for s in L]
    fixed = sum(r["ms_per_step"] for r in res[1:]) / len(L)
    res.append(dict(summary="bs 1, lengths " + str(L), switching_ms_per_step=res[0]["ms_per_step"], same_lengths_without_switching_ms_per_step=fixed))
elif "--ragged-only" in sys.argv:      # (for a kernel trace of the ragged steps alone)
    res = [run([2048, 1999, 1873, 2011, 1777, 1931, 2047, 1685], True, 8)]
else:
    res = [run([2048], False, 8),
           run([2048, 1999, 1873, 2011, 1777, 1931, 2047, 1685], True, 16),     # a new (B, S) every step, nothing a multiple of 64
           run([2048], False, 8)]
for r in res:
    print(json.dumps(r))
