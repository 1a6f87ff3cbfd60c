"""What ANY fusion of RoPE / the forward RMSNorms could return at most (measurements, not arguments).

The headline step (Llama-3-8B draft dims, bs 8 x 2048, ttt 7, optimizer included) runs in ONE process with the product library;
legs alternate (A B A B ...) so that box drift cancels.  An ablated leg SKIPS the launches of one kernel family outright -- its
outputs keep the previous (valid) step's values, so every other kernel still sees realistic data and the chip's power draw is
unchanged; the result is numerically meaningless and only timed.  "skip" is the upper bound of a fusion's return: a fused form
still has to do the arithmetic somewhere and, for the norms, still has to WRITE the normalised activations -- they are the X
operand of the deferred weight-gradient GEMMs (engine.py: hn_s / pn_s / ln_s stashes), so at most the read half disappears.

    python tools/ablate_pointwise.py --steps 8 --rounds 3 > gpurun_out/ablate_pointwise.json
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from specforge_amd import _lib, ops  # noqa: E402
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=8)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()
_lib.lib()
dev = torch.device("cuda", 0)
cfg, B, S, T = bench.LLAMA3_8B, 8, 2048, 7
torch.manual_seed(0)
model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
t2d[ids] = True
model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
eagle = OnlineEagle3Model(model, length=T).train()
head = TargetHead((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
strat = Eagle3TrainStrategy(eagle, target_head=head)
backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000))
backend.prepare_model(eagle)
batches = [TrainBatch(bench.make_batch(cfg, B, S, dev, 100 + i), {"target_repr": "hidden_state"}) for i in range(2)]


def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        out = strat.forward_loss(batches[i % 2])
        backend.backward(out.loss, is_boundary=True)
        backend.step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


class Skip:
    def __init__(self, names):
        self.names, self.saved, self.count = names, {}, 0

    def __enter__(self):
        for n in self.names:
            self.saved[n] = getattr(ops, n)

            def noop(*a, _s=self, **k):
                _s.count += 1

            setattr(ops, n, noop)
        return self

    def __exit__(self, *a):
        for n, f in self.saved.items():
            setattr(ops, n, f)


LEGS = {
    "rope (forward + backward launches)": ["rope_"],
    "rmsnorm forward (rmsnorm_fwd + rmsnorm_fwd2)": ["rmsnorm_fwd", "rmsnorm_fwd2"],
    "rmsnorm backward (rmsnorm_bwd + rmsnorm_bwd2)": ["rmsnorm_bwd", "rmsnorm_bwd2"],
}
run(3)
res = {"steps_per_leg": args.steps, "rounds": args.rounds, "legs": {}}
for what, names in LEGS.items():
    base, abl, launches = [], [], 0
    for r in range(args.rounds):
        order = ("base", "abl") if r % 2 == 0 else ("abl", "base")
        for leg in order:
            if leg == "base":
                base.append(run(args.steps))
            else:
                with Skip(names) as sk:
                    abl.append(run(args.steps))
                launches = sk.count // args.steps
            run(1)          # a valid step in between: the stale buffers of the next ablated leg hold real values again
    res["legs"][what] = {"base_ms": base, "skipped_ms": abl, "launches_per_step": launches,
                         "upper_bound_ms_per_step": sum(base) / len(base) - sum(abl) / len(abl)}
print(json.dumps(res, indent=1))
