"""Ingest rate of specforge_amd.ingest.HiddenStateIngest at the headline shape (Llama-3-8B target: Ht 4096, S 2048, B 8:
0.54 GB of bf16 hidden states per micro-step), alone and underneath a running training step.

    python tools/ingest_bench.py [--files 32] > gpurun_out/ingest_bench.json       (GPU box; writes the files to /tmp)

Reports: files -> device GB/s with nothing else running (loader thread: torch.load(mmap) from the page cache + normalise +
right-pad into pinned memory, HIP copy stream to HBM); step time of the full training step fed by the ingest vs fed by
HBM-resident batches (the bench's regime) = what the input stream costs when it is overlapped."""
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, ".")
from bench import LLAMA3_8B  # noqa: E402
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch  # noqa: E402
from specforge_amd.ingest import HiddenStateIngest  # noqa: E402
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3  # noqa: E402
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend  # noqa: E402

nfiles = int(sys.argv[sys.argv.index("--files") + 1]) if "--files" in sys.argv else 32
cfg, B, S = LLAMA3_8B, 8, 2048
Ht, Vt = cfg["target_hidden_size"], cfg["vocab_size"]
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(prefix="ingest_", dir="/tmp")
g = torch.Generator().manual_seed(0)
files = []
t0 = time.time()
for i in range(nfiles):
    p = os.path.join(d, f"{i:05d}.ckpt")
    torch.save({"input_ids": torch.randint(0, Vt, (S,), generator=g), "loss_mask": torch.ones(S, dtype=torch.long),
                "hidden_state": torch.randn(1, S, Ht, generator=g).to(torch.bfloat16),
                "aux_hidden_state": torch.randn(1, S, 3 * Ht, generator=g).to(torch.bfloat16)}, p)
    files.append(p)
bytes_per_batch = B * S * 4 * Ht * 2 + 3 * B * S * 8
res = dict(files=nfiles, write_s=round(time.time() - t0, 1), bytes_per_batch=bytes_per_batch)

ing = HiddenStateIngest(files, batch_size=B, max_len=S, target_hidden_size=Ht, device=dev, shuffle=False)
for rep in range(2):      # rep 0 warms the page cache and the pinned buffers
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for batch in ing.epoch(0):
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
res["ingest_alone"] = dict(batches=n, seconds=round(dt, 3), GBps=round(n * bytes_per_batch / dt / 1e9, 2), ms_per_batch=round(1e3 * dt / n, 1))

torch.manual_seed(0)
model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
ids = torch.randperm(Vt, generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
t2d = torch.zeros(Vt, dtype=torch.bool)
t2d[ids] = True
model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
eagle = OnlineEagle3Model(model, length=7).train()
strat = Eagle3TrainStrategy(eagle, target_head=TargetHead((torch.randn(Vt, Ht, device=dev) * 0.02).to(torch.bfloat16)))
be = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, total_steps=10_000))
be.prepare_model(eagle)


def step(batch):
    out = strat.forward_loss(batch)
    be.backward(out.loss, is_boundary=True)
    be.step()


resident = None
for batch in ing.epoch(0):     # warm-up + a resident copy of one batch
    if resident is None:
        resident = TrainBatch({k: v.clone() for k, v in batch.tensors.items()}, dict(batch.metadata))
    step(batch)
    break
for _ in range(2):
    step(resident)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(nfiles // B):
    step(resident)
torch.cuda.synchronize()
res["step_ms_resident_batches"] = round(1e3 * (time.perf_counter() - t0) / (nfiles // B), 2)
n = 0
for batch in ing.epoch(1):
    if n == 1:                     # steady state: the first batch of an epoch waits for the loader's first fill
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    step(batch)
    n += 1
torch.cuda.synchronize()
res["step_ms_fed_by_ingest"] = round(1e3 * (time.perf_counter() - t0) / (n - 1), 2)
res["steps"] = n - 1
print(json.dumps(res))
for p in files:
    os.remove(p)
os.rmdir(d)
