"""Ingest headroom for an 8-rank node measured on ONE host: P concurrent loader processes (one per would-be rank), each
running specforge_amd.ingest.HiddenStateIngest over its own shard of the same files (page cache warm) into its own
pinned double buffer and on to HBM of cuda:0.  SURVEY 8e: 8 ranks x 0.54 GB per 214 ms step = 20 GB/s aggregate.

    python tools/ingest_procs.py --procs 8 [--files 64] > gpurun_out/ingest_procs.json        (GPU box)

Reports per-process and aggregate files -> device GB/s; `host_only` repeats the run without the device copy (files ->
pinned host), which is the part that scales with host cores rather than with the single PCIe link of this box."""
import json
import os
import subprocess
import sys
import tempfile
import time

import torch

sys.path.insert(0, ".")


def arg(name, dflt):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else dflt


Ht, S, B, Vt = arg("--hidden", 4096), arg("--seq", 2048), 8, 128256      # (--hidden / --seq: the small smoke form of tests/test_bench_launch.py)
bytes_per_batch = B * S * 4 * Ht * 2 + 3 * B * S * 8

if "--worker" in sys.argv:
    from specforge_amd.ingest import HiddenStateIngest

    d, rank, procs, host_only = sys.argv[sys.argv.index("--worker") + 1], arg("--rank", 0), arg("--procs", 1), "--host-only" in sys.argv
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".ckpt"))
    dev = torch.device("cpu") if host_only else torch.device("cuda", 0)
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // procs))
    ing = HiddenStateIngest(files, batch_size=B, max_len=S, target_hidden_size=Ht, device=dev, dp_rank=rank, dp_size=procs,
                            shuffle=False)
    if host_only:     # the loader's own work only: fill the staging slot, no clone / device copy
        from specforge_amd.training import distributed_sampler_indices
        idx = distributed_sampler_indices(len(files), dp_rank=rank, dp_size=procs, seed=0, epoch=0, shuffle=False)
        groups = [idx[i:i + B] for i in range(0, len(idx) - B + 1, B)]
        for g in groups[:1]:
            ing._fill(ing._slots[0], [files[i] for i in g])
    else:
        for _ in ing.epoch(0):
            pass
        torch.cuda.synchronize()
    # rendezvous: start together
    open(os.path.join(d, f"ready.{rank}"), "w").close()
    while len([f for f in os.listdir(d) if f.startswith("ready.")]) < procs:
        time.sleep(0.01)
    t0 = time.perf_counter()
    n = 0
    for rep in range(2):
        if host_only:
            for i, g in enumerate(groups):
                ing._fill(ing._slots[i & 1], [files[j] for j in g])
                n += 1
        else:
            for _ in ing.epoch(rep):
                n += 1
            torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps(dict(rank=rank, batches=n, seconds=round(dt, 3), GBps=round(n * bytes_per_batch / dt / 1e9, 2))))
    sys.exit(0)

procs, nfiles = arg("--procs", 8), arg("--files", 64)
d = tempfile.mkdtemp(prefix="ingestp_", dir="/tmp")
g = torch.Generator().manual_seed(0)
for i in range(nfiles):
    torch.save({"input_ids": torch.randint(0, Vt, (S,), generator=g), "loss_mask": torch.ones(S, dtype=torch.long),
                "hidden_state": torch.randn(1, S, Ht, generator=g).to(torch.bfloat16),
                "aux_hidden_state": torch.randn(1, S, 3 * Ht, generator=g).to(torch.bfloat16)}, os.path.join(d, f"{i:05d}.ckpt"))
res = dict(procs=procs, files=nfiles, host_cores=os.cpu_count(), bytes_per_batch=bytes_per_batch)
for mode in (("device", "host_only") if torch.cuda.is_available() else ("host_only",)):
    for f in os.listdir(d):
        if f.startswith("ready."):
            os.remove(os.path.join(d, f))
    ps = [subprocess.Popen([sys.executable, __file__, "--worker", d, "--rank", str(r), "--procs", str(procs), "--hidden", str(Ht), "--seq", str(S)] +
                           (["--host-only"] if mode == "host_only" else []), stdout=subprocess.PIPE, text=True) for r in range(procs)]
    outs = [json.loads(p.communicate(timeout=600)[0].strip().splitlines()[-1]) for p in ps]
    res[mode] = dict(per_proc_GBps=[o["GBps"] for o in outs], aggregate_GBps=round(sum(o["GBps"] for o in outs), 2),
                     needed_GBps=round(procs * bytes_per_batch / 0.214 / 1e9, 2))
print(json.dumps(res))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
