#!/usr/bin/env python
"""Headline benchmark: EAGLE3 draft-training tokens/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A "step" = one full optimizer step of the hot path on one synthetic batch that is already
resident in HBM: teacher soft targets from the target hidden state, 7 TTT unroll steps of the
draft layer forward + fused CE, the backward sweep with deferred wgrad GEMMs, the RCCL gradient
all-reduce (N>1), grad-norm + clip + AdamW.  Workload = BASELINE.json configs[1]: Llama-3-8B
EAGLE3 draft (H 4096, I 14336, 32/8 heads, Vd 32000, Vt 128256), bf16, per-GPU batch 8 x 2048.
Weak scaling: per-GPU work is fixed, value = all ranks' tokens / max-over-ranks time.

Rank 0 prints ONE JSON line.  ``roofline`` is for the dominant kernel (the bf16 MFMA GEMM):
achieved = sum over its launches in the timed region of 2*M*N*K / sum of their HIP-event
durations on the launch stream; peak = 2500 TFLOP/s dense bf16 (MI355X_MICROARCH.md).
``cpu_baseline`` = the CPU oracle (oracle/eagle3_oracle.py, a restatement of the reference pinned
to reference-generated goldens) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LLAMA3_8B = dict(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                 vocab_size=128256, draft_vocab_size=32000, head_dim=128, target_hidden_size=4096,
                 max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=500000.0,
                 rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                   original_max_position_embeddings=8192))
SMALL = dict(hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=2, vocab_size=4096,
             draft_vocab_size=1024, head_dim=128, target_hidden_size=512, max_position_embeddings=2048, rms_norm_eps=1e-5)
PEAK_BF16_TFLOPS = 2500.0


class GemmTimer:
    """HIP events around every sf_gemm_nt launch on the current stream (the launch stream)."""

    def __init__(self):
        self.records = []

    def wrap(self, ops):
        orig = ops.gemm_nt

        def timed(a, b, out, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(a, b, out, **kw)
            e.record()
            self.records.append((2.0 * a.shape[0] * b.shape[0] * a.shape[1], s, e))
            return r

        ops.gemm_nt = timed
        return orig

    def summary(self):
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return fl, ms, len(self.records)


def make_batch(cfg, B, S, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    Ht = cfg["target_hidden_size"]
    return dict(
        input_ids=torch.randint(0, cfg["vocab_size"], (B, S), device=dev, generator=g),
        attention_mask=torch.ones(B, S, dtype=torch.int64, device=dev),
        loss_mask=torch.ones(B, S, dtype=torch.int64, device=dev),
        hidden_state=torch.randn(B, S, 3 * Ht, device=dev, generator=g).to(torch.bfloat16),
        target=torch.randn(B, S, Ht, device=dev, generator=g).to(torch.bfloat16),
    )


def cpu_baseline(cfg, S, ttt, threads):
    """Oracle forward+backward on host cores, one B=1 x S sample of the same model dimensions."""
    from oracle import eagle3_oracle as O

    torch.set_num_threads(threads)
    oc = O.DraftConfig(**{k: cfg[k] for k in ("hidden_size", "intermediate_size", "num_attention_heads", "num_key_value_heads",
                                               "vocab_size", "draft_vocab_size", "head_dim", "target_hidden_size",
                                               "max_position_embeddings", "rms_norm_eps")},
                       rope_theta=cfg.get("rope_theta", 10000.0), rope_scaling=cfg.get("rope_scaling"))
    p = {k: v.requires_grad_(True) for k, v in O.init_params(oc, seed=0).items()}
    g = torch.Generator().manual_seed(0)
    embed = torch.randn(oc.vocab_size, oc.hidden_size, generator=g) * 0.02
    head = torch.randn(oc.vocab_size, oc.target_hidden_size, generator=g) * 0.02
    t2d, d2t = O.make_vocab_mapping(oc.vocab_size, oc.draft_vocab_size, seed=0)
    b = O.make_batch(oc, 1, S, seed=1, dtype=torch.float32)
    t0 = time.time()
    out = O.eagle3_forward(p, oc, embed_weight=embed, target_head_weight=head, t2d=t2d, d2t=d2t, input_ids=b["input_ids"],
                           attention_mask=b["attention_mask"], loss_mask=b["loss_mask"], hidden_state=b["hidden_state"],
                           target_hidden=b["target"], ttt_length=ttt)
    out.loss.backward()
    dt = time.time() - t0
    return dict(value=S / dt, unit="tokens/s", cores=threads, kind="port",
                sample=f"1 micro-step fwd+bwd, B=1 x S={S} tokens of the same model dims, fp32 oracle, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--ttt", type=int, default=7)
    ap.add_argument("--small", action="store_true", help="tiny model dims (smoke / debugging only; NOT the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seq", type=int, default=64)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with python -m torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from specforge_amd import _lib, ops
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend

    _lib.lib()  # fails loudly if libsfhip.so is missing
    cfg = SMALL if args.small else LLAMA3_8B
    B, S = args.batch, args.seq
    torch.manual_seed(0)
    model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = OnlineEagle3Model(model, length=args.ttt).train()
    head = TargetHead((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
    strat = Eagle3TrainStrategy(eagle, target_head=head)
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000))
    backend.prepare_model(eagle)
    batches = [TrainBatch(make_batch(cfg, B, S, dev, 100 + rank * 10 + i), {"target_repr": "hidden_state"}) for i in range(2)]

    def step(i):
        out = strat.forward_loss(batches[i % 2])
        backend.backward(out.loss, is_boundary=True)
        backend.step()
        return out

    for i in range(args.warmup):
        step(i)
    timer = GemmTimer()
    orig = timer.wrap(ops)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.gemm_nt = orig
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out.loss.detach())
    fl, gemm_ms, nlaunch = timer.summary()
    tokens = world * B * S * args.steps
    if rank == 0:
        ach = fl / gemm_ms / 1e9 if gemm_ms > 0 else 0.0
        # HBM-side traffic of the GEMM kernel per launch: from the committed rocprofv3 --pmc passes of this same
        # command (profiles/r1_pmc_fetch_write_summary.json; FETCH_SIZE doubled per the gfx950 correction of
        # MI355X_MICROARCH.md, calibrated there on the AdamW kernel's known byte count).  PMC counters cannot be
        # collected from inside the timed run, so this field is null when the summary is absent.
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_fetch_write_summary.json")))["gemm256w4"]
            traffic = (2.0 * pm["FETCH_SIZE_sum"] + pm["WRITE_SIZE_sum"]) * 1024.0 / pm["launches"]
        except Exception:
            pass
        line = {
            "metric": "EAGLE3 draft train tokens/sec, Llama-3-8B target, seq2048 at 1/2/4/8 MI355X",
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("SMALL-debug" if args.small else "Llama-3-8B EAGLE3 offline draft")
                       + f", bf16, per-GPU batch {B} x seq {S}, ttt {args.ttt}, optimizer step included",
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_256w4_kernel (bf16 MFMA GEMM, 256x256x64, 4 waves x 128x128, software-pipelined)", "achieved": ach,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "launches_per_step": nlaunch / max(1, args.steps),
                         "gemm_ms_per_step": gemm_ms / max(1, args.steps)},
            "final_loss": loss,
            "hbm_peak_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                # bounded sample: 32 threads (more only adds oversubscription on these matrix sizes)
                line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_sample_seq, args.ttt, min(32, os.cpu_count() or 1))
            except Exception as e:  # the baseline is reported, never required for the GPU number
                line["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "port",
                                        "sample": f"failed: {e}"}
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
