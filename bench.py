#!/usr/bin/env python
"""Headline benchmark: EAGLE3 draft-training tokens/s on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N>1 works both ways: under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (RANK /
WORLD_SIZE in the environment), and as a bare ``python bench.py --gpus N`` -- then this process re-executes itself
under torch.distributed.run on 127.0.0.1 with a free port, one rank per GPU over RCCL.

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
GEMM_KERNEL_NAME = "gemm_nt_256w4_kernel (bf16 MFMA GEMM, 256x256x64, 4 waves x 128x128, plan-scheduled: one filler per MFMA slot, counted vmcnt)"


class GemmTimer:
    """HIP events around every sf_gemm_nt launch on the current stream (the launch stream)."""

    def __init__(self):
        self.records = []   # sf_gemm_nt: the plain instantiations of the dominant kernel (roofline object)
        self.fused = []     # sf_gemm_nt_swiglu_bwd: the same main loop with d(SwiGLU) in its epilogue (its own kernel symbol)

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
        orig_sw = ops.gemm_nt_swiglu_bwd

        def timed_sw(a, b, gu, dgu, dact):   # the down-projection dgrad with d(SwiGLU) in its epilogue: same kernel, same flops
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig_sw(a, b, gu, dgu, dact)
            e.record()
            self.fused.append((2.0 * a.shape[0] * b.shape[0] * a.shape[1], s, e))
            return r

        ops.gemm_nt_swiglu_bwd = timed_sw
        self._orig_sw = orig_sw
        return orig

    def summary(self):
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return fl, ms, len(self.records)

    def fused_summary(self):
        return (sum(r[0] for r in self.fused), sum(r[1].elapsed_time(r[2]) for r in self.fused), len(self.fused))


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
                sample=f"oracle/eagle3_oracle.py (a pinned restatement, NOT the reference trainer: no optimizer, no loader), "
                       f"1 micro-step fwd+bwd, B=1 x S={S} of the same model dims and ttt, fp32, {dt:.1f} s; the sdpa path's "
                       f"score tensors grow with S^2, so the rate at S=2048 is lower still")


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
    ap.add_argument("--cpu-sample-seq", type=int, default=512)
    ap.add_argument("--no-dense-mask", action="store_true", help="skip the dense-position-mask variant")
    ap.add_argument("--dp-single", action="store_true", help="one gradient all-reduce after the sweep instead of overlapped buckets (A/B)")
    ap.add_argument("--force-dp", action="store_true", help="run the gradient collectives even at world size 1 (RCCL path on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the product path) | gloo (launcher smoke test)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: all ranks on cuda:0 (with --dist-backend gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher (one rank per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ   # torch.distributed.run
    if world > 1 or launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.dist_backend)

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
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000),
                                   single_collective=args.dp_single, force_collectives=args.force_dp)
    backend.prepare_model(eagle)
    batches = [TrainBatch(make_batch(cfg, B, S, dev, 100 + rank * 10 + i), {"target_repr": "hidden_state"}) for i in range(2)]

    def timed(strategy, nsteps, timer=None):
        """W warm-up steps were done by the caller; times exactly nsteps optimizer steps: barrier + synchronize on
        both sides, MAX over ranks."""
        def step(i):
            out = strategy.forward_loss(batches[i % 2])
            backend.backward(out.loss, is_boundary=True)
            backend.step()
            return out

        orig = timer.wrap(ops) if timer is not None else None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = None
        for i in range(nsteps):
            out = step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if orig is not None:
            ops.gemm_nt = orig
            ops.gemm_nt_swiglu_bwd = timer._orig_sw
        if world > 1:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    timed(strat, args.warmup)                      # untimed warm-up
    timer = GemmTimer()
    elapsed, out = timed(strat, args.steps, timer)
    loss = float(out.loss.detach())
    fl, gemm_ms, nlaunch = timer.summary()
    tokens = world * B * S * args.steps

    # ---- dense position mask variant: the specified synthetic vocab map (random 32000 of 128256)
    # leaves ~25 % of the rows with a position mask, and the fused CE skips the soft-target read of masked rows.  Here the
    # teacher's argmax always lands inside the draft vocabulary (head rows outside it are zero), so every row reads its
    # target -- the worst case of real data.  Reported beside the headline value, never instead of it.
    dense = None
    if not args.no_dense_mask:
        hw = head.fc.weight.data.clone()
        hw[~t2d.to(dev)] = 0
        strat_dense = Eagle3TrainStrategy(eagle, target_head=TargetHead(hw))
        timed(strat_dense, 1)
        e2, out2 = timed(strat_dense, args.steps)
        pm = eagle.last_artifacts["position_mask"].float().mean()
        dense = {"value": tokens / e2, "ms_per_step": 1e3 * e2 / args.steps, "position_mask_density": float(pm)}

    # ---- RCCL evidence: the gradient all-reduce of each bucket, timed alone on the communicator (outside the timed region)
    rccl = None
    if dist.is_initialized():
        f = eagle.engine.flat
        bounds = eagle.engine.bucket_bounds() if hasattr(eagle.engine, "bucket_bounds") else [(0, f.numel)]
        per = []
        for lo, hi in bounds:
            buf = torch.zeros(hi - lo, dtype=torch.bfloat16, device=dev)
            dist.all_reduce(buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            per.append(dict(mbytes=(hi - lo) * 2 / 1e6, ms=(time.perf_counter() - t0) / 3 * 1e3))
        rccl = {"backend": args.dist_backend, "rccl_ranks": world, "buckets": per,
                "allreduce_ms_total_unoverlapped": sum(x["ms"] for x in per),
                "no_sync_backwards": backend.no_sync_backwards, "single_collective": backend._single_collective}

    if rank == 0:
        ach = fl / gemm_ms / 1e9 if gemm_ms > 0 else 0.0
        # HBM-side traffic of the GEMM kernel per launch: from the committed rocprofv3 --pmc passes of this same
        # command (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, calibrated there on the AdamW
        # kernel's known byte count).  PMC counters cannot be collected from inside the timed run, so this field is the
        # committed per-launch figure (newest profiles/r*_pmc_fetch_write_summary.json) or null.
        traffic, traffic_src = None, None
        try:
            import glob

            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_summary.json")))[-1]
            pm = json.load(open(src))["gemm256w4"]
            traffic = (2.0 * pm["FETCH_SIZE_sum"] + pm["WRITE_SIZE_sum"]) * 1024.0 / pm["launches"]
            traffic_src = os.path.relpath(src, ROOT)
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
            "roofline": {"bound": "mfma", "kernel": GEMM_KERNEL_NAME, "achieved": ach,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src, "launches_per_step": nlaunch / max(1, args.steps),
                         "fused_swiglu_dgrad": {"launches_per_step": timer.fused_summary()[2] / max(1, args.steps),
                                                "ms_per_step": timer.fused_summary()[1] / max(1, args.steps),
                                                "gemm_tflops": (timer.fused_summary()[0] / max(timer.fused_summary()[1], 1e-9)) / 1e9},
                         "gemm_ms_per_step": gemm_ms / max(1, args.steps)},
            "final_loss": loss,
            "hbm_peak_gb": torch.cuda.max_memory_allocated(dev) / 1e9,
        }
        if dense is not None:
            line["dense_mask"] = dense
        if rccl is not None:
            line["rccl"] = rccl
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
