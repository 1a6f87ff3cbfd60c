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

# dmabuf IPC is the only kind this stack's host driver supports: without it RCCL's buffer exchange between the ranks of a node fails
# (hipIpcGetMemHandle: invalid argument).  Exported on the GPU boxes already; set here, before the runtime initialises, for any other launcher.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LLAMA3_8B = dict(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                 vocab_size=128256, draft_vocab_size=32000, head_dim=128, target_hidden_size=4096,
                 max_position_embeddings=2048, rms_norm_eps=1e-5, rope_theta=500000.0,
                 rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                   original_max_position_embeddings=8192))
# The other BASELINE.json configurations (dims from the reference's configs/*.json: qwen3-8b-eagle3.json,
# qwen3-30B-A3B-eagle3.1.json, deepseek-v3-671b-eagle3.json).  Each entry: model dims + the (batch, seq)
# its line is quoted on.  `--config X` prints the same line (roofline, kernels, ...) for X; the driver's headline stays llama3-8b.
QWEN3_8B = dict(hidden_size=4096, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=8, vocab_size=151936,
                draft_vocab_size=32000, head_dim=128, target_hidden_size=4096, max_position_embeddings=40960, rms_norm_eps=1e-6,
                rope_theta=1000000.0)
QWEN3_30B_A3B_EAGLE31 = dict(hidden_size=2048, intermediate_size=12288, num_attention_heads=32, num_key_value_heads=4,
                             vocab_size=151936, draft_vocab_size=32000, head_dim=128, target_hidden_size=2048,
                             max_position_embeddings=4096, rms_norm_eps=1e-6, rope_theta=1000000.0, fc_norm=True)
DEEPSEEK_V3 = dict(hidden_size=7168, intermediate_size=40960, num_attention_heads=56, num_key_value_heads=8, vocab_size=129280,
                   draft_vocab_size=32000, head_dim=128, target_hidden_size=7168, max_position_embeddings=163840, rms_norm_eps=1e-5)
# head_dim 256 (configs/qwen3-next-80b-a3b-eagle3.json; gemma3-1b and qwen3.5-35b-a3b share the head shape)
QWEN3_NEXT_80B_A3B = dict(hidden_size=2048, intermediate_size=16384, num_attention_heads=16, num_key_value_heads=2, vocab_size=151936,
                          draft_vocab_size=32000, head_dim=256, target_hidden_size=2048, max_position_embeddings=8192,
                          rms_norm_eps=1e-6, rope_theta=10000000.0)
GEMMA3_1B = dict(hidden_size=1152, intermediate_size=6912, num_attention_heads=4, num_key_value_heads=1, vocab_size=262144,
                 draft_vocab_size=32000, head_dim=256, target_hidden_size=1152, max_position_embeddings=32768, rms_norm_eps=1e-6,
                 rope_theta=1000000.0)
CONFIGS = {
    # name: (dims, default batch, default seq, label)
    "llama3-8b": (LLAMA3_8B, 8, 2048, "Llama-3-8B EAGLE3 offline draft"),
    "qwen3-8b": (QWEN3_8B, 8, 2048, "Qwen3-8B EAGLE3 offline draft"),
    "qwen3-30b-a3b-eagle31": (QWEN3_30B_A3B_EAGLE31, 1, 4096, "Qwen3-30B-A3B EAGLE3.1 offline draft (fc_norm)"),
    "deepseek-v3": (DEEPSEEK_V3, 1, 2048, "DeepSeek-V3 671B EAGLE3 offline draft (H 7168, I 40960, Vt 129280)"),
    "qwen3-next-80b-a3b": (QWEN3_NEXT_80B_A3B, 8, 2048, "Qwen3-Next-80B-A3B EAGLE3 offline draft (head_dim 256, 16 / 2 heads)"),
    "gemma3-1b": (GEMMA3_1B, 1, 4096, "Gemma3-1B EAGLE3 offline draft (head_dim 256, 4 / 1 heads, H 1152; configs/gemma3-1b-eagle3.json)"),
}
# The legs of the default line's `configs` object (every BASELINE.json configuration and the batch-1 recipe shape
# under the driver's clock): (key, CONFIGS entry, batch, seq, what it is)
CONFIG_LEGS = [
    ("qwen3-8b_8x2048", "qwen3-8b", 8, 2048, "BASELINE.json configs[2] per-GPU shape (Qwen3-8B draft dims, 8 x 2048)"),
    ("qwen3-30b-a3b-eagle31_1x4096", "qwen3-30b-a3b-eagle31", 1, 4096, "configs[3] at its recipe's shape (EAGLE3.1, fc_norm; batch 1 x 4096)"),
    ("deepseek-v3_1x2048", "deepseek-v3", 1, 2048, "configs[4] at its recipe's shape (H 7168, I 40960, Vt 129280; batch 1 x 2048)"),
    ("llama3-8b_1x4096", "llama3-8b", 1, 4096, "the reference's own Llama recipe (examples/configs/llama3.1-8b-eagle3-offline.yaml: batch 1, max_length 4096)"),
]
SMALL = dict(hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=2, vocab_size=4096,
             draft_vocab_size=1024, head_dim=128, target_hidden_size=512, max_position_embeddings=2048, rms_norm_eps=1e-5)
PEAK_BF16_TFLOPS = 2500.0
MFMA_ONLY_SUSTAINED_TFLOPS = 2038.0     # measured: back-to-back 16x16x32 bf16 MFMAs, random operands, 256 CUs (tools/probes/mfma_power_probe.hip, profiles/r6_mfma_power_probe.jsonl)
PEAK_HBM_GBS = 8000.0          # spec; ~6300 GB/s is what a float4 copy reaches (MI355X_MICROARCH.md)
GEMM_KERNEL_NAME = "gemm_nt_256w4_kernel (bf16 MFMA GEMM, 256x256x64, 4 waves x 128x128, plan-scheduled: one filler per MFMA slot, counted vmcnt)"


class KernelTimer:
    """HIP events around the launches of the step's hot kernels on the current stream (= the launch stream).  Per kernel
    family: algorithmic work (flop or bytes) per launch, summed, against the summed event durations."""

    # family -> (ops attribute, kind, work(args, kwargs) in flop or bytes)
    def __init__(self):
        self.rec = {}
        self._orig = {}

    @staticmethod
    def _attn_units(kw):
        return kw["B"] * kw["nh"] * kw["hd"] * kw["S"] * kw["S"] / 2.0

    @staticmethod
    def _diag_bytes(k):
        N, qw, kw = k["B"] * k["S"], k["nh"] * k["hd"], k["nkv"] * k["hd"]
        own = k.get("q") is not None
        by = (3 * N * qw * 2 if own else 0) + len(k.get("kd", ())) * 2 * N * kw * 2 + len(k.get("xq", ())) * 2 * N * qw * 2
        if own and k.get("dq_init") is not None:
            by += N * qw * 4 * (2 if k.get("dq_accumulate") else 1)
        for j, first in enumerate(k.get("first", ())):
            by += (0 if first else 2 * N * kw * 4) + (2 * N * kw * 2 if k["dk_out"][j] is not None else 2 * N * kw * 4)
        return float(by)

    @staticmethod
    def _pre_bytes(a, k):     # the per-step form (A/B): every branch sum read + written in fp32, the last one written as bf16
        N, qw, kw = k["B"] * k["S"], k["nh"] * k["hd"], k["nkv"] * k["hd"]
        nd = len(a[3])
        launches = max(1, (nd + 3) // 4)
        by = launches * 3 * N * qw * 2 + nd * 2 * N * kw * 2 + nd * 2 * N * kw * 8
        if nd:
            by += N * qw * 4 * (2 * launches - 1)
        return float(by)

    def wrap(self, ops):
        el = lambda t: t.numel() * t.element_size()
        fam = {
            # MFMA-bound: 2*M*N*K
            "gemm_nt": ("gemm_nt", lambda a, k: 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]),
            "gemm_nt_swiglu_bwd": ("gemm_nt_swiglu_bwd", lambda a, k: 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]),
            "gemm_nt_rowadd": ("gemm_nt_rowadd", lambda a, k: 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]),
            "gemm_nt_swiglu_fwd": ("gemm_nt_swiglu_fwd", lambda a, k: 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]),
            "gemm_nt_teacher": ("gemm_nt_teacher", lambda a, k: 2.0 * a[0].shape[0] * a[1].shape[0] * a[0].shape[1]),
            "gemm_tn": ("gemm_tn", lambda a, k: 2.0 * a[0].shape[1] * a[1].shape[1] * a[0].shape[0]),
            # attention: matmuls of B*nh*hd*S^2/2 MACs each: forward 2, dQ 3, dK/dV 4 (each product once)
            "attn_fwd": ("attn_fwd", lambda a, k: 4.0 * self._attn_units(k)),
            "attn_bwd_dq": ("attn_bwd_dq", lambda a, k: 6.0 * self._attn_units(k)),
            "attn_bwd_dkv": ("attn_bwd_dkv", lambda a, k: 8.0 * self._attn_units(k)),
            # HBM-bound: algorithmic bytes
            #   fused CE: logits read + gradient written in place (bf16) for every row, fp32 soft target read for the rows
            #   that carry a position mask (counted from the mask the launch is given)
            #   (recorded per element; summary() applies  4 + 4 * mask density  bytes -- no device read-back per launch)
            "ce_fused": ("ce_fused", lambda a, k: float(a[0].numel())),
            #   ... with the soft target re-formed from the teacher's stored bf16 draft logits: 4 + 2 * mask density bytes per element
            "ce_fused_zt": ("ce_fused_zt", lambda a, k: float(a[0].numel())),
            #   AdamW: grad bf16 r, master / m / v fp32 r+w, param bf16 w = 28 B per parameter
            "adamw_step": ("adamw_step", lambda a, k: 28.0 * a[0].numel()),
            #   diagonal-branch backward: q / o / dO of the own step, k / v of the branches read, dq_init, the streamed steps' q / dO,
            #   and the branch sums (fp32 read unless first touch; bf16 write when final, else fp32)
            "attn_bwd_diag": ("attn_bwd_diag", lambda a, k: self._diag_bytes(k)),
            "attn_bwd_pre": ("attn_bwd_pre", lambda a, k: self._pre_bytes(a, k)),
            #   teacher reduce: one bf16 read of the logits chunk + target_p fp32 write
            "teacher_reduce": ("teacher_reduce", lambda a, k: el(a[0]) + 4.0 * a[0].shape[0] * k["Vd"]),
            #   ... on the permuted head: the stored (draft) logits + 16 B per reduced 128-column block + the probabilities
            "teacher_reduce_perm": ("teacher_reduce_perm", lambda a, k: 2.0 * a[0].numel() + 16.0 * a[0].shape[0] * k.get("nparts", 0)
                                    + (4.0 * a[0].shape[0] * k["Vd"] if k.get("target_p_pad") is not None else 0.0)),
        }
        for name, (attr, work) in fam.items():
            orig = getattr(ops, attr)
            self._orig[attr] = orig
            self.rec[name] = []

            def timed(*a, _orig=orig, _work=work, _name=name, **k):
                w = _work(a, k)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                r = _orig(*a, **k)
                e.record()
                self.rec[_name].append((w, s, e))
                return r

            setattr(ops, attr, timed)

    def unwrap(self, ops):
        for attr, orig in self._orig.items():
            setattr(ops, attr, orig)

    def summary(self, name):
        r = self.rec.get(name, [])
        return sum(x[0] for x in r), sum(x[1].elapsed_time(x[2]) for x in r), len(r)


def make_loss_mask(B, S, density, seed):
    """host [B, S] int64 loss mask: all ones (the headline: every position carries a loss), or chat-like turns -- alternating
    unmasked / masked spans of random length (mean 64 tokens) with the requested fraction of masked-in positions"""
    if density >= 1.0:
        return torch.ones(B, S, dtype=torch.int64)
    g = torch.Generator().manual_seed(seed)
    lm = torch.zeros(B, S, dtype=torch.int64)
    for b in range(B):
        pos, on = 0, False
        while pos < S:
            mean = 64.0 * (2 * density if on else 2 * (1 - density))
            n = max(1, int(torch.empty(1).exponential_(1.0 / max(mean, 1.0), generator=g)))
            if on:
                lm[b, pos:pos + n] = 1
            pos, on = pos + n, not on
    return lm


def make_batch(cfg, B, S, dev, seed, loss_mask=None):
    g = torch.Generator(device=dev).manual_seed(seed)
    Ht = cfg["target_hidden_size"]
    return dict(
        input_ids=torch.randint(0, cfg["vocab_size"], (B, S), device=dev, generator=g),
        attention_mask=torch.ones(B, S, dtype=torch.int64, device=dev),
        loss_mask=torch.ones(B, S, dtype=torch.int64, device=dev) if loss_mask is None else loss_mask.to(dev),
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
                       f"1 micro-step fwd+bwd, B=1 x S={S} of the same model dims and ttt, fp32, {dt:.1f} s; the headline step is 8 "
                       f"such samples (x8 the work at the same rate).  The reference trainer itself, timed on the build "
                       f"container's 8 cores: profiles/old/r3_reference_cpu_trainer.jsonl (re-timed: profiles/r5_reference_cpu_trainer.jsonl) / BASELINE.md section 5")


def f_draft_per_token(cfg, S, ttt):
    """F_draft = 3 (T F_step + 2 * 3Ht * H) flop per token"""
    H, I, hd, nh, nkv = (cfg[k] for k in ("hidden_size", "intermediate_size", "head_dim", "num_attention_heads", "num_key_value_heads"))
    f_step = 2.0 * (2 * H * nh * hd + 2 * 2 * H * nkv * hd + nh * hd * H + 3 * H * I + H * cfg["draft_vocab_size"]) + 4.0 * nh * hd * (S / 2.0)
    return 3.0 * (ttt * f_step + 2.0 * 3 * cfg["target_hidden_size"] * H)


def config_leg(cfg, B, S, ttt, nsteps, dev, rank, world, timed, KernelTimer, ops, K):
    """one `configs` leg: a fresh draft + engine + fused optimizer at ``cfg``'s dims, B x S synthetic HBM-resident batches, 2 warm-up +
    ``nsteps`` timed optimizer steps -> {ms_per_step, tokens_per_s, draft_frac, nt_frac, kernel fractions}"""
    if S + ttt > cfg["max_position_embeddings"] + 20:
        cfg = dict(cfg, max_position_embeddings=S + ttt)
    torch.manual_seed(0)
    model = K["LlamaForCausalLMEagle3"](K["DraftConfig"](**cfg), device=dev)
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = K["OnlineEagle3Model"](model, length=ttt).train()
    head = K["TargetHead"]((torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16))
    strat = K["Eagle3TrainStrategy"](eagle, target_head=head)
    be = K["HipDPTrainingBackend"](optimizer_factory=lambda m: K["BF16Optimizer"](m, lr=1e-4, max_grad_norm=0.5, total_steps=10_000))
    be.prepare_model(eagle)
    lm = torch.ones(B, S, dtype=torch.int64)
    bt = [K["TrainBatch"](make_batch(cfg, B, S, dev, 100 + rank * 10 + i, lm),
                          {"target_repr": "hidden_state", "loss_mask_suffix_counts": K["counts"](lm)}) for i in range(2)]
    nb = lambda i: bt[i % 2]
    timed(strat, 2, next_batch=nb, be=be)
    timer = KernelTimer()
    el, out = timed(strat, nsteps, timer, next_batch=nb, be=be)
    tokens = world * B * S * nsteps
    res = {"batch": B, "seq_len": S, "ms_per_step": 1e3 * el / nsteps, "tokens_per_s": tokens / el,
           "draft_frac": f_draft_per_token(cfg, S, ttt) * (tokens / world) / el / 1e12 / PEAK_BF16_TFLOPS,
           "final_loss": float(out.loss.detach()), "hbm_gb": eagle.engine.arena_bytes() / 1e9}
    fr = {}
    for name in ("gemm_nt", "gemm_nt_swiglu_bwd", "gemm_nt_rowadd", "gemm_nt_swiglu_fwd", "gemm_nt_teacher", "gemm_tn", "attn_fwd", "attn_bwd_dq",
                 "attn_bwd_dkv"):
        w, ms, n = timer.summary(name)
        if n and ms > 0:
            fr[name] = {"frac": w / ms / 1e9 / PEAK_BF16_TFLOPS, "ms_per_step": ms / nsteps}
    res["nt_frac"] = fr.get("gemm_nt", {}).get("frac")
    res["kernels"] = fr
    del bt, be, strat, head, eagle, model
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="llama3-8b", choices=sorted(CONFIGS),
                    help="model dims + default batch x seq (llama3-8b = BASELINE.json configs[1], the headline)")
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--seq", type=int, default=None)
    ap.add_argument("--ttt", type=int, default=7)
    ap.add_argument("--small", action="store_true", help="tiny model dims (smoke / debugging only; NOT the headline config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seq", type=int, default=2048,
                    help="sequence length of the CPU baseline sample (B = 1; the headline step is 8 such samples)")
    ap.add_argument("--no-dense-mask", action="store_true", help="skip the dense-position-mask variant")
    ap.add_argument("--dense-mask", action="store_true", help="run the dense-position-mask variant also when world > 1 (default: only at N = 1)")
    ap.add_argument("--feed", default="hbm", choices=["hbm", "ingest", "cpu_batch", "cpu_batch_pageable"],
                    help="where the TIMED region's batches come from: hbm = already resident (the metric's definition); ingest = feature "
                         "files -> pinned slots -> copy stream (what `specforge train` gets through reference_plugin); cpu_batch = CPU "
                         "(pageable) batches as the reference's FeatureDataLoader hands them over, staged by the strategy through pinned "
                         "slots; cpu_batch_pageable = the reference's own blocking pageable .to(device)")
    ap.add_argument("--no-feeds", action="store_true", help="skip the feed comparison legs (hbm / ingest / cpu_batch / cpu_batch_pageable)")
    ap.add_argument("--feeds", action="store_true", help="run the feed comparison legs also when world > 1")
    ap.add_argument("--feed-steps", type=int, default=None, help="steps per feed leg (default: min(steps, 6))")
    ap.add_argument("--dp-single", action="store_true", help="one gradient all-reduce after the sweep instead of overlapped buckets (A/B)")
    ap.add_argument("--dp-early-lm-head", action="store_true",
                    help="A/B: the lm_head weight gradient (the largest bucket) and its all-reduce BEFORE the data-gradient sweep "
                         "(engine.early_lm_head_wgrad; costs one host wait for the upstream gradient)")
    ap.add_argument("--force-dp", action="store_true", help="run the gradient collectives even at world size 1 (RCCL path on a 1-GPU box)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, the product path) | gloo (launcher smoke test)")
    ap.add_argument("--share-gpu", action="store_true", help="TEST ONLY: all ranks on cuda:0 (with --dist-backend gloo)")
    ap.add_argument("--diag-per-step", action="store_true",
                    help="A/B: the diagonal-branch backward as one sf_attn_bwd_pre per TTT step (round 3) instead of the blocked sf_attn_bwd_diag")
    ap.add_argument("--loss-mask-density", type=float, default=1.0,
                    help="fraction of positions that carry a loss (default 1.0 = the headline workload: all ones).  < 1: chat-like spans; "
                         "the engine runs lm_head / CE / lm_head gradients on those rows only (see --no-compact)")
    ap.add_argument("--no-compact", action="store_true", help="A/B: the dense lm_head part also for sparse loss masks")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` legs (cfg 3 / 4 / 5 and the batch-1 recipe shape beside the headline)")
    ap.add_argument("--configs", action="store_true", help="run the `configs` legs also when world > 1 or with a non-default --config / shape")
    ap.add_argument("--config-steps", type=int, default=4, help="timed steps per `configs` leg (2 warm-up steps before them)")
    ap.add_argument("--materialise-targets", action="store_true",
                    help="A/B: write the fp32 soft targets [B, S+T, Vd] instead of re-forming them in the fused CE from the teacher's draft logits")
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
    cfg_dims, B0, S0, cfg_label = CONFIGS[args.config]
    cfg = SMALL if args.small else cfg_dims
    B, S = args.batch or B0, args.seq or S0
    if S + args.ttt > cfg["max_position_embeddings"] + 20:      # the engine's RoPE table has max_position_embeddings + 20 rows
        cfg = dict(cfg, max_position_embeddings=S + args.ttt)
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
    eagle.engine.materialise_soft_targets = args.materialise_targets
    eagle.engine.blocked_diag = not args.diag_per_step
    eagle.engine.early_lm_head_wgrad = args.dp_early_lm_head
    eagle.engine.compact_loss_rows = not args.no_compact
    from specforge_amd.eagle3 import loss_mask_suffix_counts

    def resident_batch(i, density):
        lm = make_loss_mask(B, S, density, 7 + rank * 10 + i)
        # (the row counts a loader computes while the mask is in host memory -- HiddenStateIngest / the strategy for CPU batches)
        return TrainBatch(make_batch(cfg, B, S, dev, 100 + rank * 10 + i, lm),
                          {"target_repr": "hidden_state", "loss_mask_suffix_counts": loss_mask_suffix_counts(lm)})

    batches = [resident_batch(i, args.loss_mask_density) for i in range(2)]

    # ---- where a step's batch comes from (--feed).  hbm: resident (the metric).  The others are what a
    # `specforge train` run sees: feature files through the HIP ingest, or CPU batches as the reference's loader hands them over.
    feed_state = {}

    def make_feed(kind):
        """-> (strategy, next_batch(i), close())"""
        if kind == "hbm":
            return strat, (lambda i: batches[i % 2]), (lambda: None)
        if kind in ("cpu_batch", "cpu_batch_pageable"):
            if "cpu" not in feed_state:      # pageable host tensors, as DataCollatorWithPadding's torch.cat produces them
                feed_state["cpu"] = [TrainBatch({k: v.cpu() for k, v in b.tensors.items()}, dict(b.metadata)) for b in batches]
            st = strat if kind == "cpu_batch" else Eagle3TrainStrategy(eagle, target_head=head, pinned_staging=False)
            return st, (lambda i: feed_state["cpu"][i % 2]), (lambda: None)
        # ingest: the batches as feature files in the reference's offline format (scripts/prepare_hidden_states.py:446-480)
        import shutil
        import tempfile

        from specforge_amd.ingest import HiddenStateIngest

        if "dir" not in feed_state:
            d = feed_state["dir"] = tempfile.mkdtemp(prefix=f"sf_bench_feed_r{rank}_", dir="/tmp")
            files = []
            for bi, b in enumerate(batches):
                t = {k: v.cpu() for k, v in b.tensors.items()}
                for j in range(B):
                    fpath = os.path.join(d, f"{bi:02d}_{j:03d}.ckpt")
                    torch.save({"input_ids": t["input_ids"][j].clone(), "loss_mask": t["loss_mask"][j].clone(),
                                "hidden_state": t["target"][j:j + 1].clone(), "aux_hidden_state": t["hidden_state"][j:j + 1].clone()}, fpath)
                    files.append(fpath)
            feed_state["files"] = files
            feed_state["ingest"] = HiddenStateIngest(files, batch_size=B, max_len=S, device=dev, shuffle=False)
        files, ing = feed_state["files"], feed_state["ingest"]
        groups = [files[:B], files[B:2 * B]] * 4096
        it = ing.stream(groups)

        def close():
            it.close()

        return strat, (lambda i: next(it)), close

    def cleanup_feeds():
        if "dir" in feed_state:
            import shutil

            shutil.rmtree(feed_state.pop("dir"), ignore_errors=True)

    rank_spread = {}      # (world > 1) slowest / fastest rank of the last timed() call: a straggler shows up in the scaling record

    def timed(strategy, nsteps, timer=None, next_batch=None, be=None):
        """W warm-up steps were done by the caller; times exactly nsteps optimizer steps: barrier + synchronize on
        both sides, MAX over ranks."""
        nb = next_batch or (lambda i: batches[i % 2])
        be = be or backend

        def step(i):
            out = strategy.forward_loss(nb(i))
            be.backward(out.loss, is_boundary=True)
            be.step()
            return out

        if timer is not None:
            timer.wrap(ops)
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
        if timer is not None:
            timer.unwrap(ops)
        if world > 1:
            t = torch.tensor([elapsed, -elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            rank_spread["max_s"], rank_spread["min_s"], rank_spread["steps"] = float(t[0]), -float(t[1]), nsteps
            elapsed = float(t[0].item())
        return elapsed, out

    main_strat, main_next, main_close = make_feed(args.feed)
    timed(main_strat, args.warmup, next_batch=main_next)          # untimed warm-up
    timer = KernelTimer()
    backend.comm_wait_events = []                    # (N > 1) event pairs around the waits for the bucket all-reduces
    backend.bucket_events = [] if dist.is_initialized() else None      # (N > 1) per bucket: gradient complete -> all-reduce finished
    elapsed, out = timed(main_strat, args.steps, timer, next_batch=main_next)
    bucket_rec, backend.bucket_events = backend.bucket_events, None     # (the legs below must not append to the timed region's record)
    main_close()
    main_spread = dict(rank_spread)
    hbm_peak_main = torch.cuda.max_memory_allocated(dev) / 1e9
    loss = float(out.loss.detach())
    mask_density = float(eagle.last_artifacts["position_mask"].float().mean())   # of the timed steps (before the dense-mask variant)
    fl, gemm_ms, nlaunch = timer.summary("gemm_nt")
    tokens = world * B * S * args.steps

    # ---- dense position mask variant: the specified synthetic vocab map (random 32000 of 128256)
    # leaves ~25 % of the rows with a position mask, and the fused CE skips the soft-target read of masked rows.  Here the
    # teacher's argmax always lands inside the draft vocabulary (head rows outside it are zero), so every row reads its
    # target -- the worst case of real data.  Reported beside the headline value, never instead of it.
    dense = None
    if not args.no_dense_mask and (world == 1 or args.dense_mask):
        try:
            hw = head.fc.weight.data.clone()
            hw[~t2d.to(dev)] = 0
            strat_dense = Eagle3TrainStrategy(eagle, target_head=TargetHead(hw))
            timed(strat_dense, 1)
            e2, out2 = timed(strat_dense, args.steps)
            pm = eagle.last_artifacts["position_mask"].float().mean()
            dense = {"value": tokens / e2, "ms_per_step": 1e3 * e2 / args.steps, "position_mask_density": float(pm)}
            del strat_dense, hw
        except Exception as e:      # (an optional leg: never at the price of the headline line)
            dense = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the same step under every feed (N = 1 unless --feeds): ms per step, same process, same box, no kernel timers.  The
    # metric's `value` is the --feed of the timed region above (default hbm = the metric's definition); this table is what the
    # `specforge train` path (ingest) and the reference loader's hand-over (cpu_batch*) cost beside it.
    feeds = None
    if not args.no_feeds and (world == 1 or args.feeds):
        fs = args.feed_steps or min(args.steps, 6)
        feeds = {"steps_per_leg": fs}
        for kind in ("hbm", "ingest", "cpu_batch", "cpu_batch_pageable"):
            try:      # a leg that cannot run here (no room for the feature files in /tmp, no pinned memory ...) must not cost the headline line
                st, nb, close = make_feed(kind)
                try:
                    timed(st, 2, next_batch=nb)
                    e3, _ = timed(st, fs, next_batch=nb)
                    feeds[kind] = {"ms_per_step": 1e3 * e3 / fs, "tokens_per_s": world * B * S * fs / e3}
                finally:
                    close()
            except Exception as e:
                feeds[kind] = {"ms_per_step": None, "error": f"{type(e).__name__}: {e}"[:300]}
        for kind in ("ingest", "cpu_batch", "cpu_batch_pageable"):
            if feeds[kind].get("ms_per_step") and feeds["hbm"].get("ms_per_step"):
                feeds[kind]["vs_hbm"] = feeds[kind]["ms_per_step"] / feeds["hbm"]["ms_per_step"]
        feeds["note"] = ("hbm: batches resident (metric definition).  ingest: feature files (page cache) -> preadv into pinned slots -> "
                         "copy stream -> device TrainBatch: what reference_plugin.install() gives `specforge train`.  cpu_batch: pageable "
                         "CPU batches (what the reference's FeatureDataLoader hands over) staged by the strategy through pinned slots + "
                         "copy stream.  cpu_batch_pageable: the reference strategy's own CPU shift + blocking pageable .to(device).")
    cleanup_feeds()

    # ---- sparse loss mask (what real chat data looks like; the headline has all ones): half of the positions carry a loss, in
    # turn-like spans.  compact = the product path (lm_head / CE / lm_head gradients over the masked-in rows only, host-known row
    # counts); dense = the same batches with engine.compact_loss_rows off.  Reported beside the headline, never instead of it.
    sparse = None
    if not args.no_feeds and (world == 1 or args.feeds) and args.loss_mask_density >= 1.0:
        try:
            fs = args.feed_steps or min(args.steps, 6)
            sb = [resident_batch(i, 0.5) for i in range(2)]
            sparse = {"loss_mask_density": float(sum(float(x.tensors["loss_mask"].float().mean()) for x in sb) / 2), "steps_per_leg": fs}
            for name, flag in (("compact", True), ("dense", False)):
                eagle.engine.compact_loss_rows = flag
                timed(strat, 2, next_batch=lambda i: sb[i % 2])
                e4, _ = timed(strat, fs, next_batch=lambda i: sb[i % 2])
                sparse[name] = {"ms_per_step": 1e3 * e4 / fs, "tokens_per_s": world * B * S * fs / e4}
            del sb
        except Exception as e:
            sparse = {"error": f"{type(e).__name__}: {e}"[:300]}
        finally:
            eagle.engine.compact_loss_rows = not args.no_compact

    # ---- RCCL evidence: the gradient all-reduce of each bucket, timed alone on the communicator (outside the timed region)
    rccl = None
    if dist.is_initialized():
        try:
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
            waits = [a.elapsed_time(b) for a, b in getattr(backend, "comm_wait_events", [])]
            rccl = {"backend": args.dist_backend, "rccl_ranks": world, "buckets": per,
                    "allreduce_ms_total_unoverlapped": sum(x["ms"] for x in per),
                    # inside the timed steps: how long the compute stream sat in front of the optimizer waiting for the bucket
                    # all-reduces that were launched from inside the weight-gradient phase (0 = fully overlapped)
                    "exposed_wait_ms_per_step": (sum(waits) / max(1, len(waits))) if waits else None,
                    "no_sync_backwards": backend.no_sync_backwards, "single_collective": backend._single_collective,
                    "early_lm_head_wgrad": bool(eagle.engine.early_lm_head_wgrad)}
            # per bucket, inside the timed steps: from "its weight gradient is complete" to "its all-reduce has finished" (mean over the steps),
            # beside the same all-reduce alone (`buckets[i].ms`): the difference is the wait for a CU (a persistent TN workgroup owns its CU
            # for a whole tile) plus the slowdown of sharing the chip with the next weight-gradient GEMM
            tl = backend.bucket_timeline(bucket_rec)
            nb = len(bounds)
            if tl and len(tl) % nb == 0:
                for i in range(nb):
                    xs = [tl[j][1] for j in range(i, len(tl), nb)]
                    per[i]["ready_to_done_ms_in_step"] = sum(xs) / len(xs)
                    per[i]["in_step_minus_alone_ms"] = per[i]["ready_to_done_ms_in_step"] - per[i]["ms"]
        except Exception as e:      # the evidence leg runs AFTER the timed region: a failure here must not cost the line
            rccl = {"backend": args.dist_backend, "rccl_ranks": world, "error": f"{type(e).__name__}: {e}"[:300]}

    # ---- the other BASELINE.json configurations and the batch-1 recipe shape, same process, same box, same method (a fresh model + engine +
    # optimizer per leg; 2 warm-up + --config-steps timed optimizer steps with the HIP-event kernel timers on).  Beside the headline, never
    # instead of it; every leg in its own try/except.
    configs_out = None
    default_shape = (args.config == "llama3-8b" and not args.small and args.batch is None and args.seq is None and args.ttt == 7
                     and args.loss_mask_density >= 1.0 and args.feed == "hbm")
    if not args.no_configs and ((world == 1 and default_shape) or args.configs):
        configs_out = {"steps_per_leg": args.config_steps, "warmup_per_leg": 2,
                       "note": "ms_per_step = one optimizer step (teacher + 7 TTT steps fwd/bwd + grad-norm/clip/AdamW) on HBM-resident synthetic "
                               "batches; draft_frac = SURVEY 8d F_draft against 2500 TFLOP/s; the per-kernel fractions are HIP-event timed like `kernels`"}
        del batches, main_strat, main_next
        eagle.engine._arena.clear()          # the headline model's 50 GB of step buffers: the legs bring their own
        eagle.engine._views.clear()
        torch.cuda.empty_cache()
        for key, cname, Bc, Sc, what in CONFIG_LEGS:
            try:
                configs_out[key] = config_leg(SMALL if args.small else CONFIGS[cname][0], Bc, Sc, args.ttt, args.config_steps, dev, rank, world, timed, KernelTimer, ops,
                                              dict(Eagle3TrainStrategy=Eagle3TrainStrategy, OnlineEagle3Model=OnlineEagle3Model, TargetHead=TargetHead,
                                                   TrainBatch=TrainBatch, DraftConfig=DraftConfig, LlamaForCausalLMEagle3=LlamaForCausalLMEagle3,
                                                   BF16Optimizer=BF16Optimizer, HipDPTrainingBackend=HipDPTrainingBackend,
                                                   counts=loss_mask_suffix_counts))
                configs_out[key]["what"] = what
            except Exception as e:
                configs_out[key] = {"ms_per_step": None, "what": what, "error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()

    if rank == 0:
        ach = fl / gemm_ms / 1e9 if gemm_ms > 0 else 0.0
        # HBM-side traffic of the GEMM kernel per launch: from the committed rocprofv3 --pmc passes of this same
        # command (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md, calibrated there on the AdamW
        # kernel's known byte count).  PMC counters cannot be collected from inside the timed run, so this field is the
        # committed per-launch figure (newest profiles/r*_pmc_fetch_write_summary.json) or null.
        traffic, traffic_src, traffic_commit = None, None, None
        try:
            import glob

            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_write_summary.json")))[-1]
            pm = json.load(open(src))["gemm256w4"]
            traffic = (2.0 * pm["FETCH_SIZE_sum"] + pm["WRITE_SIZE_sum"]) * 1024.0 / pm["launches"]
            traffic_src = os.path.relpath(src, ROOT)
            traffic_commit = json.load(open(src)).get("_commit")      # the commit the PMC passes were collected on
        except Exception:
            pass
        st = max(1, args.steps)
        density = mask_density

        def kern(name, unit, peak, scale=1.0, what=""):
            w, ms, n = timer.summary(name)
            if n == 0 or ms <= 0:
                return None
            ach = w * scale / ms / 1e9 if unit == "TFLOP/s" else w * scale / ms / 1e6
            return {"bound": "mfma" if unit == "TFLOP/s" else "hbm", "achieved": ach, "peak": peak, "unit": unit,
                    "frac": ach / peak, "launches_per_step": n / st, "ms_per_step": ms / st, "work": what}

        kernels = {
            "gemm_nt_swiglu_bwd": kern("gemm_nt_swiglu_bwd", "TFLOP/s", PEAK_BF16_TFLOPS, what="2MNK; d(SwiGLU) in the epilogue"),
            "gemm_nt_rowadd": kern("gemm_nt_rowadd", "TFLOP/s", PEAK_BF16_TFLOPS, what="2MNK; fp32 row addend (hoisted embedding half of QKV)"),
            "gemm_nt_swiglu_fwd": kern("gemm_nt_swiglu_fwd", "TFLOP/s", PEAK_BF16_TFLOPS, what="2MNK, N = 2I; SwiGLU forward in the epilogue (gate|up and act stored)"),
            "gemm_nt_teacher": kern("gemm_nt_teacher", "TFLOP/s", PEAK_BF16_TFLOPS,
                                    what="2MNK, N = Vt; columns past the draft sub-vocabulary reduced in the epilogue (max / sum-exp / argmax per 128-column block), not stored"),
            "gemm_tn": kern("gemm_tn", "TFLOP/s", PEAK_BF16_TFLOPS, what="2MNK, K = T*N token rows (deferred weight gradients; split-K reduce included)"),
            "attn_fwd": kern("attn_fwd", "TFLOP/s", PEAK_BF16_TFLOPS, what="4 B nh hd S^2/2 (diagonal branches not counted)"),
            "attn_bwd_dq": kern("attn_bwd_dq", "TFLOP/s", PEAK_BF16_TFLOPS, what="6 B nh hd S^2/2"),
            "attn_bwd_dkv": kern("attn_bwd_dkv", "TFLOP/s", PEAK_BF16_TFLOPS, what="8 B nh hd S^2/2 (each product once)"),
            "attn_bwd_diag": kern("attn_bwd_diag", "GB/s", PEAK_HBM_GBS,
                                  what="diagonal-branch backward, blocked: q / o / dO + branch k / v + dq_init + streamed later steps + branch sums (first touch / bf16 final)"),
            "attn_bwd_pre": kern("attn_bwd_pre", "GB/s", PEAK_HBM_GBS, what="diagonal-branch backward, one pair per step (A/B form, --diag-per-step)"),
            "ce_fused": kern("ce_fused", "GB/s", PEAK_HBM_GBS, scale=4.0 + 4.0 * density,
                             what=f"per logit: 2 B read + 2 B gradient written in place + 4 B soft target on the {density:.2f} of rows with a position mask"),
            "ce_fused_zt": kern("ce_fused_zt", "GB/s", PEAK_HBM_GBS, scale=4.0 + 2.0 * density,
                                what=f"per logit: 2 B read + 2 B gradient written in place + 2 B of the teacher's stored draft logit on the {density:.2f} of rows with a position mask (target_p re-formed, never materialised)"),
            "adamw_step": kern("adamw_step", "GB/s", PEAK_HBM_GBS, what="28 B per parameter"),
            "teacher_reduce": kern("teacher_reduce", "GB/s", PEAK_HBM_GBS, what="2 B per target logit read + 4 B per draft-vocabulary probability written"),
            "teacher_reduce_perm": kern("teacher_reduce_perm", "GB/s", PEAK_HBM_GBS,
                                        what="2 B per STORED (draft) logit + 16 B per reduced column block (+ 4 B per draft-vocabulary probability when target_p is materialised)"),
        }
        fus = kernels["gemm_nt_swiglu_bwd"] or {}
        # the step as a whole against the MFMA roofline: SURVEY 8d's F_draft = 3 * (T * F_step + 2 * 3Ht * H) per token
        f_draft = f_draft_per_token(cfg, S, args.ttt)
        draft_tflops = f_draft * (tokens / world) / elapsed / 1e12
        line = {
            "metric": ("EAGLE3 draft train tokens/sec, Llama-3-8B target, seq2048 at 1/2/4/8 MI355X" if args.config == "llama3-8b" else
                       f"EAGLE3 draft train tokens/sec, {args.config} target dims, seq{S}"),
            "value": tokens / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": ("SMALL-debug" if args.small else cfg_label)
                       + f", bf16, per-GPU batch {B} x seq {S}, ttt {args.ttt}, optimizer step included"
                       + ("" if args.loss_mask_density >= 1.0 else f", loss mask density {args.loss_mask_density} (NOT the headline workload)"
                          + (", dense lm_head" if args.no_compact else "")),
                       "global_batch": world * B, "seq_len": S, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "kernel": GEMM_KERNEL_NAME, "achieved": ach,
                         "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_commit": traffic_commit,
                         "launches_per_step": nlaunch / st,
                         "fused_swiglu_dgrad": {"launches_per_step": fus.get("launches_per_step"), "ms_per_step": fus.get("ms_per_step"),
                                                "gemm_tflops": fus.get("achieved")},
                         "gemm_ms_per_step": gemm_ms / st,
                         # what back-to-back v_mfma_f32_16x16x32_bf16 on random bf16 operands sustain on this part with nothing else in the loop
                         # (power-limited clock 1.94 GHz, not the 2.4 GHz `peak` assumes): the practical ceiling of any bf16 GEMM here.  A measured
                         # constant (tools/probes/mfma_power_probe.hip), NOT the `peak` this line is priced against.
                         "mfma_only_sustained": {"tflops": MFMA_ONLY_SUSTAINED_TFLOPS, "frac_of_it": ach / MFMA_ONLY_SUSTAINED_TFLOPS,
                                                 "source": "profiles/r6_mfma_power_probe.jsonl"},
                         # (the driver's record keeps `roofline` whole: every hot kernel and every `configs` leg in short form)
                         "by_kernel": dict({"gemm_nt": [round(ach / PEAK_BF16_TFLOPS, 4), round(gemm_ms / st, 3)]},
                                           **{k: [round(v["frac"], 4), round(v["ms_per_step"], 3)] for k, v in kernels.items() if v is not None}),
                         "by_kernel_fields": "[fraction of the bound's peak (2500 TFLOP/s MFMA or 8000 GB/s HBM), ms per step], HIP-event timed over the timed region",
                         "by_config": ({k: [round(v["ms_per_step"], 2), round(v["tokens_per_s"], 1), round(v["draft_frac"], 4),
                                            round(v["nt_frac"], 4) if v.get("nt_frac") else None,
                                            {kk: round(vv["frac"], 3) for kk, vv in v.get("kernels", {}).items() if kk != "gemm_nt"}]
                                        for k, v in configs_out.items() if isinstance(v, dict) and v.get("ms_per_step")}
                                       if configs_out is not None else None),
                         "by_config_fields": "[ms per optimizer step, tokens/s, draft fwd+bwd fraction of the MFMA peak, plain NT GEMM fraction, {other MFMA kernels: fraction}]"},
            # every other hot kernel of the step, same method (HIP events on the launch stream over the timed region)
            "kernels": {k: v for k, v in kernels.items() if v is not None},
            "roofline_hbm": kernels["ce_fused_zt"] or kernels["ce_fused"],
            "draft_fwd_bwd": {"tflops_per_gpu": draft_tflops, "frac_of_mfma_peak": draft_tflops / PEAK_BF16_TFLOPS,
                              "flop_per_token": f_draft, "formula": "SURVEY 8d F_draft = 3 (T F_step + 2 * 3Ht * H)"},
            "final_loss": loss,
            "hbm_peak_gb": hbm_peak_main,
        }
        if main_spread:
            line["rank_ms_per_step"] = {"min": 1e3 * main_spread["min_s"] / main_spread["steps"], "max": 1e3 * main_spread["max_s"] / main_spread["steps"],
                                        "note": "fastest / slowest rank's own wall time over the timed region (value uses the max)"}
        if configs_out is not None:
            line["configs"] = configs_out
        line["feed"] = args.feed
        if feeds is not None:
            line["feeds"] = feeds
        if sparse is not None:
            line["sparse_loss_mask"] = sparse
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
