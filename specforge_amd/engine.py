"""The EAGLE3 TTT micro-step on the HIP kernels: teacher targets, 7x (embed+norms, fused QKV,
RoPE, TTT attention, O, SwiGLU MLP, lm_head, fused CE) forward, and the hand-scheduled
backward sweep with deferred, K-concatenated weight gradients.

Reference call stack this replaces: ``Eagle3TrainStrategy.forward_loss``
(specforge/training/strategies/base.py:237-304) -> ``OnlineEagle3Model.forward``
(specforge/algorithms/eagle3/model.py:244-442) -> ``LlamaForCausalLMEagle3`` /
``LlamaDecoderLayer`` (specforge/modeling/draft/llama3_eagle.py:1598-1798) ->
``LogSoftmaxLoss`` (specforge/core/loss.py:173-228) -> autograd backward.

Design notes (MI355X-first, 288 GB HBM):
* No autograd graph: every activation the backward needs lives in a persistent stash that is
  allocated once per (B, S) shape; step k's K/V are views into step k's fused QKV buffer.
* The soft-target CE writes d(logits) in place while computing the loss, so the lm_head
  input gradient is taken in the forward sweep.
* Weight gradients are NOT computed per TTT step.  Every kernel writes its output into slot k of
  a natural-layout stash [T*N, features]; after the sweep ONE sf_gemm_tn per weight contracts
  dW = dY^T . X over K = T*N token rows with fp32 accumulation and writes the bf16 gradient
  straight into the flat gradient buffer, largest first -- each finished bucket is handed to
  the DP backend, whose RCCL all-reduce overlaps the next GEMM.  No operand is ever transposed.
* The embedding half of the QKV projection (same token, shifted by the step index) is computed
  once over the padded positions and joins each step's accumulator (sf_gemm_nt_rowadd); its
  backward is contracted once from the fp32 sum of the re-aligned step gradients.
* The upstream gradient g = dLoss/d(sum_k decay^k ploss_k) enters only as the alpha of those
  final GEMMs: everything before is linear in it.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import torch

from . import ops
from .model import DraftConfig, FlatParams, LlamaForCausalLMEagle3, rope_tables


def diag_plan(T: int, NA: int = ops.DIAG_ACC, NR: int = ops.DIAG_READ, NX: int = ops.DIAG_X):
    """Launches of ``sf_attn_bwd_diag`` per backward sweep step (the TTT diagonal terms: pair (step k, branch i <= k); reference
    blueprint llama3_eagle.py:1132-1143).  dq of step k needs every pair (k, i) AT sweep step k; dK_i / dV_i are final at sweep
    step i, and a pair may add to them at any sweep step in [i, k].  Branches are grouped in blocks of ``NA``; at the sweep step of
    a block's HIGHEST branch ("top") one launch gives the block's branches every step from the top upwards -- the own step plus the
    later ones streamed again -- first touch, no read of the sums; the pairs inside a block (k below its top) are added at their own
    step k (read-modify-write).  Per step: ``[dict(own, read, nacc, first, final, stream, dq_accumulate)]`` -- ``read`` lists the
    branches whose k / v the launch reads (the accumulating ``nacc`` first), ``stream`` the later steps streamed into them."""
    plan = {}
    for s in range(T - 1, -1, -1):
        if s == 0:
            plan[s] = [dict(own=True, read=[], nacc=0, first=[], final=[], stream=[], dq_accumulate=False)]
            continue
        lo = (s - 1) // NA * NA + 1
        top = min(lo + NA - 1, T - 1)
        acc = list(range(s, lo - 1, -1))                       # the branches of s's block that exist at step s, s itself first
        others = [i for i in range(1, s + 1) if i not in acc]
        stream = list(range(s + 1, T)) if s == top else []
        chunks = [stream[i:i + NX] for i in range(0, len(stream), NX)] or [[]]
        launches = []
        for ci, ch in enumerate(chunks):                       # launch 0 = the own step; further ones only stream
            launches.append(dict(own=ci == 0, read=(acc + others[:NR - len(acc)]) if ci == 0 else list(acc), nacc=len(acc),
                                 first=[s == top and ci == 0] * len(acc),
                                 final=[i == s and ci == len(chunks) - 1 for i in acc], stream=ch, dq_accumulate=False))
        rest = others[NR - len(acc):]
        for i in range(0, len(rest), NR):                      # more than NR branches (ttt_length > 9): dq of the own step in chunks
            launches.append(dict(own=True, read=rest[i:i + NR], nacc=0, first=[], final=[], stream=[], dq_accumulate=True))
        plan[s] = launches
    return plan


class Eagle3Engine:
    # the norm weights every TTT step differentiates (input_layernorm sees the hoisted embedding rows once, fc_norm runs once)
    _NORMS_PER_STEP = ("norm.weight", "midlayer.hidden_norm.weight", "midlayer.post_attention_layernorm.weight")

    def __init__(self, model: LlamaForCausalLMEagle3, *, ttt_length: int = 7, ploss_decay: float = 0.8,
                 teacher_rows: int = 16384, lk_loss_type: Optional[str] = None, kl_scale: float = 1.0,
                 kl_decay: float = 1.0):
        self.model = model
        self.cfg: DraftConfig = getattr(model, "draft_config", None) or model.config   # (the plugin class keeps the HF config in .config)
        c = self.cfg
        if lk_loss_type not in (None, "alpha", "lambda"):
            raise ValueError(f"Unknown lk loss type: {lk_loss_type}")  # core/lk_loss.py:99
        self.lk_loss_type, self.kl_scale, self.kl_decay = lk_loss_type, float(kl_scale), float(kl_decay)
        # The MFMA attention kernels are instantiated for head_dim 64, 128 and 256 (256: gemma3-1b, qwen3-next-80b-a3b,
        # qwen3.5-35b-a3b among the reference's configs/*.json; its attention takes any head_dim, llama3_eagle.py:547-550).
        # Other widths (the reference's own test fixture uses 16: tests/test_runtime/_fixtures.py:15-34) run through the same
        # kernels on zero-padded heads: extra zero columns change neither q.k nor the softmax, and their outputs / gradients
        # are dropped again.  That path copies q / k / v / o per step with torch ops -- correct and slow, for small models
        # only; 64 / 128 / 256 never take it.
        hd = c.head_dim
        if hd % 16 != 0 or hd > 256:
            raise NotImplementedError("head_dim must be a multiple of 16 and <= 256 (64 / 128 / 256 native, other widths zero-padded)")
        self.hdp = hd if hd in (64, 128, 256) else (64 if hd < 64 else 128 if hd < 128 else 256)
        self.T = int(ttt_length)
        self._diag_plan = diag_plan(self.T)      # launches of the blocked diagonal-branch backward, per sweep step
        self.blocked_diag = True                 # False (A/B, bench.py --diag-per-step): one sf_attn_bwd_pre per step, every pair at its step
        self.norm_colsum_batched = True          # False (A/B): one column sum per norm backward launch instead of one per weight and sweep
        # Loss-row compaction (round 4): lm_head forward, the fused CE and the lm_head input / weight gradients -- a third of the step's
        # flops -- run only on the rows of a TTT step that carry a loss mask (real data: the assistant turns; padding never).  Needs the
        # per-step row counts on the HOST (forward(loss_counts=...): the ingest has the mask in host memory; no device read-back);
        # without them, or with fewer than 10 % of the rows masked out, the dense form runs.  Same losses / metrics / gradients: masked
        # rows contribute exact zeros either way (bench.py --loss-mask-density D measures it; False = always dense).
        self.compact_loss_rows = True
        self._lm_compact_K = None                # rows of the compact lm_head stash of the last training forward (None: dense)
        self._teacher_compacted = False          # last forward: target ids / soft targets exist only where the loss mask is set
        if not 1 <= self.T <= ops.MAX_DIAG + 1:
            raise ValueError(f"ttt_length must be in 1..{ops.MAX_DIAG + 1} (one diagonal branch per earlier TTT step)")
        self.decay = float(ploss_decay)
        self.flat = FlatParams(model)
        self.dev = self.flat.data.device
        # rows per teacher-head launch.  Round 6: 16384 (was 4096) -- at the headline one launch of 64 x 501 tiles (125.25 rounds of 256) instead of
        # four of 31.3 rounds each: three partly filled last rounds and three reduce launches fewer, -0.35 ms per step (profiles/r6_teacher_rows_ab.jsonl);
        # the per-row block records grow to 12 KB x rows (197 MB), the full-vocabulary scratch of the materialised forms to 2 x Vt x rows bytes
        self.teacher_rows = teacher_rows
        self.teacher_compact_rows = 4096
        cos, sin = rope_tables(c, torch.bfloat16)
        self.cos, self.sin = cos.to(self.dev), sin.to(self.dev)
        # the reference's rotary module REBUILDS its cos / sin cache when a step's seq_len = S + k exceeds the cached length
        # (llama3_eagle.py:303-306, called with seq_len = q_len + lck at 733): `_rope_len` is its max_seq_len_cached (_rope_steps)
        self._rope_len = c.max_position_embeddings + 20
        # device-side input checks whose verdict is read back with the upstream gradient at the END of the backward sweep (no host
        # sync in the step): slot 0 = loss_counts disagree with the loss mask, slot 1 = position ids outside the reference's range
        self.early_lm_head_wgrad = False         # see backward(): the lm_head weight gradient + its bucket before the data-gradient sweep
        self.max_rope_positions = 1 << 20        # upper bound for a RoPE table grown from host-known mrope ids (see forward)
        self._flags = torch.zeros(2, dtype=torch.float32, device=self.dev)
        self._flags_set = False
        # multimodal rope (llama3_eagle.py:389-427, 145-182): [3, B, S] position ids; rotary channel d of a head takes its
        # angle from position axis mrope_axis[d].  The rope kernel is unchanged: per TTT step it is handed cos / sin tables
        # with ONE ROW PER TOKEN (gathered from the plain tables with torch indexing -- table preparation, [N, hd] bf16).
        rs = c.rope_scaling or {}
        self.mrope = rs.get("rope_type", rs.get("type")) == "mrope"
        if self.mrope:
            sec = list(rs["mrope_section"]) * 2
            if sum(sec) != hd:
                raise ValueError("mrope_section must sum to head_dim / 2")
            self._mrope_axis = torch.cat([torch.full((n,), i % 3, dtype=torch.long) for i, n in enumerate(sec)]).to(self.dev)
        self._arena: Dict[str, torch.Tensor] = {}   # name -> flat storage shared by every batch shape
        self._views: Dict = {}                       # (B, S) -> dict of views into the arena (LRU, bounded)
        self._max_cached_shapes = 16
        self._regrown = False
        self._active = None                          # the shape whose constants are currently laid down
        self._pos_default = None                     # the (B, S) whose default position ids 0..S-1 sit in the shared `pos` buffer
        self._tp_state = None                        # (B, S, storage) whose constant tails the materialised soft targets carry
        self._soft = None                            # ("tp", tensor) | ("zt", (zd, zmd, zinv)): the last forward's soft-target form
        self.materialise_soft_targets = False        # True: always write target_p [B, S+T, Vd] fp32 (A/B and inspection)
        self._wt_version = -1
        self.weights_version = 0          # bumped by the optimizer after every step
        self.micro_in_window = 0          # micro-steps accumulated into flat.grad since the last optimizer step
        self.on_bucket_ready: Optional[Callable[[int, int], None]] = None
        self._t2d_u8 = None
        self._fwd_state = None
        H, I, hd = c.hidden_size, c.intermediate_size, c.head_dim
        self.QW = (c.num_attention_heads + 2 * c.num_key_value_heads) * hd
        self.w_qkv = self.flat.fused("midlayer.self_attn.q_proj.weight", "midlayer.self_attn.v_proj.weight", self.QW, 2 * H)
        self.w_gu = self.flat.fused("midlayer.mlp.gate_proj.weight", "midlayer.mlp.up_proj.weight", 2 * I, H)
        self.g_qkv = self.flat.fused("midlayer.self_attn.q_proj.weight", "midlayer.self_attn.v_proj.weight", self.QW, 2 * H, grad=True)
        self.g_gu = self.flat.fused("midlayer.mlp.gate_proj.weight", "midlayer.mlp.up_proj.weight", 2 * I, H, grad=True)
        # fp32 running totals of the norm-weight gradients over the accumulation window
        self._norm_names = [n for n in self.flat.names if n.endswith("norm.weight") or "layernorm" in n or "fc_norm" in n]
        # (one buffer each, the weights in flat order: when they are also contiguous in the flat gradient -- they are its tail -- the fold at the end
        # of a micro-step is two launches for all of them instead of two per weight)
        sl = [self.flat.slices[n] for n in self._norm_names]
        self._norm_contig = all(sl[i][1] == sl[i + 1][0] for i in range(len(sl) - 1)) and sl[-1][1] == self.flat.numel and sl[0][0] % 8 == 0
        tot = sum(hi - lo for lo, hi in sl)
        self._norm_total_all, self._norm_micro_all = torch.zeros(tot, device=self.dev), torch.zeros(tot, device=self.dev)
        o0 = sl[0][0]
        view = lambda buf: ({n: buf[lo - o0:hi - o0] for n, (lo, hi) in zip(self._norm_names, sl)} if self._norm_contig else
                            {n: torch.zeros(hi - lo, device=self.dev) for n, (lo, hi) in zip(self._norm_names, sl)})
        self._norm_total, self._norm_micro = view(self._norm_total_all), view(self._norm_micro_all)

    # ------------------------------------------------------------------ buffers
    def _e(self, *shape, dtype=torch.bfloat16):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def _carve(self, name: str, *shape, dtype=torch.bfloat16, invalidate: bool = True) -> torch.Tensor:
        """A contiguous view of ``shape`` at the start of the named arena entry, which is (re)allocated only when the
        request exceeds its capacity.  Every batch shape runs inside the SAME storage: variable-length data (the
        collator pads each batch to its own longest sample) never grows HBM beyond the largest shape seen, and
        ``reserve(B, S_max)`` makes that a single allocation up front."""
        n = 1
        for x in shape:
            n *= int(x)
        cur = self._arena.get(name)
        if cur is None or cur.numel() < n or cur.dtype != dtype:
            self._arena.pop(name, None)
            del cur
            if invalidate:               # cached views of other shapes may alias the freed storage
                self._views.clear()
                self._active = None
                self._pos_default = None
                self._regrown = True
            self._arena[name] = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
        return self._arena[name][:n].view(*shape)

    def reserve(self, B: int, S: int) -> None:
        """size every buffer for batches up to [B, S] now (one allocation per buffer; smaller batches are views).  The teacher's scratch is
        carved at the first forward (the head's vocabulary is an argument of the call, not of the engine) -- for these many rows at once."""
        self._reserved_rows = max(getattr(self, "_reserved_rows", 0), B * S)
        self._buffers(B, S)

    def arena_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._arena.values())

    def _buffers(self, B: int, S: int):
        key = (B, S)
        b = self._views.pop(key, None)
        if b is None:
            self._regrown = False
            b = self._build_views(B, S)
            if self._regrown:                    # storage was (re)allocated while carving: earlier views of THIS pass may dangle
                b = self._build_views(B, S)
            while len(self._views) >= self._max_cached_shapes:   # LRU: ragged data brings a new (B, S) almost every step
                self._views.pop(next(iter(self._views)))
        self._views[key] = b                     # (re)inserted last = most recently used
        if self._active != key:
            # per SHAPE SWITCH, not per step: on ragged data (the collator pads a batch to its own longest sample) that is nearly every
            # step -- ~150 view carvings on a cache miss plus the constants below (a few dozen small fills; the padded q / k / v / o
            # copies only with zero-padded heads).  It is host + launch overhead of the ragged path (tools/ragged_bench.py: 96 % of the
            # aligned shape's throughput); bucketing S in the collator (ingest pad_multiple) makes shapes repeat, at the price of the
            # reference's loss normalisation (a mean over ALL B * S rows), so it is not the default.
            self._init_constants(b, B, S)
            self._active = key
        return b

    def _init_constants(self, b, B: int, S: int) -> None:
        """the parts of the (shared) storage whose value the kernels rely on without writing it: padded tails of the
        teacher arrays (eagle3/model.py:445-484: 1/Vd, 0, 0), zero pad rows of the K-concatenated stashes"""
        Vd, T = self.cfg.draft_vocab_size, self.T
        self._tp_state = None      # (the shared `tp` storage, if any, was laid out for another shape: _target_p refills its tails)
        b["tsum"][:, S:].fill_(float(torch.full((Vd,), 1.0 / Vd).sum()))
        for nm in ("pod", "tids", "pm", "lm", "ids"):
            b[nm].zero_()
        b["kvlen"].zero_()
        b["arange"].copy_(torch.arange(B * S, dtype=torch.int32, device=self.dev))
        TN = T * B * S
        for nm in self._stash_names:
            if b["Kp"] > TN:
                b[nm + "_s"][TN:].zero_()
        if b["Kp"] > TN:
            b["h_s"][TN:].zero_()
        # zero pad rows of the K-concatenated operands (the kernels write the real rows every step; only the rows past
        # them must read as zero -- re-zeroing the whole buffers cost ~600 MB of memsets per shape switch at 8B dims)
        Np, Npr, N_ = b["Np"], b["Np_real"], B * S
        for buf in (b["en2"], b["ds2"]):
            buf[Npr:Np].zero_()
            buf[Np + Npr:].zero_()
        b["hs_s"][N_:].zero_()
        b["dh0_s"][N_:].zero_()
        b["metrics"].zero_()
        b["msum"].zero_()
        b["lk_logsum"].zero_()
        if self.hdp != self.cfg.head_dim:    # the pad columns are never written afterwards
            for nm in ("qp", "kp", "vp", "op"):
                for t in b[nm]:
                    t.zero_()
            b["dop"].zero_()
            b["dqp"].zero_()

    _stash_names = ("hn", "o", "pn", "act", "ln", "logits", "dh", "dgu", "dh1", "dqkv")

    def _build_views(self, B: int, S: int):
        c, T = self.cfg, self.T
        N, Spad = B * S, S + T
        H, I, hd, nh, nkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        Vd, Ht3 = c.draft_vocab_size, 3 * c.target_hidden_size
        f32, i32, i64, bf = torch.float32, torch.int32, torch.int64, torch.bfloat16
        cv = self._carve
        b = dict(N=N, Spad=Spad)
        # teacher targets (padded tails are constants: 1/Vd, 0, 0 -- eagle3/model.py:445-484)
        # (the [B, Spad, Vd] fp32 soft targets are carved on demand (_target_p): the usual path never materialises them -- the fused
        # CE re-forms them from the teacher's stored draft logits (teacher_zd) and the per-row (max, 1 / sum-exp) below)
        b["zmd"] = cv("zmd", B, Spad, dtype=f32)
        b["zinv"] = cv("zinv", B, Spad, dtype=f32)
        b["pod"] = cv("pod", B, Spad, dtype=f32)
        b["tsum"] = cv("tsum", B, Spad, dtype=f32)
        b["tids"] = cv("tids", B, Spad, dtype=i64)
        b["pm"] = cv("pm", B, Spad, dtype=i32)
        b["lm"] = cv("lm", B, Spad, dtype=i32)
        b["ids"] = cv("ids", B, Spad, dtype=i64)
        b["kvlen"] = cv("kvlen", B, dtype=i32)
        b["pos"] = cv("pos", N, dtype=i64)
        # per-step stash
        # Natural-layout stashes [T*N (+ zero pad rows to a multiple of 64), features]: row block k is TTT step k.  They are
        # the operands of the deferred weight-gradient GEMMs (sf_gemm_tn contracts over the token rows of dY and X as
        # stored, K = T*N), and the per-step kernels write straight into their slot -- no copies, no transposes.
        TNk = T * N
        Kp = (TNk + 63) // 64 * 64
        b["Kp"] = Kp
        slots = lambda t: [t[k * N:(k + 1) * N] for k in range(T)]
        h_all = cv("h_all", N + Kp, H)                  # h[0] | h[1..T] (the lm_head input when norm_output=False)
        b["h_s"] = h_all[N:]
        b["h"] = [h_all[k * N:(k + 1) * N] for k in range(T + 1)]
        b["h1"] = [cv(f"h1_{k}", N, H) for k in range(T)]
        b["qkv"] = [cv(f"qkv_{k}", N, self.QW) for k in range(T)]
        for nm, feat in zip(self._stash_names, (H, nh * hd, H, I, H, Vd, H, 2 * I, H, self.QW)):
            b[nm + "_s"] = cv(nm + "_s", Kp, feat)
            b[nm] = slots(b[nm + "_s"])
        b["lse"] = [cv(f"lse_{k}", B, nh, S, dtype=f32) for k in range(T)]
        b["gu"] = [cv(f"gu_{k}", N, 2 * I) for k in range(T)]
        b["dln"] = [cv(f"dln_{k}", N, H) for k in range(T)]
        for nm in ("rstd_h", "rstd_p", "rstd_n"):
            b[nm] = [cv(f"{nm}_{k}", N, dtype=f32) for k in range(T)]
        # embedding half of the QKV projection, hoisted out of the TTT loop: step k's token at position s is step 0's
        # token at s + k (eagle3/model.py:428-432), so input_layernorm(embed(.)) and its product with Wqkv[:, :H] are
        # computed once over the padded [B, S+T] positions and re-used by every step (sf_gemm_nt_rowadd)
        b["Np_real"] = B * Spad
        Np = (B * Spad + 31) // 32 * 32       # 2*Np is the K of a wgrad GEMM (64-aligned K -> 256-tile kernels); extra rows stay zero
        b["Np"] = Np
        b["en2"] = cv("en2", 2 * Np, H)                  # [en ; en]: pairs with [hi ; lo]
        b["en"] = b["en2"][:Np]
        b["rstd_e1"] = cv("rstd_e1", Np, dtype=f32)
        b["epart"] = cv("epart", Np, self.QW, dtype=f32)
        b["rstd_fc"] = [cv(f"rstd_fc_{i}", N, dtype=f32) for i in range(3)]
        # transient per-step work buffers
        N64 = (N + 63) // 64 * 64
        b["hs_s"] = cv("hs_s", N64, Ht3)      # fc input (fc_norm output, or a copy of the hidden states when N % 64 != 0)
        b["hsn"] = b["hs_s"][:N]
        b["dh0_s"] = cv("dh0_s", N64, H)
        b["rows"] = cv("rows", 3, N, dtype=f32)
        b["metrics"] = cv("metrics", T, 3, dtype=f32)
        b["msum"] = cv("msum", T, dtype=f32)             # LK: sum of the position mask per TTT step
        b["lk_logsum"] = cv("lk_logsum", T, dtype=f32)   # LK: sum_r m_r log(accept_r) per TTT step
        # backward work buffers
        b["dh_b"] = [cv("dh_b0", N, H), cv("dh_b1", N, H)]   # residual-stream gradient handed from step k to k-1 (ping-pong)
        b["dact"] = cv("dact", N, I)
        b["dpn"] = cv("dpn", N, H)
        # d(attention output) and delta of EVERY step are kept: the blocked diagonal-branch backward streams the later steps'
        # q / dO / lse / delta again when it finishes a block of branches (diag_plan); with padded heads: one buffer, the per-step form
        nat = self.hdp == hd
        b["do"] = [cv(f"do_{k}", N, nh * hd) for k in range(T)] if nat else [cv("do", N, nh * hd)] * T
        b["dxh"] = cv("dxh", N, H)
        # backward of the hoisted embedding half: the fp32 sum over the steps of dqkv re-aligned to token positions, as a
        # two-term bf16 expansion [hi ; lo] (one pass over the dqkv stash after the sweep), and the matching operand [en ; en]
        b["ds2"] = cv("ds2", 2 * Np, self.QW)              # [hi ; lo] stacked along the contraction
        b["ds_hi"], b["ds_lo"] = b["ds2"][:Np], b["ds2"][Np:]
        b["dE"] = cv("dE", Np, H)
        b["dhs"] = cv("dhs", N, Ht3) if c.fc_norm else None
        b["delta"] = [cv(f"delta_{k}", B, nh, S, dtype=f32) for k in range(T)] if nat else [cv("delta", B, nh, S, dtype=f32)] * T
        hdp = self.hdp
        b["dq_init"] = cv("dq_init", N, nh * hdp, dtype=f32)
        b["dk"] = [cv(f"dk_{k}", N, nkv * hdp, dtype=f32) for k in range(T)]
        b["dv"] = [cv(f"dv_{k}", N, nkv * hdp, dtype=f32) for k in range(T)]
        if hdp != hd:       # zero-padded heads (see __init__): padded copies of q / k / v / o per step, dO / dQ per sweep
            for nm, n in (("qp", nh), ("kp", nkv), ("vp", nkv), ("op", nh)):
                b[nm] = [cv(f"{nm}_{k}", N, n * hdp) for k in range(T)]
            b["dop"] = cv("dop", N, nh * hdp)
            b["dqp"] = cv("dqp", N, nh * hdp)
        if self.mrope:
            b["cos_rows"] = [cv(f"cos_rows_{k}", N, hd) for k in range(T)]
            b["sin_rows"] = [cv(f"sin_rows_{k}", N, hd) for k in range(T)]
        b["nws"] = cv("nws", 2 * ops.rmsnorm_bwd_workspace(N, max(H, c.target_hidden_size)), dtype=f32)   # (x 2: sf_rmsnorm_bwd2)
        b["nws_e"] = cv("nws_e", ops.rmsnorm_bwd_workspace(Np, H), dtype=f32)
        # per-block partials of the three norm weights every TTT step differentiates (final norm, hidden_norm, post-attention norm): the T
        # launches of a sweep write side by side and ONE column sum per weight follows the sweep (norm_colsum_batched)
        b["npart"] = {n: cv("npart_" + n.split(".")[-2], T * ops.rmsnorm_bwd_workspace(N, H), dtype=f32) for n in self._NORMS_PER_STEP}
        # fp32 partials for the 2-way split-K of weight-gradient GEMMs whose tile count fills the CUs badly (down, q|k|v)
        # (+ 4096 floats at the tail: pace-keeping counters of sf_gemm_tn)
        b["tn_ws"] = cv("tn_ws", 2 * max(H * I, self.QW * H) + 4096, dtype=f32)
        # split-K partials of the plain NT GEMMs when their grids are under-filled (few tokens x narrow outputs: bs 1 recipes); else unused
        b["inv"] = cv("inv", N, dtype=i32)                # loss-row compaction: token row -> compact row (or -1)
        b["arange"] = cv("arange", N, dtype=i32)
        b["nt_ws"] = (cv("nt_ws", 4 * 128 * 65536, dtype=f32) if ((N + 255) // 256) * ((H + 255) // 256) <= 128 else None)
        nws = ops.attn_bwd_dkv_workspace_floats(B, S, nh, nkv, hdp)     # head-split partials (small B * nkv only; else 0)
        b["dkv_ws"] = cv("dkv_ws", nws, dtype=f32) if nws else None
        return b

    def _target_p(self, b, B: int, S: int) -> torch.Tensor:
        """the materialised soft targets [B, S+T, Vd] fp32 (constant 1/Vd tails, eagle3/model.py:445-484) for the paths that need them:
        online ``target_logits``, LK objectives, vocabulary mappings the permuted head cannot take, shapes too small for the reduced GEMM"""
        Vd = self.cfg.draft_vocab_size
        tp = self._carve("tp", B, S + self.T, Vd, dtype=torch.float32, invalidate=False)     # (never cached in a view dict)
        if self._tp_state != (B, S, tp.data_ptr()):
            tp[:, S:].fill_(1.0 / Vd)
            self._tp_state = (B, S, tp.data_ptr())
        self._soft = ("tp", tp)
        return tp

    def soft_targets(self, B: int, S: int) -> torch.Tensor:
        """target_p [B, S, Vd] fp32 of the last forward, materialised from whichever form it used (inspection / tests)"""
        kind, v = self._soft
        if kind == "tp":
            return v[:, :S]
        zd, zmd, zinv = v
        Vd = self.cfg.draft_vocab_size
        return torch.exp(zd[:B * S, :Vd].float().view(B, S, Vd) - zmd[:, :S, None]) * zinv[:, :S, None]

    def _pad_heads(self, src: torch.Tensor, n: int, dst: torch.Tensor) -> None:
        """[N, n*hd] (any row stride) -> the first hd columns of each head of dst [N, n*hdp]; the pad columns stay zero"""
        N, hd = src.shape[0], self.cfg.head_dim
        dst.view(N, n, self.hdp)[:, :, :hd].copy_(src.unflatten(1, (n, hd)))

    def _unpad_heads(self, src: torch.Tensor, n: int, dst: torch.Tensor) -> None:
        """inverse of _pad_heads (also fp32 -> bf16 for the K / V gradient accumulators)"""
        N, hd = src.shape[0], self.cfg.head_dim
        dst.unflatten(1, (n, hd)).copy_(src.view(N, n, self.hdp)[:, :, :hd])

    def _permuted_teacher_head(self, w: torch.Tensor):
        """(perm, W[perm], ascending?): the frozen teacher head with its ROWS reordered so that the logits of the draft sub-vocabulary come
        out of the head GEMM first and contiguous (perm[j] = j + d2t[j], then every other vocabulary entry in ascending order).
        Built once per head weight (a 1 GB copy at Llama-3 dims: nothing against 288 GB) -- the target model is frozen
        (target_head.py:60-66).  (None, w, False) when the vocabulary mapping is not a clean injection (then the natural layout is used)."""
        key = (w.data_ptr(), tuple(w.shape), w._version, self.model.d2t._version, self.model.t2d._version)
        if getattr(self, "_thead_key", None) != key:
            Vt, Vd = w.shape[0], self.cfg.draft_vocab_size
            cols = torch.arange(Vd, device=self.dev) + self.model.d2t.to(self.dev)
            t2d = self.model.t2d.to(self.dev).bool()
            ok = (Vd % 8 == 0 and Vt < (1 << 24) and t2d.numel() == Vt and int(cols.min()) >= 0 and int(cols.max()) < Vt
                  and int(t2d.sum()) == Vd and bool(t2d[cols].all()) and int(torch.unique(cols).numel()) == Vd)
            if ok:
                rest = torch.nonzero(~t2d).flatten()
                perm = torch.cat([cols, rest])
                ordered = Vd < 2 or bool((cols[1:] > cols[:-1]).all())      # (`rest` is ascending by construction)
                self._thead = (perm.to(torch.int32).contiguous(), w.index_select(0, perm).contiguous(), ordered)
            else:
                self._thead = (None, w, False)
            self._thead_key = key
        return self._thead

    def _refresh_weight_transposes(self):
        """W^T images for the dgrad GEMMs (NT form); rebuilt only after an optimizer step."""
        if self._wt_version == self.weights_version:
            return
        c, f = self.cfg, self.flat
        H, I, hd, nh = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads
        if not hasattr(self, "wlmT"):
            self.wlmT = self._e(H, c.draft_vocab_size)
            self.wguT = self._e(H, 2 * I)
            self.wdT = self._e(I, H)
            self.wqkvT = self._e(2 * H, self.QW)
            self.woT = self._e(nh * hd, H)
            self.wfcT = self._e(3 * c.target_hidden_size, H) if c.fc_norm else None
        ops.transpose2d(f.view("lm_head.weight"), self.wlmT)
        ops.transpose2d(self.w_gu, self.wguT)
        ops.transpose2d(f.view("midlayer.mlp.down_proj.weight"), self.wdT)
        ops.transpose2d(self.w_qkv, self.wqkvT)
        ops.transpose2d(f.view("midlayer.self_attn.o_proj.weight"), self.woT)
        if self.wfcT is not None:
            ops.transpose2d(f.view("fc.weight"), self.wfcT)
        self._wt_version = self.weights_version

    FLAG_MESSAGES = (
        "loss_counts passed to the forward do not match the loss mask (loss_counts[k] must be the number of rows with "
        "loss_mask[b, s + k] != 0): the compacted lm_head rows of this step are wrong",
        "position_ids outside the range the reference accepts (plain ids index cos[:S + k] at ids + k, llama3_eagle.py:303-311 / "
        "134-139: 0 <= id < seq_length; mrope ids index the precomputed table: pass them as a CPU tensor so the table can grow, "
        "or raise max_position_embeddings)",
    )

    def check_flags(self, values) -> None:
        """raise for the device-side input checks of the last forward (``values`` = a host copy of ``_flags``)"""
        for v, msg in zip(values, self.FLAG_MESSAGES):
            if v != 0.0:
                raise RuntimeError(msg)

    def _rope_steps(self, S: int):
        """(cos, sin) per TTT step.  Step k rotates with the table ``rotary_emb(x, seq_len=S + k)`` returns (llama3_eagle.py:733); when
        that exceeds the cached length the reference rebuilds the cache for exactly S + k positions (303-306) and keeps it.  For every
        variant but dynamic NTK the rebuilt rows equal the old ones, so the table just grows (in strides: real data brings a new S
        almost every step); dynamic NTK derives its base from the rebuilt length (362-371), so each growing step gets its own table --
        and, as in the reference, the last one stays cached for the following forwards."""
        T, c = self.T, self.cfg
        if S + T - 1 <= self._rope_len:
            return [(self.cos, self.sin)] * T
        rs = c.rope_scaling or {}
        if rs.get("rope_type", rs.get("type")) != "dynamic":
            self._grow_rope(S + T - 1)
            return [(self.cos, self.sin)] * T
        tabs = []
        for k in range(T):
            if S + k > self._rope_len:
                cos, sin = rope_tables(c, torch.bfloat16, n_pos=S + k)
                self.cos, self.sin, self._rope_len = cos.to(self.dev), sin.to(self.dev), S + k
            tabs.append((self.cos, self.sin))
        return tabs

    def _grow_rope(self, rows: int) -> None:
        """plain (non-dynamic) tables for at least ``rows`` positions: the rows already there keep their values"""
        if rows > self.cos.shape[0]:
            n = max(rows, self.cos.shape[0] + 1024, int(1.25 * self.cos.shape[0]))
            cos, sin = rope_tables(self.cfg, torch.bfloat16, n_pos=n)
            self.cos, self.sin = cos.to(self.dev), sin.to(self.dev)
        self._rope_len = max(self._rope_len, rows)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def _teacher_compact(self, b, th, Nm, zd, Vz, perm, head, part, cnt_probe, B, S, Spad) -> bool:
        """The teacher of a sparse loss mask (loss-row compaction, see __init__): TTT step k scores row (b, s) against position s + k only
        where loss_mask[b, s + k] != 0, so only the Nm = loss_counts[0] positions with a loss mask need target logits.  Their hidden rows
        are gathered, go through the permuted head GEMM + reduction in chunks of ``teacher_rows``, and the stored draft logits / per-row
        scalars are scattered back to their natural positions (what the fused CE addresses); every other position gets position mask 0 and
        target id 0 (nothing reads its soft target; ``last_artifacts`` are meaningful where the loss mask is set).  -> False when a chunk
        would not take the reduced GEMM form (the caller runs the dense teacher)."""
        Vd, Vt, Ht = self.cfg.draft_vocab_size, head.shape[0], th.shape[1]
        # (the head GEMM's per-row partials were carved for the dense teacher's chunk; the gathered form keeps chunks of <= 4096 rows: its scratch --
        #  [chunk, roundup(Vd, 256)] logits -- is sized by the largest count seen, and a long run on sparse masks should reach its floor early)
        cap = min(self.teacher_compact_rows, self.teacher_rows, part.shape[0])
        nch = max(1, -(-Nm // cap))
        m = -(-Nm // nch)
        sizes = [min(m, Nm - i * m) for i in range(nch)] if Nm else []
        if Nm and not all(x > 0 and ops.gemm_nt_teacher_reduces(x, Vt, Ht, Vd) for x in sizes):
            return False
        for nm in ("pm", "tids"):
            b[nm].zero_()
        if Nm == 0:     # (the count is checked like every other: no row may carry a loss)
            probe = torch.nonzero_static(b["lm"][:, :S].reshape(-1), size=1, fill_value=-1).view(-1)
            cnt_probe.append(torch.cat([probe.new_zeros(1), probe]))
            return True
        idx = torch.nonzero_static(b["lm"][:, :S].reshape(-1), size=Nm + 1, fill_value=-1).view(-1)
        cnt_probe.append(idx[Nm - 1:Nm + 1])
        rows = idx[:Nm].clamp_min(0)
        pidx = rows + torch.div(rows, S, rounding_mode="floor") * (Spad - S)         # index into the padded [B, Spad] arrays
        thc = self._carve("teacher_thc", Nm, Ht, invalidate=False)
        torch.index_select(th, 0, rows, out=thc)
        zc = self._carve("teacher_zc", m, Vz, invalidate=False)
        f32 = {nm: self._carve("teacher_c_" + nm, 1, Nm, dtype=torch.float32, invalidate=False) for nm in ("pod", "tsum", "zmd", "zinv")}
        ids_c = self._carve("teacher_c_tids", 1, Nm, dtype=torch.int64, invalidate=False)
        pm_c = self._carve("teacher_c_pm", 1, Nm, dtype=torch.int32, invalidate=False)
        ones = self._carve("teacher_c_ones", 1, Nm, dtype=torch.int32, invalidate=False)
        ones.fill_(1)
        lo = 0
        for x in sizes:
            sl = slice(lo, lo + x)
            vz, nparts = ops.gemm_nt_teacher(thc[sl], head, zc[:x], part, Vd=Vd)
            assert vz == Vz and nparts > 0
            ops.teacher_reduce_perm(zc[:x], Vt=Vt, Vd=Vd, perm=perm, t2d_u8=self._t2d_u8, part=part, nparts=nparts, target_p_pad=None,
                                    loss_mask_pad=ones[:, sl], S=x, Spad=x, pod_scale_pad=f32["pod"][:, sl], tsum_pad=f32["tsum"][:, sl],
                                    ids_pad=ids_c[:, sl], pos_mask_pad=pm_c[:, sl], zmd_pad=f32["zmd"][:, sl], zinv_pad=f32["zinv"][:, sl])
            zd.index_copy_(0, rows[sl], zc[:x])
            lo += x
        for nm in ("pod", "tsum", "zmd", "zinv"):
            b[nm].view(-1).index_copy_(0, pidx, f32[nm].view(-1))
        b["tids"].view(-1).index_copy_(0, pidx, ids_c.view(-1))
        b["pm"].view(-1).index_copy_(0, pidx, pm_c.view(-1))
        return True

    def forward(self, *, input_ids, attention_mask, loss_mask, hidden_states, target_hidden=None,
                target_head_weight=None, target_logits=None, position_ids=None, train: bool = True, loss_counts=None,
                position_span=None):
        """One micro-step forward.  ``input_ids`` / ``target_*`` are already shifted by
        ``TargetHead.preprocess`` (target_head.py:103-108); ``loss_mask`` is [B,S] or [B,S,1].
        ``position_span`` = (min, max) of ``position_ids`` as host integers when the caller still had them in host memory.
        Returns the metric dict of ``Eagle3TrainStrategy.forward_loss`` (lists of 0-dim tensors)."""
        c, T, f = self.cfg, self.T, self.flat
        for n in (f.names[0], f.names[-1]):      # the draft's parameters must still BE the flat buffer (a .to() / .float() /
            if f.params[n].data_ptr() != f.data[f.slices[n][0]:].data_ptr():   # .half() on the draft module re-creates them)
                raise RuntimeError("the draft model's parameters were moved or cast after the HIP engine adopted them; training "
                                   "would update stale buffers -- move / cast the model before its first forward")
        B, S = input_ids.shape
        b = self._buffers(B, S)
        N, Spad = b["N"], b["Spad"]
        H, I, hd, nh, nkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        Vd, Ht = c.draft_vocab_size, c.target_hidden_size
        eps, scale = c.rms_norm_eps, 1.0 / math.sqrt(hd)
        rope = self._rope_steps(S)          # per-step (cos, sin); grows the table like the reference's rotary cache
        if self._flags_set:
            self._flags.zero_()
            self._flags_set = False
        loss_mask = loss_mask.reshape(B, S)
        if attention_mask is None:
            attention_mask = torch.ones(B, S, dtype=torch.int64, device=self.dev)
        if not attention_mask.is_cuda and attention_mask.numel():
            am = attention_mask.bool()
            if bool((am[:, 1:] & ~am[:, :-1]).any()):
                raise ValueError("attention_mask must be right-padded (1...1 0...0): the collator's contract, data/utils.py:122-196")
        self._refresh_weight_transposes()
        # ---- host-side plumbing into the padded, typed buffers
        b["ids"][:, :S].copy_(input_ids)
        b["lm"][:, :S].copy_(loss_mask)
        b["kvlen"].copy_(attention_mask.to(self.dev).sum(dim=1))
        if self.mrope:
            if position_ids is None or position_ids.dim() != 3 or tuple(position_ids.shape) != (3, B, S):
                raise ValueError("rope_type 'mrope' needs position_ids of shape [3, batch, seq_length] (eagle3/model.py:228-242)")
            # the reference computes the angles analytically from the ids (llama3_eagle.py:719-733, no limit); this engine gathers
            # rows of the precomputed plain table (exact for any length): with the ids' span known on the HOST the table grows to
            # it; ids that arrive as a device tensor are checked by a device flag (read back at the end of the backward sweep --
            # no host sync) and must fit the table as it is.  Ids past it must fail, never rotate by a clamped angle.
            if position_span is None and not position_ids.is_cuda:
                position_span = (int(position_ids.min()), int(position_ids.max()))
            if position_span is not None:
                lo, hi = position_span
                if lo < 0:
                    raise ValueError(f"mrope position_ids must be >= 0 (got {lo})")
                # a corrupt sample (an int64 garbage id) must fail here, not while building a [hi, head_dim] table on the host
                cap = max(self.max_rope_positions, 64 * (c.max_position_embeddings + 20))
                if hi + T > cap:
                    raise ValueError(f"mrope position_ids span [{lo}, {hi}] is beyond {cap} positions (64 x max_position_embeddings; "
                                     "raise Eagle3Engine.max_rope_positions if the data really is that long)")
                self._grow_rope(hi + T)
                rope = [(self.cos, self.sin)] * T
            pos3 = position_ids.to(self.dev).long().reshape(3, N)
            if position_span is None:
                self._flags[1] = ((pos3 < 0) | (pos3 + (T - 1) >= self.cos.shape[0])).any()
                self._flags_set = True
            cols = torch.arange(hd, device=self.dev)
            for k in range(T):   # step k rotates at position + k on every axis (llama3_eagle.py:719-733)
                idx = (pos3 + k)[self._mrope_axis].t()      # [N, hd]
                if position_span is None:                    # (device ids: a flagged batch must not fault in the gather either)
                    idx = idx.clamp(0, self.cos.shape[0] - 1)
                b["cos_rows"][k].copy_(self.cos[idx, cols])
                b["sin_rows"][k].copy_(self.sin[idx, cols])
            b["pos"].copy_(torch.arange(N, device=self.dev))                                  # row r reads table row r
            self._pos_default = None
        elif position_ids is None:
            if self._pos_default != (B, S):      # 0..S-1 per sample: laid down once per shape, not every step
                b["pos"].copy_(torch.arange(S, device=self.dev).repeat(B))
                self._pos_default = (B, S)
        else:
            if position_ids.dim() != 2:
                raise ValueError("position_ids must be [batch, seq_length] (three-axis ids need rope_type 'mrope')")
            # the reference indexes cos[:S + k][position_ids + k] (llama3_eagle.py:303-311, 134-139): an id >= S is an IndexError
            # there at every step (negative ids wrap around python-style; refused here).  Host tensors / a host-known span are
            # checked now, device tensors by a device flag read back at the end of the backward sweep.
            if position_span is None and not position_ids.is_cuda and position_ids.numel():
                position_span = (int(position_ids.min()), int(position_ids.max()))
            if position_span is not None:
                if position_span[0] < 0 or position_span[1] >= S:
                    raise IndexError(f"position_ids span [{position_span[0]}, {position_span[1]}] but must lie in [0, {S}): the reference "
                                     "indexes its RoPE cache cos[:seq_length + k] at position_ids + k (llama3_eagle.py:303-311, 134-139)")
            b["pos"].copy_(position_ids.reshape(-1))
            if position_span is None:
                self._flags[1] = ((b["pos"] < 0) | (b["pos"] >= S)).any()
                self._flags_set = True
            self._pos_default = None
        if self._t2d_u8 is None or self._t2d_u8.device != self.dev:
            self._t2d_u8 = self.model.t2d.to(self.dev).to(torch.uint8).contiguous()
            self._d2t = self.model.d2t.to(self.dev).contiguous()
        hs = hidden_states.reshape(N, 3 * Ht)
        self._last_hs = hs

        # ---- loss-row compaction (see __init__): host-known row counts per step, or the dense form
        cnt = None
        if (self.compact_loss_rows and train and loss_counts is not None and self.lk_loss_type is None and c.norm_output
                and len(loss_counts) >= T):
            cnt = [int(x) for x in loss_counts[:T]]
            if not all(0 <= x <= N for x in cnt) or sum(cnt) > 0.9 * T * N:
                cnt = None
        cum = [0]
        if cnt is not None:
            for x in cnt:
                cum.append(cum[-1] + x)
        cnt_probe = []
        self._teacher_compacted = False

        # ---- teacher: target logits (chunked GEMM, bf16 like TargetHead.forward) -> soft targets
        zt = None        # (zd, zmd, zinv) when the soft targets stay un-materialised (see below); else b["tp"] holds them
        if target_logits is not None:
            z = target_logits.reshape(N, -1)
            ops.teacher_reduce(z, Vd=Vd, d2t=self._d2t, t2d_u8=self._t2d_u8, loss_mask_pad=b["lm"], S=S, Spad=Spad,
                               target_p_pad=self._target_p(b, B, S), pod_scale_pad=b["pod"], tsum_pad=b["tsum"], ids_pad=b["tids"],
                               pos_mask_pad=b["pm"])
        else:
            th = target_hidden.reshape(B, S, Ht)
            Vt = target_head_weight.shape[0]
            cb = max(1, self.teacher_rows // S)
            perm, head, ordered = self._permuted_teacher_head(target_head_weight)
            # per-block partials of the columns the head GEMM reduces instead of storing (only when ties inside a block resolve
            # to the lowest ORIGINAL index by column order alone, i.e. the mapping is ascending)
            prow = min(cb, B) * S
            if ordered:      # (capacity for the reserved shape at once: a long run's arena stops growing at its first step, not at its largest batch)
                self._carve("teacher_part", max(prow, min(self.teacher_rows, getattr(self, "_reserved_rows", 0))), (Vt - Vd + 127) // 128, 4, dtype=torch.float32, invalidate=False)
            part = self._carve("teacher_part", prow, (Vt - Vd + 127) // 128, 4, dtype=torch.float32, invalidate=False) if ordered else None
            chunks = [(b0, min(cb, B - b0)) for b0 in range(0, B, cb)]
            # The usual case -- every chunk takes the reduced head GEMM, no LK objective: the GEMM writes the draft logits of its rows
            # straight into a persistent [N, roundup(Vd, 256)] bf16 array, the reduce kernel adds (max, 1 / sum-exp) per row, and the
            # fused CE of each TTT step re-forms target_p from them.  [B, S, Vd] fp32 (2.1 GB at the headline shape, written once and
            # read by 7 steps) does not exist; the scratch for a chunk's full-vocabulary logits neither.
            if (part is not None and self.lk_loss_type is None and not self.materialise_soft_targets
                    and all(ops.gemm_nt_teacher_reduces(nb * S, Vt, Ht, Vd) for _, nb in chunks)):
                Vz = (Vd + 255) // 256 * 256
                zd = self._carve("teacher_zd", N, Vz, invalidate=False)
                zt = (zd, b["zmd"], b["zinv"])
                self._soft = ("zt", zt)
                if cnt is not None and self._teacher_compact(b, th.reshape(N, Ht), cnt[0], zd, Vz, perm, head, part, cnt_probe, B, S, Spad):
                    chunks = []         # done: only the positions with a loss mask went through the head
                    self._teacher_compacted = True
            else:
                zbuf = self._carve("teacher_z", min(cb, B) * S, Vt, invalidate=False)   # scratch: no cached view aliases it
            for b0, nb in chunks:
                out = dict(loss_mask_pad=b["lm"][b0:b0 + nb], S=S, Spad=Spad, pod_scale_pad=b["pod"][b0:b0 + nb],
                           tsum_pad=b["tsum"][b0:b0 + nb], ids_pad=b["tids"][b0:b0 + nb], pos_mask_pad=b["pm"][b0:b0 + nb])
                x = th[b0:b0 + nb].reshape(nb * S, Ht)
                if zt is not None:
                    z = zd[b0 * S:(b0 + nb) * S]
                    vz, nparts = ops.gemm_nt_teacher(x, head, z, part, Vd=Vd)
                    assert vz == Vz and nparts > 0
                    ops.teacher_reduce_perm(z, Vt=Vt, Vd=Vd, perm=perm, t2d_u8=self._t2d_u8, part=part, nparts=nparts, target_p_pad=None,
                                            zmd_pad=b["zmd"][b0:b0 + nb], zinv_pad=b["zinv"][b0:b0 + nb], **out)
                    continue
                z = zbuf[: nb * S]
                out["target_p_pad"] = self._target_p(b, B, S)[b0:b0 + nb]
                if perm is not None:     # columns of z: the draft sub-vocabulary first, in draft order
                    vz, nparts = ops.gemm_nt_teacher(x, head, z, part, Vd=Vd)
                    ops.teacher_reduce_perm(z[:, :vz], Vt=Vt, Vd=Vd, perm=perm, t2d_u8=self._t2d_u8, part=part, nparts=nparts, **out)
                else:
                    ops.gemm_nt(x, head, z)
                    ops.teacher_reduce(z, Vd=Vd, d2t=self._d2t, t2d_u8=self._t2d_u8, **out)

        # ---- fc (optionally 3x RMSNorm first): llama3_eagle.py:1762-1770
        if c.fc_norm:
            for i in range(3):
                ops.rmsnorm_fwd(hs[:, i * Ht:(i + 1) * Ht], f.view(f"fc_norm.{i}.weight"), eps,
                                b["hsn"][:, i * Ht:(i + 1) * Ht], b["rstd_fc"][i])
            fc_in = b["hsn"]
            self._fc_x = b["hs_s"]
        elif N % 64 == 0:
            fc_in = self._fc_x = hs                    # already a valid sf_gemm_tn operand (K = N)
        else:
            b["hsn"].copy_(hs)                          # N % 64 != 0 (the usual case on ragged real data: the collator pads to
            # the longest sample, whatever it is): zero-padded copy of the fc input for the K = N weight-gradient GEMM
            fc_in, self._fc_x = b["hsn"], b["hs_s"]
        ops.gemm_nt(fc_in, f.view("fc.weight"), b["h"][0], workspace=b["nt_ws"])

        kcol, vcol = slice(nh * hd, (nh + nkv) * hd), slice((nh + nkv) * hd, self.QW)
        lk = self.lk_loss_type
        if lk is not None:
            for k in range(T):
                b["msum"][k] = b["pm"][:, k:k + S].sum()
        # input_layernorm(embed(ids)) . Wqkv[:, :H]^T for every padded position, once (fp32, joins each step's
        # accumulator before the rounding)
        Np = b["Np"]
        ops.rmsnorm_fwd(self.model.embed_tokens.weight.data, f.view("midlayer.input_layernorm.weight"), eps, b["en"],
                        b["rstd_e1"], ids_pad=b["ids"], S=Spad, Spad=Spad, off=0, rows=b["Np_real"])
        ops.gemm_nt(b["en"], self.w_qkv[:, :H], b["epart"], workspace=b["nt_ws"])
        if train:
            b["en2"][Np:].copy_(b["en"])
        self._lm_compact_K = None
        for k in range(T):
            hn, qkv, pn, act, logits = b["hn"][k], b["qkv"][k], b["pn"][k], b["act"][k], b["logits"][k]
            # q/k/v of cat(input_layernorm(embed(ids<<k)), hidden_norm(h_k))   (llama3_eagle.py:1625-1630)
            if k == 0 or not c.norm_output or cnt is not None:     # (else: the final norm of step k - 1 wrote hn[k] in the same pass over h[k])
                ops.rmsnorm_fwd(b["h"][k], f.view("midlayer.hidden_norm.weight"), eps, hn, b["rstd_h"][k])
            ops.gemm_nt_rowadd(hn, self.w_qkv[:, H:], qkv, b["epart"], S=S, Spad=Spad, off=k, workspace=b["nt_ws"])
            if self.mrope:
                ops.rope_(qkv, nh + nkv, hd, b["cos_rows"][k], b["sin_rows"][k], b["pos"], 0)
            else:
                ops.rope_(qkv, nh + nkv, hd, rope[k][0], rope[k][1], b["pos"], k)
            if self.hdp == hd:
                ops.attn_fwd(qkv[:, :nh * hd], b["qkv"][0][:, kcol], b["qkv"][0][:, vcol], [b["qkv"][i][:, kcol] for i in range(1, k + 1)],
                             [b["qkv"][i][:, vcol] for i in range(1, k + 1)], b["kvlen"], b["o"][k], b["lse"][k],
                             B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
            else:
                self._pad_heads(qkv[:, :nh * hd], nh, b["qp"][k])
                self._pad_heads(qkv[:, kcol], nkv, b["kp"][k])
                self._pad_heads(qkv[:, vcol], nkv, b["vp"][k])
                ops.attn_fwd(b["qp"][k], b["kp"][0], b["vp"][0], b["kp"][1:k + 1], b["vp"][1:k + 1], b["kvlen"], b["op"][k],
                             b["lse"][k], B=B, S=S, nh=nh, nkv=nkv, hd=self.hdp, scale=scale)
                self._unpad_heads(b["op"][k], nh, b["o"][k])
            ops.gemm_nt(b["o"][k], f.view("midlayer.self_attn.o_proj.weight"), b["h1"][k], residual=b["h"][k], workspace=b["nt_ws"])
            ops.rmsnorm_fwd(b["h1"][k], f.view("midlayer.post_attention_layernorm.weight"), eps, pn, b["rstd_p"][k])
            ops.gemm_nt_swiglu_fwd(pn, self.w_gu, b["gu"][k], act)      # gate|up projection, SwiGLU in its epilogue
            ops.gemm_nt(act, f.view("midlayer.mlp.down_proj.weight"), b["h"][k + 1], residual=b["h1"][k], workspace=b["nt_ws"])
            if cnt is not None:
                # compact form: the rows of this step that carry a loss mask (lm_pad[b, s + k] != 0), in token order
                Nc, lo = cnt[k], cum[k]
                if Nc == 0:     # (checked like every other count: no row of this step may carry a loss)
                    probe = torch.nonzero_static(b["lm"][:, k:k + S].reshape(-1), size=1, fill_value=-1).view(-1)
                    cnt_probe.append(torch.cat([probe.new_zeros(1), probe]))
                    b["metrics"][k].zero_()
                    if train:
                        b["dln"][k].zero_()
                        b["rstd_n"][k].zero_()
                    continue
                # (one slot more than the count: entry Nc - 1 must be a row and entry Nc the fill value -- checked on the device,
                #  read back with the upstream gradient at the end of the backward sweep; a wrong count never indexes out of range)
                idx = torch.nonzero_static(b["lm"][:, k:k + S].reshape(-1), size=Nc + 1, fill_value=-1).view(-1)
                cnt_probe.append(idx[Nc - 1:Nc + 1])
                rows_k = idx[:Nc].clamp_min(0)
                lnc, logc = b["ln_s"][lo:lo + Nc], b["logits_s"][lo:lo + Nc]
                rstd_c = b["rows"].view(-1)[3 * N - Nc:3 * N]       # (scratch: the tail of the row-statistics buffer, free until the CE)
                ops.rmsnorm_fwd(b["h"][k + 1], f.view("norm.weight"), eps, lnc, rstd_c, ids_pad=rows_k, S=Nc, Spad=Nc, off=0, rows=Nc)
                if train:       # the norm backward runs over ALL rows: zero rstd where dy is zero keeps dx = 0 finite
                    b["rstd_n"][k].zero_()
                    b["rstd_n"][k].index_copy_(0, rows_k, rstd_c)
                ops.gemm_nt(lnc, f.view("lm_head.weight"), logc)
                rows_c = b["rows"].view(-1)[:3 * Nc].view(3, Nc)
                ce_kw = dict(S=S, Spad=Spad, off=k, pos_mask_pad=b["pm"], loss_mask_pad=b["lm"], tgt_ids_pad=b["tids"], pod_scale_pad=b["pod"],
                             tsum_pad=b["tsum"], d2t=self._d2t, grad_scale=(self.decay ** k) / N, write_grad=train, row_loss=rows_c[0],
                             row_correct=rows_c[1], row_accept=rows_c[2], row_map=rows_k)
                if zt is not None:
                    ops.ce_fused_zt(logc, zt[0], zt[1], zt[2], **ce_kw)
                else:
                    ops.ce_fused(logc, self._soft[1], **ce_kw)
                ops.reduce_sum(rows_c, Nc, 3, b["metrics"][k], 1.0)
                if train:
                    dlnc = b["dxh"][:Nc]            # (a backward work buffer, idle during the forward)
                    ops.gemm_nt(logc, self.wlmT, dlnc, workspace=b["nt_ws"])
                    # back to token rows, zeros where no loss was taken: one pass (sf_rows_expand) instead of a fill + an indexed copy
                    inv = b["inv"]
                    inv.fill_(-1)
                    inv.index_copy_(0, rows_k, b["arange"][:Nc])
                    ops.rows_expand(dlnc, inv, b["dln"][k])
                continue
            if c.norm_output:   # compute_logits (llama3_eagle.py:1772-1777)
                ln = b["ln"][k]
                if k + 1 < T:       # ... and the next step's hidden_norm of the same h[k+1]: one pass, one row statistic
                    ops.rmsnorm_fwd2(b["h"][k + 1], f.view("norm.weight"), ln, b["rstd_n"][k], f.view("midlayer.hidden_norm.weight"),
                                     b["hn"][k + 1], b["rstd_h"][k + 1], eps)
                else:
                    ops.rmsnorm_fwd(b["h"][k + 1], f.view("norm.weight"), eps, ln, b["rstd_n"][k])
            else:
                ln = b["h"][k + 1]
            ops.gemm_nt(ln, f.view("lm_head.weight"), logits)
            # loss + d(logits) in place + accuracy + acceptance; ploss_k = mean over ALL B*S rows
            ce_kw = dict(S=S, Spad=Spad, off=k, pos_mask_pad=b["pm"], loss_mask_pad=b["lm"], tgt_ids_pad=b["tids"], pod_scale_pad=b["pod"],
                         tsum_pad=b["tsum"], d2t=self._d2t, grad_scale=(self.decay ** k) / N, write_grad=train and lk is None,
                         row_loss=b["rows"][0], row_correct=b["rows"][1], row_accept=b["rows"][2])
            if zt is not None:
                ops.ce_fused_zt(logits, zt[0], zt[1], zt[2], **ce_kw)
            else:
                ops.ce_fused(logits, self._soft[1], **ce_kw)
            ops.reduce_sum(b["rows"], N, 3, b["metrics"][k], 1.0)
            if lk is not None:
                # LK objectives (core/lk_loss.py:83-99): the step loss needs the masked MEANS over all rows, so the
                # gradient is a second pass once the row sums are reduced (device scalars, no host sync)
                ra = b["rows"][2]
                b["lk_logsum"][k] = torch.where(ra > 0, torch.log(ra), torch.zeros_like(ra)).sum()
                if train:
                    ops.ce_lk_grad(logits, self._soft[1], S=S, Spad=Spad, off=k, pos_mask_pad=b["pm"],
                                   pod_scale_pad=b["pod"], tsum_pad=b["tsum"], lk_loss_type=lk, kl_scale=self.kl_scale,
                                   kl_decay=self.kl_decay, step_scale=self.decay ** k, kl_row_scale=1.0 / N,
                                   accept_sum=b["metrics"][k][2:3], mask_sum=b["msum"][k:k + 1])
            if train:
                ops.gemm_nt(logits, self.wlmT, b["dln"][k], workspace=b["nt_ws"])        # lm_head dgrad, taken now

        if cnt_probe:
            pr = torch.stack(cnt_probe)
            self._flags[0] = ((pr[:, 0] < 0) | (pr[:, 1] >= 0)).any()
            self._flags_set = True
        self._rope_fwd = rope if train else None
        if not train and self._flags_set:       # no backward will read the flags: an eval forward pays the read-back itself
            self.check_flags(self._flags.tolist())
        if cnt is not None and train:
            # the compact lm_head stash: rows [0, cum[T]) of logits_s / ln_s, zero rows up to the next multiple of 64 (K of sf_gemm_tn)
            Kc = max(64, (cum[T] + 63) // 64 * 64)
            b["ln_s"][cum[T]:Kc].zero_()
            b["logits_s"][cum[T]:Kc].zero_()
            self._lm_compact_K = Kc
        self._fwd_state = (B, S) if train else None
        # ---- metrics (tiny integer-mask sums; eagle3/model.py:161-190, core/lk_loss.py:43-80)
        met = b["metrics"]
        out = dict(plosses=[], acces=[], acceptance_rates=[], acc_corrects=[], acc_denoms=[], metric_losses=[],
                   metric_loss_denoms=[])
        if lk is None:          # one launch for all 7 x T scalars (the lists are 0-dim views of a fresh [T, 8] tensor)
            ms = torch.empty(T, 8, dtype=torch.float32, device=self.dev)
            ops.eagle3_metrics(met, b["lm"], b["pm"], ms, B=B, S=S, Spad=Spad, T=T)
            for k in range(T):
                r = ms[k]
                out["plosses"].append(r[0])
                out["acc_corrects"].append(r[1])
                out["acc_denoms"].append(r[2])
                out["acces"].append(r[3])
                out["acceptance_rates"].append(r[4])
                out["metric_losses"].append(r[7])
                out["metric_loss_denoms"].append(r[6])
            out["target_token_ids"] = b["tids"][:, :S]
            out["position_mask"] = b["pm"][:, :S]
            return out
        lm_f, pm_f = b["lm"].float(), b["pm"].float()
        for k in range(T):
            denom = lm_f[:, k:k + S].sum().clamp_min(1e-6)
            pden = pm_f[:, k:k + S].sum().clamp_min(1e-8)
            ploss = met[k, 0] / N
            if lk == "alpha":      # -masked mean of log acceptance (core/lk_loss.py:92-93)
                ploss = -b["lk_logsum"][k] / pden
            elif lk == "lambda":   # w*KL + (1-w)*(1-acceptance), w detached (core/lk_loss.py:94-97)
                acc_rate = met[k, 2] / pden
                w = self.kl_scale * torch.exp(-self.kl_decay * acc_rate)
                ploss = w * ploss + (1 - w) * (1 - acc_rate)
            out["plosses"].append(ploss)
            out["acc_corrects"].append(met[k, 1].clone())
            out["acc_denoms"].append(denom)
            out["acces"].append(met[k, 1] / denom)
            out["acceptance_rates"].append(met[k, 2] / pden)
            out["metric_losses"].append(ploss.clone())
            out["metric_loss_denoms"].append(torch.full((), float(N), device=self.dev))   # device-side fill: torch.tensor(x,
            # device=...) is a pageable host-to-device copy, which blocks the host until the whole forward has run
        out["target_token_ids"] = b["tids"][:, :S]
        out["position_mask"] = b["pm"][:, :S]
        return out

    # ----------------------------------------------------------------- backward
    @torch.no_grad()
    def backward(self, g=1.0):
        """Backward sweep of the last ``forward(train=True)``; accumulates g * dLoss/dW into
        ``flat.grad`` (overwrites on the first micro-step of an accumulation window).

        ``g`` is a float or a zero-argument callable returning one.  It enters only as the alpha of the weight-gradient
        GEMMs at the END of the sweep, so a callable is resolved there: when it has to wait for a device value (the
        autograd upstream gradient), the whole data-gradient sweep is already queued behind the forward and the wait
        costs no GPU time."""
        if self._fwd_state is None:
            raise RuntimeError("Eagle3Engine.backward called without a training forward")
        B, S = self._fwd_state
        self._fwd_state = None
        c, T, f = self.cfg, self.T, self.flat
        b = self._buffers(B, S)
        N, Spad = b["N"], b["Spad"]
        H, I, hd, nh, nkv = c.hidden_size, c.intermediate_size, c.head_dim, c.num_attention_heads, c.num_key_value_heads
        Ht = c.target_hidden_size
        scale = 1.0 / math.sqrt(hd)
        kcol, vcol = slice(nh * hd, (nh + nkv) * hd), slice((nh + nkv) * hd, self.QW)
        ws = b["nws"]
        nat = self.hdp == hd
        plan = self._diag_plan if (nat and self.blocked_diag) else None
        for t in (b["dk"][:1] + b["dv"][:1]) if plan is not None else (b["dk"] + b["dv"]):   # (blocked form: the branch sums are
            t.zero_()                                              # first-touch writes of sf_attn_bwd_diag, only block 0's accumulate)
        # early_lm_head_wgrad (A/B option for multi-GPU boxes, default off): both operands of the lm_head weight gradient -- d(logits) of all
        # T steps (the fused CE wrote them during the FORWARD sweep) and the normed hidden states -- exist before the backward starts, and its
        # bucket is the largest (Vd x H: 262 MB at the headline).  Issued here, its all-reduce has the whole data-gradient sweep to hide in
        # instead of the rest of the weight-gradient phase.  Price: the upstream gradient g is the GEMM's alpha, so a callable g is resolved
        # NOW -- one host wait for the forward to drain, where the default path reads it at the end of the sweep for free.
        early_lm = None
        if self.early_lm_head_wgrad:
            if callable(g):
                g = float(g())
            early_lm = g
            ln_e = b["ln_s"] if c.norm_output else b["h_s"]
            K_e = self._lm_compact_K
            ops.gemm_tn(b["logits_s"] if K_e is None else b["logits_s"][:K_e], ln_e if K_e is None else ln_e[:K_e], f.gview("lm_head.weight"),
                        alpha=g, beta=0.0 if self.micro_in_window == 0 else 1.0, workspace=b["tn_ws"])
            if self.on_bucket_ready is not None:
                self.on_bucket_ready(f.slices["lm_head.weight"][0], f.slices["lm_head.weight"][1])
        nm = self._norm_micro
        first = {n: True for n in nm}

        def nacc(name):  # (dw_acc, accumulate?) for a norm weight within this micro-step
            acc = not first[name]
            first[name] = False
            return nm[name], acc

        # norm_colsum_batched: the weight-gradient partials of each launch ([rows / 16, H] fp32) are kept, side by side per weight, and reduced
        # by ONE column sum per weight after the sweep instead of one per launch (3 instead of 21 of the 22 launches of a 7-step sweep)
        batched = self.norm_colsum_batched
        part_rows = {n: 0 for n in self._NORMS_PER_STEP}

        def part_slot(name, rows):
            nbk = ops.rmsnorm_bwd_workspace(rows, H) // H
            seg = b["npart"][name][part_rows[name] * H:(part_rows[name] + nbk) * H]
            part_rows[name] += nbk
            return seg

        def norm_bwd(name, dy, x, w, rstd, *, dx, add):
            if batched and name in part_rows:
                ops.rmsnorm_bwd(dy, x, w, rstd, dx=dx, add=add, partial_only=True, workspace=part_slot(name, dy.shape[0]))
                return
            acc, a = nacc(name)
            ops.rmsnorm_bwd(dy, x, w, rstd, dx=dx, add=add, dw_acc=acc, dw_accumulate=a, workspace=ws)

        def norm_bwd2(name1, dy1, w1, name2, dy2, w2, x, rstd, *, dx, add):
            if batched:
                ops.rmsnorm_bwd2(dy1, w1, part_slot(name1, dy1.shape[0]), False, dy2, w2, part_slot(name2, dy1.shape[0]), False, x, rstd, dx=dx, add=add,
                                 workspace=ws, partial_only=True)
                return
            acc1, a1 = nacc(name1)
            acc2, a2 = nacc(name2)
            ops.rmsnorm_bwd2(dy1, w1, acc1, a1, dy2, w2, acc2, a2, x, rstd, dx=dx, add=add, workspace=ws)

        dh_next = None
        # h[k] feeds the final norm of step k - 1 AND the hidden_norm of step k: with the final norm in use (and H <= 4096) the
        # hidden_norm backward of step k is not run at the end of step k but together with the final-norm backward of step k - 1,
        # one pass over h[k] (sf_rmsnorm_bwd2): `pending` = (d(hidden_norm output), the residual-stream gradient that joins it)
        pair = c.norm_output and H <= 4096
        pending = None
        for k in range(T - 1, -1, -1):
            # final norm + (already taken) lm_head dgrad; residual-stream gradient of step k+1 joins here.
            # dh / dgu / dh1 / dqkv are written straight into their slot of the weight-gradient stash.
            dh, dgu, dh1, dqkv = b["dh"][k], b["dgu"][k], b["dh1"][k], b["dqkv"][k]
            if pending is not None:
                # (one rstd serves both norms of h[k+1]; in the compact form rstd_n[k] is zero on rows without a loss -- the hidden
                # norm's own copy, computed over all rows by the next step's forward, is the same value)
                norm_bwd2("norm.weight", b["dln"][k], f.view("norm.weight"), "midlayer.hidden_norm.weight", pending[0],
                          f.view("midlayer.hidden_norm.weight"), b["h"][k + 1],
                          b["rstd_h"][k + 1] if self._lm_compact_K is not None else b["rstd_n"][k], dx=dh, add=pending[1])
                pending = None
            elif c.norm_output:
                norm_bwd("norm.weight", b["dln"][k], b["h"][k + 1], f.view("norm.weight"), b["rstd_n"][k], dx=dh, add=dh_next)
            elif dh_next is None:   # lm_head reads the un-normed hidden state; `norm` gets no gradient
                dh.copy_(b["dln"][k])
            else:
                ops.add_bf16(b["dln"][k], dh_next, dh)
            # MLP
            ops.gemm_nt_swiglu_bwd(dh, self.wdT, b["gu"][k], dgu, b["dact"])   # down dgrad + d(SwiGLU) in its epilogue
            ops.gemm_nt(dgu, self.wguT, b["dpn"], workspace=b["nt_ws"])
            norm_bwd("midlayer.post_attention_layernorm.weight", b["dpn"], b["h1"][k], f.view("midlayer.post_attention_layernorm.weight"),
                     b["rstd_p"][k], dx=dh1, add=dh)
            # attention
            ops.gemm_nt(dh1, self.woT, b["do"][k], workspace=b["nt_ws"])
            qkv = b["qkv"][k]
            if self.hdp == hd:
                q, o_k, do, dq_out = qkv[:, :nh * hd], b["o"][k], b["do"][k], dqkv[:, :nh * hd]
                k0, v0 = b["qkv"][0][:, kcol], b["qkv"][0][:, vcol]
                kd = [b["qkv"][i][:, kcol] for i in range(1, k + 1)]
                vd = [b["qkv"][i][:, vcol] for i in range(1, k + 1)]
            else:
                self._pad_heads(b["do"][k], nh, b["dop"])
                q, o_k, do, dq_out = b["qp"][k], b["op"][k], b["dop"], b["dqp"]
                k0, v0, kd, vd = b["kp"][0], b["vp"][0], b["kp"][1:k + 1], b["vp"][1:k + 1]
            # (K_k / V_k receive their last contribution here -- steps k..T-1 have all run: for k >= 1 their gradients leave the
            # kernel as bf16, straight into the dqkv slot)
            delta = b["delta"][k]
            if plan is not None:
                # blocked form (diag_plan): dq_init of this step from all its branches; the branches of this step's block accumulate --
                # at the block's top step over every later step too (streamed), first touch -- and K_k / V_k, final here, leave as bf16
                for L in plan[k]:
                    rd, na = L["read"], L["nacc"]
                    fin = L["final"]
                    ops.attn_bwd_diag(
                        q=q if L["own"] else None, o=o_k if L["own"] else None, dout=do if L["own"] else None,
                        lse=b["lse"][k] if L["own"] else None, delta=delta if L["own"] else None,
                        dq_init=b["dq_init"] if (L["own"] and rd) else None, dq_accumulate=L["dq_accumulate"],
                        kd=[b["qkv"][i][:, kcol] for i in rd], vd=[b["qkv"][i][:, vcol] for i in rd],
                        dkd=[None if (L["first"][j] and fin[j]) else b["dk"][rd[j]] for j in range(na)],
                        dvd=[None if (L["first"][j] and fin[j]) else b["dv"][rd[j]] for j in range(na)],
                        first=L["first"], dk_out=[b["dqkv"][rd[j]][:, kcol] if fin[j] else None for j in range(na)],
                        dv_out=[b["dqkv"][rd[j]][:, vcol] if fin[j] else None for j in range(na)],
                        xq=[b["qkv"][x][:, :nh * hd] for x in L["stream"]], xdo=[b["do"][x] for x in L["stream"]],
                        xlse=[b["lse"][x] for x in L["stream"]], xdelta=[b["delta"][x] for x in L["stream"]],
                        B=B, S=S, nh=nh, nkv=nkv, hd=hd, scale=scale)
            else:
                fin = k > 0 and nat     # (per-step form: K_k / V_k receive their last contribution here and leave as bf16)
                ops.attn_bwd_pre(q, o_k, do, kd, vd, b["dk"][1:k + 1], b["dv"][1:k + 1], b["lse"][k], delta,
                                 b["dq_init"] if k > 0 else None, B=B, S=S, nh=nh, nkv=nkv, hd=self.hdp, scale=scale,
                                 dk_last=dqkv[:, kcol] if fin else None, dv_last=dqkv[:, vcol] if fin else None)
            ops.attn_bwd_dq(q, do, k0, v0, b["kvlen"], b["lse"][k], delta, b["dq_init"] if k > 0 else None, dq_out,
                            B=B, S=S, nh=nh, nkv=nkv, hd=self.hdp, scale=scale)
            ops.attn_bwd_dkv(q, do, k0, v0, b["kvlen"], b["lse"][k], delta, b["dk"][0], b["dv"][0], B=B, S=S, nh=nh,
                             nkv=nkv, hd=self.hdp, scale=scale, workspace=b["dkv_ws"])
            if self.hdp == hd:
                if k == 0:      # block 0's keys: the dK/dV kernel of this step added the last term
                    ops.cast_from_f32(b["dk"][0], dqkv[:, kcol])
                    ops.cast_from_f32(b["dv"][0], dqkv[:, vcol])
            else:
                self._unpad_heads(b["dqp"], nh, dqkv[:, :nh * hd])
                self._unpad_heads(b["dk"][k], nkv, dqkv[:, kcol])
                self._unpad_heads(b["dv"][k], nkv, dqkv[:, vcol])
            if self.mrope:
                ops.rope_(dqkv, nh + nkv, hd, b["cos_rows"][k], b["sin_rows"][k], b["pos"], 0, backward=True)
            else:
                ops.rope_(dqkv, nh + nkv, hd, self._rope_fwd[k][0], self._rope_fwd[k][1], b["pos"], k, backward=True)
            ops.gemm_nt(dqkv, self.wqkvT[H:], b["dxh"], workspace=b["nt_ws"])                 # hidden half of the QKV dgrad
            if pair and k > 0:
                pending = (b["dxh"], dh1)      # consumed at the top of step k - 1, before that step rewrites dxh
                continue
            dh_prev = b["dh_b"][0]
            norm_bwd("midlayer.hidden_norm.weight", b["dxh"], b["h"][k], f.view("midlayer.hidden_norm.weight"), b["rstd_h"][k],
                     dx=dh_prev, add=dh1)
            dh_next = dh_prev  # consumed (as `add`) by step k-1 before it is rewritten at the end of that step
        dh0 = dh_next
        if N % 64 == 0:
            fc_dy = dh0
        else:                       # N % 64 != 0 (ragged batches): zero-padded copy (K of the fc weight gradient)
            b["dh0_s"][:N].copy_(dh0)
            fc_dy = b["dh0_s"]
        # embedding half of the QKV backward, once for all steps.  The summed gradient enters the bf16 GEMMs as a
        # two-term expansion hi + lo for the weight gradient; input_layernorm.weight only needs the leading term (the
        # reference rounds every step's d(input) to bf16 before the norm backward, which is coarser than that).
        ops.shift_sum_split(b["dqkv_s"], b["ds_hi"], b["ds_lo"], T=T, B=B, S=S, Spad=Spad)   # sum over the steps, re-aligned, hi + lo
        ops.gemm_nt(b["ds_hi"], self.wqkvT[:H], b["dE"], workspace=b["nt_ws"])
        acc, a = nacc("midlayer.input_layernorm.weight")
        ops.rmsnorm_bwd(b["dE"][:b["Np_real"]], self.model.embed_tokens.weight.data, f.view("midlayer.input_layernorm.weight"),
                        b["rstd_e1"], dx=None, dw_acc=acc, dw_accumulate=a, workspace=b["nws_e"], ids_pad=b["ids"], S=Spad,
                        Spad=Spad, off=0)
        if c.fc_norm:
            ops.gemm_nt(dh0, self.wfcT, b["dhs"])
            hs = self._last_hs
            for i in range(3):
                norm_bwd(f"fc_norm.{i}.weight", b["dhs"][:, i * Ht:(i + 1) * Ht], hs[:, i * Ht:(i + 1) * Ht], f.view(f"fc_norm.{i}.weight"),
                         b["rstd_fc"][i], dx=None, add=None)

        # ---- deferred weight gradients: dW = dY^T . X over all T*N token rows of the natural-layout stashes
        # (sf_gemm_tn), bf16 straight into flat.grad in all-reduce bucket order
        if callable(g):
            g = float(g())          # (eagle3._TTTStep: the callable reads the flags from the same pinned copy as the upstream gradient)
        elif self._flags_set:
            # a caller driving forward(train=True) / backward() directly (no autograd wrapper): the device-side input checks of the forward
            # (row counts vs mask, position ids in range) must not go unread -- out-of-range ids were clamped, the step would train on wrong
            # angles.  One small read-back at the end of the data-gradient sweep, only for batches that were checked on the device.
            self.check_flags(self._flags.tolist())
        beta = 0.0 if self.micro_in_window == 0 else 1.0
        ln_s = b["ln_s"] if c.norm_output else b["h_s"]
        jobs = [
            ("lm_head.weight", "lm_head.weight", [(b["logits_s"][:self._lm_compact_K], ln_s[:self._lm_compact_K], f.gview("lm_head.weight"))
                                                  if self._lm_compact_K is not None else (b["logits_s"], ln_s, f.gview("lm_head.weight"))]),
            ("midlayer.mlp.gate_proj.weight", "midlayer.mlp.up_proj.weight", [(b["dgu_s"], b["pn_s"], self.g_gu)]),
            ("midlayer.mlp.down_proj.weight", "midlayer.mlp.down_proj.weight",
             [(b["dh_s"], b["act_s"], f.gview("midlayer.mlp.down_proj.weight"))]),
            ("midlayer.self_attn.q_proj.weight", "midlayer.self_attn.v_proj.weight",   # [QW, 2H] = [embedding | hidden] half
             [(b["ds2"], b["en2"], self.g_qkv[:, :H]), (b["dqkv_s"], b["hn_s"], self.g_qkv[:, H:])]),
            ("midlayer.self_attn.o_proj.weight", "midlayer.self_attn.o_proj.weight",
             [(b["dh1_s"], b["o_s"], f.gview("midlayer.self_attn.o_proj.weight"))]),
            ("fc.weight", "fc.weight", [(fc_dy, self._fc_x, f.gview("fc.weight"))]),
        ]
        if early_lm is not None:
            jobs = jobs[1:]          # (early_lm_head_wgrad: the first bucket went out before the data-gradient sweep)
        for first_name, last_name, gemms in jobs:
            for dy, x, gout in gemms:
                ops.gemm_tn(dy, x, gout, alpha=g, beta=beta, workspace=b["tn_ws"])
            if self.on_bucket_ready is not None:
                self.on_bucket_ready(f.slices[first_name][0], f.slices[last_name][1])
        if batched:
            for n, rows_n in part_rows.items():
                if rows_n:      # (a norm the forward does not read -- `norm` with norm_output = false -- has no partials and keeps its zeros)
                    ops.colsum_accum(b["npart"][n], rows_n, H, nm[n], accumulate=False)
        # norm weights: fp32 running total over the window, then one cast into the flat gradient
        lo = f.slices[self._norm_names[0]][0]
        if self._norm_contig:
            ops.axpy_f32(g, self._norm_micro_all, self._norm_total_all, accumulate=self.micro_in_window > 0)
            ops.cast_from_f32(self._norm_total_all.view(1, -1), f.grad[lo:].view(1, -1))
        else:
            for n in self._norm_names:
                ops.axpy_f32(g, nm[n], self._norm_total[n], accumulate=self.micro_in_window > 0)
                ops.cast_from_f32(self._norm_total[n].view(1, -1), f.gview(n).view(1, -1))
        if self.on_bucket_ready is not None:
            self.on_bucket_ready(lo, f.numel)
        self.micro_in_window += 1
        f.realias_grads()     # an external optimizer's zero_grad(set_to_none=True) must not detach .grad from flat.grad

    def bucket_bounds(self):
        """(lo, hi) element ranges of flat.grad in the order the backward sweep hands them to ``on_bucket_ready``"""
        f = self.flat
        pairs = [("lm_head.weight", "lm_head.weight"), ("midlayer.mlp.gate_proj.weight", "midlayer.mlp.up_proj.weight"),
                 ("midlayer.mlp.down_proj.weight", "midlayer.mlp.down_proj.weight"),
                 ("midlayer.self_attn.q_proj.weight", "midlayer.self_attn.v_proj.weight"),
                 ("midlayer.self_attn.o_proj.weight", "midlayer.self_attn.o_proj.weight"), ("fc.weight", "fc.weight")]
        return [(f.slices[a][0], f.slices[b][1]) for a, b in pairs] + [(f.slices[self._norm_names[0]][0], f.numel)]

    def end_window(self):
        """called by the optimizer after it consumed flat.grad"""
        self.micro_in_window = 0
        self.weights_version += 1
