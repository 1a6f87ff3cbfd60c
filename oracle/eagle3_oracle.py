"""CPU ORACLE for the EAGLE3 offline draft-training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (plain torch tensor algebra, autograd for
gradients; CPU by default -- it follows the device of its inputs, so the full-size parity
tests run the same code in fp32 on the GPU box as the checker) of the reference algorithm.  It is imported ONLY by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg -- never by the
product package ``specforge_amd`` (which fails loudly without its HIP library).

Parity status: PINNED.  ``oracle/gen_golden.py`` imports the real reference from
``/root/reference`` (sdpa backend, eager ``_compute_loss``) and
dumps inputs/outputs into ``tests/golden/*.pt``; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (losses, metrics, every parameter
gradient, integer artefacts bit-exact) and against the reference's own golden TTT
block mask (tests/test_utils/test_flex_attention.py:245-284).

Every function cites the reference file:line (relative to /root/reference) it follows.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- config
@dataclass
class DraftConfig:
    """Subset of the draft ``LlamaConfig`` the hot path reads
    (specforge/modeling/draft/llama3_eagle.py:542-566,1658-1693)."""

    hidden_size: int
    intermediate_size: int
    num_attention_heads: int
    num_key_value_heads: int
    vocab_size: int
    draft_vocab_size: int
    head_dim: Optional[int] = None
    target_hidden_size: Optional[int] = None
    max_position_embeddings: int = 2048
    rms_norm_eps: float = 1e-6
    rope_theta: float = 10000.0
    rope_scaling: Optional[dict] = None
    fc_norm: bool = False
    norm_output: bool = True
    initializer_range: float = 0.02

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        if self.target_hidden_size is None:
            self.target_hidden_size = self.hidden_size


PARAM_KEYS = [
    "fc.weight",
    "midlayer.hidden_norm.weight",
    "midlayer.input_layernorm.weight",
    "midlayer.self_attn.q_proj.weight",
    "midlayer.self_attn.k_proj.weight",
    "midlayer.self_attn.v_proj.weight",
    "midlayer.self_attn.o_proj.weight",
    "midlayer.post_attention_layernorm.weight",
    "midlayer.mlp.gate_proj.weight",
    "midlayer.mlp.up_proj.weight",
    "midlayer.mlp.down_proj.weight",
    "norm.weight",
    "lm_head.weight",
]


def param_shapes(cfg: DraftConfig) -> Dict[str, Tuple[int, ...]]:
    """Trainable tensor shapes; key names are the checkpoint/export contract
    (specforge/export/to_sglang.py:33-54; llama3_eagle.py:555-566,1513-1515,1674-1693)."""
    H, I, hd = cfg.hidden_size, cfg.intermediate_size, cfg.head_dim
    nh, nkv = cfg.num_attention_heads, cfg.num_key_value_heads
    shapes = {
        "fc.weight": (H, 3 * cfg.target_hidden_size),
        "midlayer.hidden_norm.weight": (H,),
        "midlayer.input_layernorm.weight": (H,),
        "midlayer.self_attn.q_proj.weight": (nh * hd, 2 * H),
        "midlayer.self_attn.k_proj.weight": (nkv * hd, 2 * H),
        "midlayer.self_attn.v_proj.weight": (nkv * hd, 2 * H),
        "midlayer.self_attn.o_proj.weight": (H, nh * hd),
        "midlayer.post_attention_layernorm.weight": (H,),
        "midlayer.mlp.gate_proj.weight": (I, H),
        "midlayer.mlp.up_proj.weight": (I, H),
        "midlayer.mlp.down_proj.weight": (H, I),
        "norm.weight": (H,),
        "lm_head.weight": (cfg.draft_vocab_size, H),
    }
    if cfg.fc_norm:
        for i in range(3):
            shapes[f"fc_norm.{i}.weight"] = (cfg.target_hidden_size,)
    return shapes


def init_params(cfg: DraftConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Random init in the spirit of ``post_init`` (normal(0, initializer_range) for
    matrices, ones for norm weights).  Not bit-identical to HF init; golden tests
    load the reference's own weights instead."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, shp in param_shapes(cfg).items():
        if len(shp) == 1:
            out[k] = torch.ones(shp, dtype=dtype)
        else:
            out[k] = (torch.randn(shp, generator=g) * cfg.initializer_range).to(dtype)
    return out


# --------------------------------------------------------------------------- pieces
def padding_left_shift(t: torch.Tensor) -> torch.Tensor:
    """``padding(tensor, left=False)``: shift left by one along dim 1, zero fill
    (specforge/utils.py:128-135)."""
    return torch.cat((t[:, 1:], torch.zeros_like(t[:, -1:])), dim=1)


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """``LlamaRMSNorm.forward`` (llama3_eagle.py:1561-1567): fp32 statistics, cast the
    normalised value back to the input dtype, then multiply by the weight."""
    dt = x.dtype
    xf = x.to(torch.float32)
    var = xf.pow(2).mean(-1, keepdim=True)
    xf = xf * torch.rsqrt(var + eps)
    return w * xf.to(dt)


def rope_inv_freq(cfg: DraftConfig, n_pos: Optional[int] = None) -> torch.Tensor:
    """inv_freq incl. the llama3 / linear / dynamic-NTK / yarn variants (llama3_eagle.py:235-276,315-540)."""
    dim = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2).float() / dim))
    rs = cfg.rope_scaling
    if rs is None:
        return inv_freq
    rtype = rs.get("rope_type", rs.get("type"))
    if rtype in (None, "default", "mrope"):    # mrope: plain frequencies, three position axes (apply_mrope)
        return inv_freq
    if rtype == "llama3":
        factor = rs.get("factor") or 1.0
        lo, hi = rs["low_freq_factor"], rs["high_freq_factor"]
        orig = rs["original_max_position_embeddings"]
        low_wl, high_wl = orig / lo, orig / hi
        wl = 2 * math.pi / inv_freq
        smooth = (orig / wl - lo) / (hi - lo) if lo != hi else 0
        return torch.where(
            wl < high_wl,
            inv_freq,
            torch.where(wl > low_wl, inv_freq / factor, (1 - smooth) * inv_freq / factor + smooth * inv_freq),
        )
    if rtype == "linear":
        return inv_freq  # positions are scaled instead, see rope_tables
    if rtype == "dynamic":  # llama3_eagle.py:361-372, evaluated for the pre-built cache length (281-285)
        seq_len = n_pos if n_pos is not None else cfg.max_position_embeddings + 20
        if seq_len <= cfg.max_position_embeddings:
            return inv_freq
        f = rs["factor"]
        base = cfg.rope_theta * ((f * seq_len / cfg.max_position_embeddings) - (f - 1)) ** (dim / (dim - 2))
        return 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    if rtype == "yarn":  # llama3_eagle.py:430-508
        f, orig, base = rs["factor"], rs["original_max_position_embeddings"], cfg.rope_theta

        def corr(rot):
            return (dim * math.log(orig / (rot * 2 * math.pi))) / (2 * math.log(base))

        low, high = max(math.floor(corr(rs["beta_fast"])), 0), min(math.ceil(corr(rs["beta_slow"])), dim - 1)
        if low == high:
            high += 0.001
        ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (high - low), 0, 1)
        mask = 1.0 - ramp
        pw = base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim)
        return (1.0 / (f * pw)) * (1 - mask) + (1.0 / pw) * mask
    raise NotImplementedError(f"oracle: rope type {rtype}")


def rope_tables(cfg: DraftConfig, n_pos: int, dtype=torch.float32) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin cache ``[n_pos, head_dim]`` built in fp32 then cast to the activation
    dtype (llama3_eagle.py:287-312); yarn scales both by mscale/mscale_all_dim (510-540)."""
    inv_freq = rope_inv_freq(cfg, n_pos)
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    rs = cfg.rope_scaling or {}
    rtype = rs.get("rope_type", rs.get("type"))
    if rtype == "linear":
        t = t / rs["factor"]
    amp = 1.0
    if rtype == "yarn":
        ms = lambda scale, m: 1.0 if scale <= 1 else 0.1 * m * math.log(scale) + 1.0
        amp = float(ms(rs["factor"], rs["mscale"]) / ms(rs["factor"], rs["mscale_all_dim"]))
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * amp).to(dtype), (emb.sin() * amp).to(dtype)


class RopeCache:
    """The state of ``LlamaRotaryEmbedding`` (llama3_eagle.py:281-312): the cos / sin cache is built for max_pos + 20 positions and
    REBUILT for exactly ``seq_len`` positions whenever a call asks for more (303-306; the attention asks for seq_len = q_len + lck,
    733) -- for dynamic NTK the rebuild also re-derives the base from that length (362-371) -- and the rebuilt cache stays (it is a
    module buffer), also for later forwards when the same object is passed again.  ``get(seq_len)`` returns the first ``seq_len``
    rows as ``forward`` does (308-311), so a position id >= seq_len is an IndexError in ``apply_rope`` exactly like in the reference."""

    def __init__(self, cfg: DraftConfig, dtype, device):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.len = cfg.max_position_embeddings + 20
        self.cos, self.sin = (t.to(device) for t in rope_tables(cfg, self.len, dtype))

    def get(self, seq_len: int):
        if seq_len > self.len:
            self.len = seq_len
            self.cos, self.sin = (t.to(self.device) for t in rope_tables(self.cfg, seq_len, self.dtype))
        return self.cos[:seq_len], self.sin[:seq_len]

    def plain_rows(self, n: int):
        """mrope (llama3_eagle.py:389-427) computes its angles analytically from the ids -- no cache, no limit; this restatement
        gathers them from the plain table, which is exact for any length"""
        if n > self.cos.shape[0]:
            self.cos, self.sin = (t.to(self.device) for t in rope_tables(self.cfg, n, self.dtype))
        return self.cos, self.sin


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin, position_ids):
    """``apply_rotary_pos_emb`` (llama3_eagle.py:133-142); q,k are [B, heads, S, hd]."""
    c = cos[position_ids].unsqueeze(1)
    s = sin[position_ids].unsqueeze(1)
    return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s


def apply_mrope(q, k, cos, sin, position_ids, mrope_section):
    """``LlamaMutiRotaryEmbedding.forward`` + ``apply_multimodal_rotary_pos_emb`` (llama3_eagle.py:389-427,145-182):
    position_ids is [3, B, S] (temporal, height, width); the head's rotary channels are cut into the sections
    ``mrope_section * 2`` and section i takes its angle from axis i % 3.  cos/sin are the plain tables."""
    c3, s3 = cos[position_ids], sin[position_ids]                  # [3, B, S, hd]
    sec = list(mrope_section) * 2
    c = torch.cat([m[i % 3] for i, m in enumerate(c3.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    s = torch.cat([m[i % 3] for i, m in enumerate(s3.split(sec, dim=-1))], dim=-1).unsqueeze(1)
    return q * c + rotate_half(q) * s, k * c + rotate_half(k) * s


def additive_attention_mask(attention_mask: torch.Tensor, S: int, dtype) -> torch.Tensor:
    """``prepare_decoder_attention_mask`` (modeling/draft/base.py:64-96,
    modeling/_mask_utils.py:29-73): causal + key-padding, both ``finfo(dtype).min``."""
    B = attention_mask.shape[0]
    minv = torch.finfo(dtype).min
    causal = torch.full((S, S), minv)
    idx = torch.arange(S)
    causal.masked_fill_(idx < (idx + 1).view(S, 1), 0)
    causal = causal.to(dtype)[None, None].expand(B, 1, S, S)
    expanded = attention_mask[:, None, None, :].expand(B, 1, S, S).to(dtype)
    inv = 1.0 - expanded
    inv = inv.masked_fill(inv.to(torch.bool), minv)
    return inv + causal


def ttt_mask_dense(seq_len_valid: int, q_len: int, n_blocks: int) -> torch.Tensor:
    """Integer TTT branch mask ``[q_len, n_blocks*q_len]`` (1 = attend), the integer
    artefact of modeling/draft/flex_attention.py:108-127: block 0 causal, one diagonal
    per later block, both restricted to ``idx < seq_len_valid``."""
    q = torch.arange(q_len)[:, None]
    kv = torch.arange(n_blocks * q_len)[None, :]
    causal = (q >= kv) & (kv < seq_len_valid) & (q < seq_len_valid)
    suffix = (kv >= q_len) & ((kv % q_len) < seq_len_valid) & (((kv - q) % q_len) == 0)
    return (causal | suffix).to(torch.int32)


def ttt_attention(q, cache_k: List[torch.Tensor], cache_v: List[torch.Tensor], add_mask, head_dim: int):
    """sdpa-backend TTT attention (llama3_eagle.py:745-778).  q, cache_*[i]: [B,nh,S,hd]
    (kv already repeated).  Scores of block 0 are causal-masked; every later block
    contributes its diagonal column only; softmax in fp32 over S + (lck-1) columns."""
    S = q.shape[2]
    k0, v0 = cache_k[0], cache_v[0]
    w = torch.matmul(q, k0.transpose(2, 3)) / math.sqrt(head_dim)
    w = w + add_mask
    for i in range(1, len(cache_k)):
        wi = (q * cache_k[i]).sum(-1) / math.sqrt(head_dim)
        w = torch.cat((w, wi[..., None]), dim=-1)
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    out = torch.matmul(w[..., :S], v0)
    for i in range(1, len(cache_k)):
        out = out + w[..., S + i - 1][..., None] * cache_v[i]
    return out


def repeat_kv(x: torch.Tensor, n_rep: int) -> torch.Tensor:
    """llama3_eagle.py:96-105."""
    if n_rep == 1:
        return x
    B, nkv, S, hd = x.shape
    return x[:, :, None].expand(B, nkv, n_rep, S, hd).reshape(B, nkv * n_rep, S, hd)


def decoder_layer(p, cfg: DraftConfig, emb, hidden, cache, add_mask, position_ids, rope: "RopeCache"):
    """``LlamaDecoderLayer.forward`` + ``LlamaAttention.forward`` cache branch +
    ``LlamaMLP`` (llama3_eagle.py:1598-1650, 661-785, 1518-1549)."""
    B, S, H = hidden.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    residual = hidden
    hn = rmsnorm(hidden, p["midlayer.hidden_norm.weight"], cfg.rms_norm_eps)
    en = rmsnorm(emb, p["midlayer.input_layernorm.weight"], cfg.rms_norm_eps)
    x = torch.cat((en, hn), dim=-1)
    q = F.linear(x, p["midlayer.self_attn.q_proj.weight"]).view(B, S, nh, hd).transpose(1, 2)
    k = F.linear(x, p["midlayer.self_attn.k_proj.weight"]).view(B, S, nkv, hd).transpose(1, 2)
    v = F.linear(x, p["midlayer.self_attn.v_proj.weight"]).view(B, S, nkv, hd).transpose(1, 2)
    lck = len(cache[0])
    rs = cfg.rope_scaling or {}
    if rs.get("rope_type", rs.get("type")) == "mrope":
        cos, sin = rope.plain_rows(int(position_ids.max()) + lck + 1)
        q, k = apply_mrope(q, k, cos, sin, position_ids + lck, rs["mrope_section"])
    else:
        cos, sin = rope.get(S + lck)                    # rotary_emb(x, seq_len=q_len + lck), llama3_eagle.py:733
        q, k = apply_rope(q, k, cos, sin, position_ids + lck)
    k = repeat_kv(k, nh // nkv)
    v = repeat_kv(v, nh // nkv)
    cache[0] = cache[0] + [k]
    cache[1] = cache[1] + [v]
    a = ttt_attention(q, cache[0], cache[1], add_mask, hd)
    a = a.transpose(1, 2).contiguous().reshape(B, S, nh * hd)
    a = F.linear(a, p["midlayer.self_attn.o_proj.weight"])
    hidden = residual + a
    residual = hidden
    pn = rmsnorm(hidden, p["midlayer.post_attention_layernorm.weight"], cfg.rms_norm_eps)
    g = F.linear(pn, p["midlayer.mlp.gate_proj.weight"])
    u = F.linear(pn, p["midlayer.mlp.up_proj.weight"])
    d = F.linear(F.silu(g) * u, p["midlayer.mlp.down_proj.weight"])
    return residual + d


def project_hidden_states(p, cfg: DraftConfig, hs):
    """llama3_eagle.py:1762-1770 (optional 3x fc_norm, then fc)."""
    if cfg.fc_norm:
        chunks = hs.chunk(3, dim=-1)
        hs = torch.cat(
            [rmsnorm(c, p[f"fc_norm.{i}.weight"], cfg.rms_norm_eps) for i, c in enumerate(chunks)], dim=-1
        )
    return F.linear(hs, p["fc.weight"])


def compute_logits(p, cfg: DraftConfig, hidden):
    """llama3_eagle.py:1772-1777."""
    if cfg.norm_output:
        hidden = rmsnorm(hidden, p["norm.weight"], cfg.rms_norm_eps)
    return F.linear(hidden, p["lm_head.weight"])


def compute_target_p(target_logits, t2d, loss_mask):
    """``_compute_target_p`` (algorithms/eagle3/model.py:487-501).  loss_mask [B,S,1]."""
    th = target_logits.float()
    ids = th.argmax(-1)
    pos_mask = t2d[ids][..., None].int() * loss_mask
    dth = th[..., t2d]
    target_p = torch.softmax(dth, dim=2)
    lse = torch.logsumexp(th, dim=-1, keepdim=True)
    target_p_on_draft = torch.exp(dth - lse)
    return target_p, target_p_on_draft, ids, pos_mask


def soft_ce_loss(logits, target_p, position_mask):
    """eager ``_compute_loss`` (core/loss.py:15-21): mean over ALL B*S rows."""
    lp = torch.log_softmax(logits.float(), dim=2)
    return -torch.sum(position_mask * (target_p * lp), 2).mean()


def acceptance_rate(logits, target_p_on_draft, position_mask, eps=1e-8):
    """``compute_acceptance_rate`` first output (core/lk_loss.py:43-80)."""
    dp = torch.softmax(logits.to(torch.float32), dim=-1).to(target_p_on_draft.dtype)
    per_tok = torch.minimum(target_p_on_draft, dp).sum(-1)
    mask = position_mask.squeeze(-1).to(per_tok.dtype)
    return (per_tok * mask).sum() / mask.sum().clamp_min(eps)


def acceptance_and_log_rate(logits, target_p_on_draft, position_mask, eps=1e-8):
    """both outputs of ``compute_acceptance_rate`` (core/lk_loss.py:52-80), differentiable."""
    dp = torch.softmax(logits.to(torch.float32), dim=-1).to(target_p_on_draft.dtype)
    per_tok = torch.minimum(target_p_on_draft, dp).sum(-1)
    mask = position_mask.squeeze(-1).to(per_tok.dtype)
    den = mask.sum().clamp_min(eps)
    log_tok = torch.where(per_tok > 0, torch.log(per_tok), torch.zeros_like(per_tok))
    return (per_tok * mask).sum() / den, (log_tok * mask).sum() / den


def lk_loss(kl_loss, acc_rate, log_acc_rate, lk_loss_type: str, kl_scale: float, kl_decay: float):
    """``compute_lk_loss`` (core/lk_loss.py:83-99)."""
    if lk_loss_type == "alpha":
        return -log_acc_rate
    if lk_loss_type == "lambda":
        w = kl_scale * torch.exp(-kl_decay * acc_rate.detach())
        return w * kl_loss + (1 - w) * (1 - acc_rate)
    raise ValueError(f"Unknown lk loss type: {lk_loss_type}")


@dataclass
class Eagle3Out:
    plosses: List[torch.Tensor] = field(default_factory=list)
    acces: List[torch.Tensor] = field(default_factory=list)
    acceptance_rates: List[torch.Tensor] = field(default_factory=list)
    acc_corrects: List[torch.Tensor] = field(default_factory=list)
    acc_denoms: List[torch.Tensor] = field(default_factory=list)
    target_token_ids: Optional[torch.Tensor] = None
    position_mask: Optional[torch.Tensor] = None
    logits: List[torch.Tensor] = field(default_factory=list)
    loss: Optional[torch.Tensor] = None


def eagle3_forward(
    p: Dict[str, torch.Tensor],
    cfg: DraftConfig,
    *,
    embed_weight: torch.Tensor,
    target_head_weight: torch.Tensor,
    t2d: torch.Tensor,
    d2t: torch.Tensor,
    input_ids: torch.Tensor,      # [B,S] raw (unshifted)
    attention_mask: torch.Tensor,  # [B,S]
    loss_mask: torch.Tensor,       # [B,S]
    hidden_state: torch.Tensor,    # [B,S,3Ht]  (aux hidden states)
    target_hidden: torch.Tensor,   # [B,S,Ht]   (final hidden state of the target)
    ttt_length: int = 7,
    ploss_decay: float = 0.8,
    position_ids: Optional[torch.Tensor] = None,
    keep_logits: bool = False,
    lk_loss_type: Optional[str] = None,
    kl_scale: float = 1.0,
    kl_decay: float = 1.0,
    rope_cache: Optional[RopeCache] = None,
) -> Eagle3Out:
    """``Eagle3TrainStrategy.forward_loss`` -> ``OnlineEagle3Model.forward``
    (training/strategies/base.py:237-304; algorithms/eagle3/model.py:244-442),
    sdpa backend, full-vocab teacher."""
    out = Eagle3Out()
    B, S, _ = hidden_state.shape
    dt = hidden_state.dtype
    dev = hidden_state.device   # the same restatement is the full-size checker on cuda (fp32) in tests/test_full_size.py
    # TargetHead.preprocess: shift target + input_ids left by one (target_head.py:103-108)
    target_hidden = padding_left_shift(target_hidden)
    input_ids = padding_left_shift(input_ids)
    lm = loss_mask[..., None]
    with torch.no_grad():
        target_logits = F.linear(target_hidden.to(target_head_weight.dtype), target_head_weight)
        target_p, target_pod, ids, pos_mask = compute_target_p(target_logits, t2d, lm)
        # _compute_target_p_padded (eagle3/model.py:445-484)
        target_p = F.pad(target_p, (0, 0, 0, ttt_length), value=1 / target_p.shape[-1])
        target_pod = F.pad(target_pod, (0, 0, 0, ttt_length), value=0.0)
        ids_p = F.pad(ids, (0, ttt_length), value=0)
    out.target_token_ids, out.position_mask = ids, pos_mask

    hidden = project_hidden_states(p, cfg, hidden_state)
    if position_ids is None:
        position_ids = torch.arange(0, S, dtype=torch.long).unsqueeze(0)
    position_ids = position_ids.to(dev)
    add_mask = additive_attention_mask(attention_mask.bool().cpu() if attention_mask is not None else torch.ones(B, S, dtype=torch.bool), S, dt).to(dev)
    rope = rope_cache if rope_cache is not None else RopeCache(cfg, dt, dev)   # (pass one object to several forwards to carry the
    # rotary module's cache over, as the reference's draft model does)
    cache = [[], []]
    g_ids, g_pm, g_lm = input_ids, pos_mask, lm
    for idx in range(ttt_length):
        tp = target_p[:, idx : idx + S]
        tpod = target_pod[:, idx : idx + S]
        tid = ids_p[:, idx : idx + S]
        emb = F.embedding(g_ids, embed_weight).to(dt)
        hidden = decoder_layer(p, cfg, emb, hidden, cache, add_mask, position_ids, rope)
        logits = compute_logits(p, cfg, hidden)
        if keep_logits:
            out.logits.append(logits.detach())
        with torch.no_grad():  # _acc_and_loss (eagle3/model.py:161-173)
            pred = logits.argmax(-1)
            pred_t = pred + d2t[pred]
            correct = ((pred_t == tid) * g_lm.squeeze(-1)).sum()
            denom = g_lm.sum().clamp_min(1e-6)
            out.acc_corrects.append(correct)
            out.acc_denoms.append(denom)
            out.acces.append(correct / denom)
        kl = soft_ce_loss(logits, tp, g_pm)
        if lk_loss_type is None:   # eagle3/model.py:78: acceptance under no_grad unless an LK objective is on
            with torch.no_grad():
                out.acceptance_rates.append(acceptance_rate(logits, tpod, g_pm))
            out.plosses.append(kl)
        else:
            acc_rate, log_acc = acceptance_and_log_rate(logits, tpod, g_pm)
            out.acceptance_rates.append(acc_rate.detach())
            out.plosses.append(lk_loss(kl, acc_rate, log_acc, lk_loss_type, kl_scale, kl_decay))
        if idx != ttt_length - 1:
            g_ids = padding_left_shift(g_ids)
            g_pm = padding_left_shift(g_pm)
            g_lm = padding_left_shift(g_lm)
    out.loss = sum((ploss_decay ** i) * l for i, l in enumerate(out.plosses))
    return out


# --------------------------------------------------------------------------- optimizer
def adamw_clip_step(params, grads, m, v, step, *, lr, max_grad_norm=0.5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """``BF16Optimizer.step`` math on fp32 masters (optimizer.py:95-168) with
    ``torch.optim.AdamW`` defaults.  ``grads`` are the (bf16) model grads; returns the
    global grad norm.  Updates params/m/v in place (all fp32)."""
    total_sq = sum(g.float().square().sum() for g in grads)
    norm = total_sq.sqrt()
    clip = torch.clamp(max_grad_norm / (norm + 1e-6), max=1.0)
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for p_, g_, m_, v_ in zip(params, grads, m, v):
        g32 = g_.float() * clip
        p_.mul_(1 - lr * weight_decay)
        m_.mul_(b1).add_(g32, alpha=1 - b1)
        v_.mul_(b2).addcmul_(g32, g32, value=1 - b2)
        denom = (v_.sqrt() / math.sqrt(bc2)).add_(eps)
        p_.addcdiv_(m_, denom, value=-lr / bc1)
    return norm


def cosine_warmup_lr(step: int, base_lr: float, total_steps: int, warmup_steps: int, eta_min: float = 0.0) -> float:
    """LR in effect after ``step`` scheduler steps of ``CosineAnnealingWarmupLR``
    (lr_scheduler.py:72-119): linear warmup ``(k+1)/W*lr``; afterwards the reference
    chains torch's *recursive* ``CosineAnnealingLR`` whose first post-warmup call runs the
    recursion from ``last_epoch == 0`` (factor ``2/(1+cos(pi/T))``), so the whole cosine
    branch is ``base*(1+cos(pi*e/T))/(1+cos(pi/T))`` with ``e = step-W`` (verified against
    the reference run in tests/golden/optimizer_bf16.pt; torch 2.10 behaviour)."""
    if step < warmup_steps:
        return (step + 1) / warmup_steps * base_lr
    e = step - warmup_steps
    T = total_steps - warmup_steps
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi / T))


def distributed_sampler_indices(size, *, dp_rank, dp_size, seed, epoch, shuffle=True):
    """``_distributed_sampler_indices`` (specforge/launch.py:219-239)."""
    if size <= 0:
        return []
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(seed) + int(epoch))
        idx = torch.randperm(size, generator=g).tolist()
    else:
        idx = list(range(size))
    total = math.ceil(size / dp_size) * dp_size
    pad = total - len(idx)
    if pad:
        reps = math.ceil(pad / len(idx))
        idx.extend((idx * reps)[:pad])
    return idx[dp_rank:total:dp_size]


# --------------------------------------------------------------------------- synthetic data
def make_vocab_mapping(vocab: int, draft_vocab: int, seed: int = 0):
    """tests/test_runtime/_fixtures.py:121-128."""
    g = torch.Generator().manual_seed(seed)
    draft_ids = torch.randperm(vocab, generator=g)[:draft_vocab].sort().values
    t2d = torch.zeros(vocab, dtype=torch.bool)
    t2d[draft_ids] = True
    d2t = (draft_ids - torch.arange(draft_vocab)).to(torch.int64)
    return t2d, d2t


def make_batch(cfg: DraftConfig, B: int, S: int, seed: int = 0, dtype=torch.bfloat16, lengths=None):
    """Synthetic offline features shaped like tests/test_runtime/_fixtures.py:131-149,
    normalised like algorithms/eagle3/data.py:10-27 and right-padded like the collator."""
    g = torch.Generator().manual_seed(seed)
    Ht = cfg.target_hidden_size
    input_ids = torch.randint(0, cfg.vocab_size, (B, S), generator=g)
    target_hidden = torch.randn(B, S, Ht, generator=g).to(dtype)
    hidden_state = torch.randn(B, S, 3 * Ht, generator=g).to(dtype)
    loss_mask = torch.ones(B, S, dtype=torch.long)
    attention_mask = torch.ones(B, S, dtype=torch.long)
    if lengths is None:
        lengths = [S] * B
    for b, L in enumerate(lengths):
        loss_mask[b, L - 1 :] = 0  # last valid token + padding
        attention_mask[b, L:] = 0
        input_ids[b, L:] = 0
        target_hidden[b, L:] = 0
        hidden_state[b, L:] = 0
    return dict(
        input_ids=input_ids,
        attention_mask=attention_mask,
        loss_mask=loss_mask,
        hidden_state=hidden_state,
        target=target_hidden,
    )
