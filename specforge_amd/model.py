"""EAGLE3 draft model container: same constructor surface, attribute names, buffers and
state-dict keys as the reference's ``LlamaForCausalLMEagle3``
(specforge/modeling/draft/llama3_eagle.py:1653-1798; ABC specforge/modeling/draft/base.py:38-206),
so checkpoints / ``export --to sglang`` stay interchangeable:

    embed_tokens.weight (frozen, not checkpointed), fc.weight, [fc_norm.{0,1,2}.weight],
    midlayer.self_attn.{q,k,v,o}_proj.weight, midlayer.mlp.{gate,up,down}_proj.weight,
    midlayer.{hidden_norm,input_layernorm,post_attention_layernorm}.weight, norm.weight,
    lm_head.weight, buffers t2d (bool[V]) and d2t (int64[Vd]).

The module holds no math: the TTT step is executed by ``specforge_amd.engine.Eagle3Engine``
on the HIP kernels.  ``FlatParams.adopt`` re-points every trainable parameter into ONE
contiguous bf16 buffer (q|k|v and gate|up adjacent, so they are also fused GEMM operands)
with matching flat gradient / fp32-master / Adam-moment buffers: one grad-norm kernel,
one AdamW kernel, all-reduce on contiguous slices with no bucket copies.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn


@dataclass
class DraftConfig:
    """The fields of the draft ``LlamaConfig`` the hot path reads (llama3_eagle.py:542-566,1658-1693)."""

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
    pad_token_id: Optional[int] = None

    def __post_init__(self):
        if self.head_dim is None:
            self.head_dim = self.hidden_size // self.num_attention_heads
        if self.target_hidden_size is None:
            self.target_hidden_size = self.hidden_size

    @classmethod
    def from_hf(cls, cfg) -> "DraftConfig":
        """Accepts a transformers ``LlamaConfig``-like object or a dict (configs/*.json of the reference)."""
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        rs = get("rope_scaling")
        if rs is None and get("rope_parameters") is not None:
            rp = dict(get("rope_parameters"))
            theta = rp.pop("rope_theta", None)
            rs = rp if rp.get("rope_type", "default") not in (None, "default") else None
        else:
            theta = None
        return cls(
            hidden_size=get("hidden_size"), intermediate_size=get("intermediate_size"),
            num_attention_heads=get("num_attention_heads"),
            num_key_value_heads=get("num_key_value_heads") or get("num_attention_heads"),
            vocab_size=get("vocab_size"), draft_vocab_size=get("draft_vocab_size"), head_dim=get("head_dim"),
            target_hidden_size=get("target_hidden_size"), max_position_embeddings=get("max_position_embeddings", 2048),
            rms_norm_eps=get("rms_norm_eps", 1e-6), rope_theta=theta or get("rope_theta", 10000.0) or 10000.0,
            rope_scaling=rs, fc_norm=bool(get("fc_norm", False)), norm_output=bool(get("norm_output", True)),
            initializer_range=get("initializer_range", 0.02), pad_token_id=get("pad_token_id"),
        )


def _yarn_correction_dim(num_rotations, dim, base, max_pos):
    return (dim * math.log(max_pos / (num_rotations * 2 * math.pi))) / (2 * math.log(base))


def _yarn_mscale(scale, mscale):
    return 1.0 if scale <= 1 else 0.1 * mscale * math.log(scale) + 1.0


def rope_tables(cfg: DraftConfig, dtype=torch.bfloat16) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin caches [max_pos+20, head_dim]: fp32 then cast to the activation dtype
    (llama3_eagle.py:218-312).  Variants: llama3 frequency smoothing (235-276), linear position scaling
    (315-344), dynamic NTK (347-386: the cache is pre-built for max_pos+20 > max_pos positions, so the base is
    already rescaled for that length), yarn (430-540: blended frequencies and an amplitude factor on cos/sin).
    mrope (3-D position ids) is not on the text-only EAGLE3 path."""
    dim = cfg.head_dim
    n_pos = cfg.max_position_embeddings + 20
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2).float() / dim))
    rs = cfg.rope_scaling or {}
    rtype = rs.get("rope_type", rs.get("type"))
    amp = 1.0
    if rtype == "llama3":
        factor = rs.get("factor") or 1.0
        lo, hi = rs["low_freq_factor"], rs["high_freq_factor"]
        orig = rs["original_max_position_embeddings"]
        wl = 2 * math.pi / inv_freq
        smooth = (orig / wl - lo) / (hi - lo) if lo != hi else 0
        inv_freq = torch.where(wl < orig / hi, inv_freq,
                               torch.where(wl > orig / lo, inv_freq / factor,
                                           (1 - smooth) * inv_freq / factor + smooth * inv_freq))
    elif rtype == "dynamic":
        f = rs["factor"]
        mp = cfg.max_position_embeddings
        base = cfg.rope_theta * ((f * n_pos / mp) - (f - 1)) ** (dim / (dim - 2))
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
    elif rtype == "yarn":
        f = rs["factor"]
        orig = rs["original_max_position_embeddings"]
        pw = cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim)
        freq_extra, freq_inter = 1.0 / pw, 1.0 / (f * pw)
        low = max(math.floor(_yarn_correction_dim(rs["beta_fast"], dim, cfg.rope_theta, orig)), 0)
        high = min(math.ceil(_yarn_correction_dim(rs["beta_slow"], dim, cfg.rope_theta, orig)), dim - 1)
        hi_ = high + 0.001 if low == high else high
        ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (hi_ - low), 0, 1)
        mask = 1.0 - ramp
        inv_freq = freq_inter * (1 - mask) + freq_extra * mask
        amp = float(_yarn_mscale(f, rs["mscale"]) / _yarn_mscale(f, rs["mscale_all_dim"]))
    elif rtype not in (None, "default", "linear"):
        raise NotImplementedError(f"specforge_amd: rope type {rtype!r} is not on the EAGLE3 offline path")
    t = torch.arange(n_pos, dtype=inv_freq.dtype)
    if rtype == "linear":
        t = t / rs["factor"]
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return (emb.cos() * amp).to(dtype).contiguous(), (emb.sin() * amp).to(dtype).contiguous()


class _W(nn.Module):
    """a bias-free projection / norm: just owns ``weight`` under the reference's name"""

    def __init__(self, *shape, ones=False, std=0.02, dtype=torch.bfloat16, device=None):
        super().__init__()
        w = torch.ones(*shape, device=device) if ones else torch.randn(*shape, device=device) * std
        self.weight = nn.Parameter(w.to(dtype=dtype))


class _Attn(nn.Module):
    def __init__(self, c: DraftConfig, **kw):
        super().__init__()
        H2, hd = 2 * c.hidden_size, c.head_dim
        self.q_proj = _W(c.num_attention_heads * hd, H2, **kw)
        self.k_proj = _W(c.num_key_value_heads * hd, H2, **kw)
        self.v_proj = _W(c.num_key_value_heads * hd, H2, **kw)
        self.o_proj = _W(c.hidden_size, c.num_attention_heads * hd, **kw)


class _MLP(nn.Module):
    def __init__(self, c: DraftConfig, **kw):
        super().__init__()
        self.gate_proj = _W(c.intermediate_size, c.hidden_size, **kw)
        self.up_proj = _W(c.intermediate_size, c.hidden_size, **kw)
        self.down_proj = _W(c.hidden_size, c.intermediate_size, **kw)


class _Layer(nn.Module):
    def __init__(self, c: DraftConfig, **kw):
        super().__init__()
        self.self_attn = _Attn(c, **kw)
        self.mlp = _MLP(c, **kw)
        nkw = dict(kw, ones=True)
        self.hidden_norm = _W(c.hidden_size, **nkw)
        self.input_layernorm = _W(c.hidden_size, **nkw)
        self.post_attention_layernorm = _W(c.hidden_size, **nkw)


class LlamaForCausalLMEagle3(nn.Module):
    """Parameter container with the reference's names and registration order
    (llama3_eagle.py:1658-1700): embed_tokens, midlayer, fc, [fc_norm], norm, lm_head, t2d, d2t."""

    def __init__(self, config, attention_backend: str = "hip", dtype=torch.bfloat16, device=None):
        super().__init__()
        c = config if isinstance(config, DraftConfig) else DraftConfig.from_hf(config)
        self.config = c
        self.attention_backend = attention_backend
        self.vocab_size, self.draft_vocab_size = c.vocab_size, c.draft_vocab_size
        self.target_hidden_size = c.target_hidden_size
        kw = dict(std=c.initializer_range, dtype=dtype, device=device)
        self.embed_tokens = _W(c.vocab_size, c.hidden_size, **kw)
        self.midlayer = _Layer(c, **kw)
        self.fc = _W(c.hidden_size, 3 * c.target_hidden_size, **kw)
        self.fc_norm = nn.ModuleList([_W(c.target_hidden_size, **dict(kw, ones=True)) for _ in range(3)]) if c.fc_norm else None
        self.norm = _W(c.hidden_size, **dict(kw, ones=True))
        self.norm_output = c.norm_output
        self.lm_head = _W(c.draft_vocab_size, c.hidden_size, **kw)
        self.register_buffer("t2d", torch.ones(c.vocab_size, dtype=torch.bool, device=device))
        self.register_buffer("d2t", torch.zeros(c.draft_vocab_size, dtype=torch.int64, device=device))
        self.freeze_embedding()

    # -- the reference ABC's helpers (modeling/draft/base.py:128-206) -------------------------
    def freeze_embedding(self) -> None:
        self.embed_tokens.weight.requires_grad = False

    def load_embedding_weight(self, weight: torch.Tensor) -> None:
        with torch.no_grad():
            self.embed_tokens.weight.copy_(weight.to(self.embed_tokens.weight.dtype))

    def load_vocab_mapping_tensors(self, t2d: torch.Tensor, d2t: torch.Tensor) -> None:
        with torch.no_grad():
            self.t2d.copy_(t2d.to(torch.bool))
            self.d2t.copy_(d2t.to(torch.int64))


# order of the flat buffer = order in which the backward sweep finishes the gradients, so the DP
# all-reduce of bucket i overlaps the wgrad GEMM of bucket i+1 (largest first)
FLAT_ORDER = [
    "lm_head.weight",
    "midlayer.mlp.gate_proj.weight", "midlayer.mlp.up_proj.weight",
    "midlayer.mlp.down_proj.weight",
    "midlayer.self_attn.q_proj.weight", "midlayer.self_attn.k_proj.weight", "midlayer.self_attn.v_proj.weight",
    "midlayer.self_attn.o_proj.weight",
    "fc.weight",
    "midlayer.hidden_norm.weight", "midlayer.input_layernorm.weight", "midlayer.post_attention_layernorm.weight",
    "norm.weight", "fc_norm.0.weight", "fc_norm.1.weight", "fc_norm.2.weight",
]


class FlatParams:
    """One contiguous bf16 parameter buffer + flat grad (bf16) of the same layout."""

    def __init__(self, model: LlamaForCausalLMEagle3):
        named = {n: p for n, p in model.named_parameters() if p.requires_grad}
        order = [n for n in FLAT_ORDER if n in named]
        extra = [n for n in named if n not in order]
        if extra:
            raise ValueError(f"unexpected trainable parameters {extra}")
        dev, dt = named[order[0]].device, named[order[0]].dtype
        if dt != torch.bfloat16:
            raise TypeError("the HIP training path stores parameters in bf16 (fp32 masters live in the optimizer)")
        self.names: List[str] = order
        self.slices: Dict[str, Tuple[int, int]] = {}
        off = 0
        for n in order:
            k = named[n].numel()
            self.slices[n] = (off, off + k)
            off += (k + 7) // 8 * 8
        self.numel = off
        self.data = torch.zeros(off, dtype=dt, device=dev)
        self.grad = torch.zeros(off, dtype=dt, device=dev)
        self.params: Dict[str, nn.Parameter] = {}
        with torch.no_grad():
            for n in order:
                p = named[n]
                lo, hi = self.slices[n]
                view = self.data[lo:hi].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.grad[lo:hi].view(p.shape)
                self.params[n] = p
        # parameters in the reference's ``model.parameters()`` order (optimizer state index order)
        self.module_order: List[str] = [n for n, p in model.named_parameters() if p.requires_grad]

    def view(self, name: str) -> torch.Tensor:
        return self.params[name].data

    def gview(self, name: str) -> torch.Tensor:
        lo, hi = self.slices[name]
        return self.grad[lo:hi].view(self.params[name].shape)

    def fused(self, first: str, last: str, rows: int, cols: int, grad: bool = False) -> torch.Tensor:
        """[rows, cols] view over the adjacent parameters first..last (q|k|v, gate|up)"""
        lo, hi = self.slices[first][0], self.slices[last][1]
        assert hi - lo == rows * cols, "fused parameters must be adjacent and unpadded"
        return (self.grad if grad else self.data)[lo:hi].view(rows, cols)
