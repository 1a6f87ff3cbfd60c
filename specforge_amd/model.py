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


def rope_tables(cfg: DraftConfig, dtype=torch.bfloat16, n_pos: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin caches [n_pos (default max_pos+20), head_dim]: fp32 then cast to the activation dtype
    (llama3_eagle.py:218-312).  Variants: llama3 frequency smoothing (235-276), linear position scaling
    (315-344), dynamic NTK (347-386: the cache is pre-built for max_pos+20 > max_pos positions, so the base is
    already rescaled for that length), yarn (430-540: blended frequencies and an amplitude factor on cos/sin).
    mrope (llama3_eagle.py:389-427, 145-182) uses these plain tables too: the engine gathers one row per position AXIS
    and interleaves the head's rotary channels by ``mrope_section`` (Eagle3Engine._rope_rows).
    ``n_pos`` > max_pos+20: a table REBUILT for a longer sequence (``_set_cos_sin_cache(seq_len)``, llama3_eagle.py:303-306) --
    the same rows for every variant but dynamic NTK, whose base is a function of that length (llama3_eagle.py:362-371)."""
    dim = cfg.head_dim
    n_pos = cfg.max_position_embeddings + 20 if n_pos is None else int(n_pos)
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
    elif rtype not in (None, "default", "linear", "mrope"):   # mrope: plain tables; the engine combines the three axes
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


class Eagle3DraftMethods:
    """The ``Eagle3DraftModel`` seam (specforge/modeling/draft/base.py:38-206) on the C-ABI: parameter construction
    under the reference's names, the four abstract methods (``embed_input_ids`` / ``project_hidden_states`` /
    ``backbone`` / ``compute_logits``; base.py:45-109, llama3_eagle.py:1702-1798), ``load_embedding(path)`` (base.py:135)
    and ``load_vocab_mapping(path)`` (base.py:193).  A mixin so the same code serves the standalone container below
    (``nn.Module``) and the ``@register_draft`` class that ``specforge_amd.reference_plugin`` derives from the
    reference's own ``Eagle3DraftModel`` when that package is importable.

    The four methods are forward-only (evaluation, export checks, the reference's assembly code): each is one or a few
    C-ABI calls on the current stream with no autograd graph.  TRAINING never goes through them -- the whole TTT
    micro-step runs in ``specforge_amd.engine.Eagle3Engine`` (one level up, the ``OnlineEagle3Model`` seam), which is
    what keeps activations in persistent stashes and the weight gradients deferred."""

    def _build_parameters(self, c: DraftConfig, attention_backend: str, dtype, device) -> None:
        """registration order of the reference (llama3_eagle.py:1658-1700): embed_tokens, midlayer, fc, [fc_norm], norm,
        lm_head, t2d, d2t"""
        self.draft_config = c
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
        self.vocab_mapping_loaded = False
        self._rope = None

    # -- helpers of the reference ABC (modeling/draft/base.py:128-206) ------------------------
    def freeze_embedding(self) -> None:
        self.embed_tokens.weight.requires_grad = False

    def load_embedding_weight(self, weight: torch.Tensor) -> None:
        with torch.no_grad():
            self.embed_tokens.weight.copy_(weight.to(self.embed_tokens.weight.dtype))

    def load_vocab_mapping_tensors(self, t2d: torch.Tensor, d2t: torch.Tensor) -> None:
        with torch.no_grad():
            self.t2d.copy_(t2d.to(torch.bool))
            self.d2t.copy_(d2t.to(torch.int64))
        self.vocab_mapping_loaded = True

    @torch.no_grad()
    def load_embedding(self, model_path: str, embedding_key: str = "model.embed_tokens.weight") -> None:
        """base.py:135-191: copy the target's embedding table from a local model directory (``*.index.json`` ->
        shard, else ``model.safetensors`` / ``pytorch_model.bin``).  Hub ids are not resolved here (no network on the
        training nodes): pass the local snapshot directory."""
        import glob
        import json
        import os

        if not os.path.isdir(model_path):
            raise FileNotFoundError(f"load_embedding: {model_path!r} is not a local model directory")
        index = glob.glob(os.path.join(model_path, "*.index.json"))
        if len(index) > 1:
            raise FileNotFoundError(f"Multiple index.json files found in {model_path}")
        if index:
            with open(index[0]) as f:
                ckpt = os.path.join(model_path, json.load(f)["weight_map"][embedding_key])
        elif os.path.exists(os.path.join(model_path, "model.safetensors")):
            ckpt = os.path.join(model_path, "model.safetensors")
        elif os.path.exists(os.path.join(model_path, "pytorch_model.bin")):
            ckpt = os.path.join(model_path, "pytorch_model.bin")
        else:
            raise FileNotFoundError(f"No index.json, model.safetensors or pytorch_model.bin found in {model_path}")
        if ckpt.endswith(".safetensors"):
            from safetensors import safe_open

            with safe_open(ckpt, framework="pt") as f:
                w = f.get_tensor(embedding_key)
        else:
            w = torch.load(ckpt, map_location="cpu")[embedding_key]
        self.load_embedding_weight(w)

    def load_vocab_mapping(self, file_path: str) -> None:
        """base.py:193-206: ``torch.save({"t2d": bool[Vt], "d2t": int64[Vd]})``"""
        m = torch.load(file_path, map_location="cpu")
        self.load_vocab_mapping_tensors(m["t2d"], m["d2t"])

    # -- the four abstract methods, forward-only on the C-ABI ---------------------------------------
    def _w(self, name: str) -> torch.Tensor:
        return self.get_parameter(name).data

    @torch.no_grad()
    def embed_input_ids(self, input_ids: torch.Tensor) -> torch.Tensor:
        """llama3_eagle.py:1759-1760: frozen embedding gather, [B,S] -> [B,S,H]"""
        return self.embed_tokens.weight.data[input_ids]

    @torch.no_grad()
    def project_hidden_states(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """llama3_eagle.py:1762-1770: [B,S,3Ht] -> (3x fc_norm) -> fc -> [B,S,H]"""
        from . import ops

        c = self.draft_config
        B, S, W = hidden_states.shape
        x = hidden_states.reshape(B * S, W).to(torch.bfloat16)
        if not x.is_contiguous():
            x = x.contiguous()
        if c.fc_norm:
            Ht = c.target_hidden_size
            xn = torch.empty_like(x)
            for i in range(3):
                ops.rmsnorm_fwd(x[:, i * Ht:(i + 1) * Ht], self.fc_norm[i].weight.data, c.rms_norm_eps,
                                xn[:, i * Ht:(i + 1) * Ht], None)
            x = xn
        out = torch.empty(B * S, c.hidden_size, dtype=torch.bfloat16, device=x.device)
        ops.gemm_nt(x, self.fc.weight.data, out)
        return out.view(B, S, c.hidden_size)

    @torch.no_grad()
    def compute_logits(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """llama3_eagle.py:1772-1777: lm_head(norm(h)) (norm skipped when ``norm_output`` is False)"""
        from . import ops

        c = self.draft_config
        B, S, H = hidden_states.shape
        x = hidden_states.reshape(B * S, H).contiguous()
        if c.norm_output:
            xn = torch.empty_like(x)
            ops.rmsnorm_fwd(x, self.norm.weight.data, c.rms_norm_eps, xn, None)
            x = xn
        out = torch.empty(B * S, c.draft_vocab_size, dtype=torch.bfloat16, device=x.device)
        ops.gemm_nt(x, self.lm_head.weight.data, out)
        return out.view(B, S, c.draft_vocab_size)

    @torch.no_grad()
    def backbone(self, input_embeds, hidden_states, cache_hidden, attention_mask, position_ids, past_key_values=None,
                 use_cache: bool = True) -> torch.Tensor:
        """One TTT step of ``LlamaDecoderLayer`` (llama3_eagle.py:1598-1650; attention cache branch 717-778):
        ``cache_hidden`` is the caller-owned ``[[k...],[v...]]`` list of one micro-step -- this step's K/V are appended
        (as [B*S, nkv*hd] bf16 views) and every earlier step contributes its diagonal.  ``attention_mask`` is the
        [B,S] padding mask or the 4-D additive mask of ``prepare_decoder_attention_mask`` (its last query row tells the
        number of valid keys: the collator right-pads)."""
        from . import ops

        if past_key_values is not None:
            raise NotImplementedError("past_key_values is unused by EAGLE3 training (eagle3/model.py:262)")
        c = self.draft_config
        B, S, H = hidden_states.shape
        N = B * S
        dev = hidden_states.device
        nh, nkv, hd, eps = c.num_attention_heads, c.num_key_value_heads, c.head_dim, c.rms_norm_eps
        if attention_mask is None:
            kv_len = torch.full((B,), S, dtype=torch.int32, device=dev)
        elif attention_mask.dim() == 2:
            kv_len = attention_mask.to(dev).sum(dim=1).to(torch.int32)
        else:
            kv_len = (attention_mask[:, 0, -1, :] == 0).sum(dim=-1).to(torch.int32).to(dev)
        lck = len(cache_hidden[0])
        if self._rope is None or self._rope[0].device != dev:
            cos, sin = rope_tables(c, torch.bfloat16)
            self._rope = (cos.to(dev), sin.to(dev))
            self._rope_len = c.max_position_embeddings + 20
        if S + lck > self._rope_len:
            # the reference's rotary module rebuilds its cache for seq_len = q_len + lck when that exceeds it (llama3_eagle.py:303-306,
            # 733) and keeps the rebuilt one: dynamic NTK re-derives its base from exactly that length, every other variant extends
            rs = c.rope_scaling or {}
            dyn = rs.get("rope_type", rs.get("type")) == "dynamic"
            n = S + lck if dyn else max(S + lck, self._rope[0].shape[0] + 1024)
            cos, sin = rope_tables(c, torch.bfloat16, n_pos=n)
            self._rope = (cos.to(dev), sin.to(dev))
            self._rope_len = n
        cos, sin = self._rope
        pos = (torch.arange(S, device=dev).repeat(B) if position_ids is None
               else position_ids.to(dev).expand(B, S).reshape(-1)).to(torch.int64).contiguous()
        ml = self.midlayer
        bf = torch.bfloat16
        e = lambda *shape, dtype=bf: torch.empty(*shape, dtype=dtype, device=dev)
        h = hidden_states.reshape(N, H).to(bf).contiguous()
        x = e(N, 2 * H)            # cat(input_layernorm(embeds), hidden_norm(hidden))  (1625-1630)
        ops.rmsnorm_fwd(input_embeds.reshape(N, H).to(bf).contiguous(), ml.input_layernorm.weight.data, eps, x[:, :H], None)
        ops.rmsnorm_fwd(h, ml.hidden_norm.weight.data, eps, x[:, H:], None)
        QW = (nh + 2 * nkv) * hd
        qkv = e(N, QW)
        at = ml.self_attn
        ops.gemm_nt(x, at.q_proj.weight.data, qkv[:, :nh * hd])
        ops.gemm_nt(x, at.k_proj.weight.data, qkv[:, nh * hd:(nh + nkv) * hd])
        ops.gemm_nt(x, at.v_proj.weight.data, qkv[:, (nh + nkv) * hd:])
        ops.rope_(qkv, nh + nkv, hd, cos, sin, pos, lck)
        cache_hidden[0].append(qkv[:, nh * hd:(nh + nkv) * hd])
        cache_hidden[1].append(qkv[:, (nh + nkv) * hd:])
        ks, vs = cache_hidden
        o, lse = e(N, nh * hd), e(B, nh, S, dtype=torch.float32)
        ops.attn_fwd(qkv[:, :nh * hd], ks[0], vs[0], ks[1:], vs[1:], kv_len, o, lse, B=B, S=S, nh=nh, nkv=nkv, hd=hd,
                     scale=1.0 / math.sqrt(hd))
        h1 = e(N, H)
        ops.gemm_nt(o, at.o_proj.weight.data, h1, residual=h)
        pn = e(N, H)
        ops.rmsnorm_fwd(h1, ml.post_attention_layernorm.weight.data, eps, pn, None)
        I = c.intermediate_size
        gu = e(N, 2 * I)
        ops.gemm_nt(pn, ml.mlp.gate_proj.weight.data, gu[:, :I])
        ops.gemm_nt(pn, ml.mlp.up_proj.weight.data, gu[:, I:])
        act = e(N, I)
        ops.swiglu_fwd(gu, act)
        out = e(N, H)
        ops.gemm_nt(act, ml.mlp.down_proj.weight.data, out, residual=h1)
        return out.view(B, S, H)


class LlamaForCausalLMEagle3(Eagle3DraftMethods, nn.Module):
    """Standalone parameter container with the reference's names and registration order."""

    def __init__(self, config, attention_backend: str = "hip", dtype=torch.bfloat16, device=None):
        super().__init__()
        c = config if isinstance(config, DraftConfig) else DraftConfig.from_hf(config)
        self.config = c
        self._build_parameters(c, attention_backend, dtype, device)
        self.freeze_embedding()


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
    "fc_norm.0.weight", "fc_norm.1.weight", "fc_norm.2.weight",
    "norm.weight",   # last: with norm_output=False it receives no gradient and is left out of the optimizer range
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
        # parameters the path never gives a gradient: with norm_output=False the final `norm` is not applied
        # (llama3_eagle.py:1772-1777), so the reference leaves norm.weight.grad None and its optimizer skips the parameter
        # (optimizer.py:139-142).  Their .grad stays None here as well -- a foreign optimizer would otherwise apply weight
        # decay to it and build Adam state for it.
        cfg = getattr(model, "draft_config", None) or getattr(model, "config", None)
        self.gradless = {"norm.weight"} if (cfg is not None and not getattr(cfg, "norm_output", True)) else set()
        with torch.no_grad():
            for n in order:
                p = named[n]
                lo, hi = self.slices[n]
                view = self.data[lo:hi].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None if n in self.gradless else self.grad[lo:hi].view(p.shape)
                self.params[n] = p
        # parameters in the reference's ``model.parameters()`` order (optimizer state index order)
        self.module_order: List[str] = [n for n, p in model.named_parameters() if p.requires_grad]

    def view(self, name: str) -> torch.Tensor:
        return self.params[name].data

    def realias_grads(self) -> None:
        for n, p in self.params.items():
            if n in self.gradless:
                p.grad = None
                continue
            if p.grad is None or p.grad.data_ptr() != self.grad[self.slices[n][0]:].data_ptr():
                lo, hi = self.slices[n]
                p.grad = self.grad[lo:hi].view(p.shape)

    def gview(self, name: str) -> torch.Tensor:
        lo, hi = self.slices[name]
        return self.grad[lo:hi].view(self.params[name].shape)

    def fused(self, first: str, last: str, rows: int, cols: int, grad: bool = False) -> torch.Tensor:
        """[rows, cols] view over the adjacent parameters first..last (q|k|v, gate|up)"""
        lo, hi = self.slices[first][0], self.slices[last][1]
        assert hi - lo == rows * cols, "fused parameters must be adjacent and unpadded"
        return (self.grad if grad else self.data)[lo:hi].view(rows, cols)
