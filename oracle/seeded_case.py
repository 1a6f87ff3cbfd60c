"""Seeded inputs that are BIT-IDENTICAL on every host.  TEST INFRASTRUCTURE ONLY (oracle/ header rule: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import anything from here).

The real-dimension goldens (``oracle/gen_golden_realdims.py`` -> ``tests/golden/realdims_*.pt``) cannot carry their inputs:
a Llama-3-8B draft, its frozen embedding and the target head are 3 GB.  The fixture holds the SEED, a checksum of every
regenerated tensor and the reference's outputs; the inputs are regenerated on the GPU box from the seed.  ``torch.randn`` is
not fit for that (its Box-Muller uses vectorised log / cos whose last ulp depends on the host's SIMD width), so every value
here is built from ``torch.randint`` draws (mt19937 integers: identical everywhere), integer sums and ONE exact fp32
multiplication: an Irwin-Hall(4) variate, i.e. bell-shaped, standard deviation ``std``.
"""
import hashlib

import torch

_IH_STD = (4 * (256 * 256 - 1) / 12.0) ** 0.5          # std of the sum of four uniform integers on 0..255


def ih_normal(shape, gen, std=1.0, dtype=torch.bfloat16):
    """sum of four uniform bytes, centred, scaled by one fp32 constant, rounded once to ``dtype``: deterministic IEEE arithmetic"""
    n = 1
    for s in shape:
        n *= s
    acc = torch.zeros(n, dtype=torch.int16)
    for _ in range(4):
        acc += torch.randint(0, 256, (n,), generator=gen, dtype=torch.int16)
    acc -= 510
    return (acc.to(torch.float32) * (std / _IH_STD)).to(dtype).view(shape)


def checksum(t: torch.Tensor) -> str:
    """sha256 of the tensor's bytes (bf16 viewed as int16)"""
    t = t.detach().contiguous().cpu()
    if t.dtype == torch.bfloat16:
        t = t.view(torch.int16)
    if t.dtype == torch.bool:
        t = t.to(torch.uint8)
    return hashlib.sha256(t.numpy().tobytes()).hexdigest()[:16]


def param_shapes(c):
    """trainable parameters of the draft in the reference's state-dict order (llama3_eagle.py:1653-1700)"""
    H, Ht, I, nh, nkv, hd, Vd = c["H"], c["Ht"], c["I"], c["nh"], c["nkv"], c["hd"], c["Vd"]
    out = {}
    out["midlayer.self_attn.q_proj.weight"] = (nh * hd, 2 * H)
    out["midlayer.self_attn.k_proj.weight"] = (nkv * hd, 2 * H)
    out["midlayer.self_attn.v_proj.weight"] = (nkv * hd, 2 * H)
    out["midlayer.self_attn.o_proj.weight"] = (H, nh * hd)
    out["midlayer.mlp.gate_proj.weight"] = (I, H)
    out["midlayer.mlp.up_proj.weight"] = (I, H)
    out["midlayer.mlp.down_proj.weight"] = (H, I)
    out["midlayer.hidden_norm.weight"] = (H,)
    out["midlayer.input_layernorm.weight"] = (H,)
    out["midlayer.post_attention_layernorm.weight"] = (H,)
    out["fc.weight"] = (H, 3 * Ht)
    if c.get("fc_norm"):
        for i in range(3):
            out[f"fc_norm.{i}.weight"] = (Ht,)
    out["norm.weight"] = (H,)
    out["lm_head.weight"] = (Vd, H)
    return out


def make_case(c, seed):
    """-> params (bf16), embed, head_w, t2d, d2t, batch -- all CPU tensors, bf16-representable.
    ``c``: H, Ht, I, nh, nkv, hd, Vt, Vd, B, S, lengths, prompt (leading positions without loss)."""
    g = torch.Generator().manual_seed(seed)
    params = {}
    for k, shp in param_shapes(c).items():
        if len(shp) == 1:
            params[k] = (1.0 + ih_normal(shp, g, 0.1, torch.float32)).to(torch.bfloat16)
        else:
            params[k] = ih_normal(shp, g, 0.02)
    embed = ih_normal((c["Vt"], c["H"]), g, 0.05)
    head_w = ih_normal((c["Vt"], c["Ht"]), g, 0.05)
    # vocabulary map: Vd distinct target ids, ascending (training/vocab_mapping.py: the most frequent ids, sorted)
    perm = torch.randperm(c["Vt"], generator=g)
    keep = perm[: c["Vd"]].sort().values
    t2d = torch.zeros(c["Vt"], dtype=torch.bool)
    t2d[keep] = True
    # real vocabulary maps keep the FREQUENT tokens: the teacher's argmax lies inside the draft vocabulary for most positions.  Raise
    # the head rows of the kept ids (x 1.25: exact in fp32, one more deterministic rounding) so that most rows carry a loss
    head_w[keep] = (head_w[keep].float() * 1.25).to(torch.bfloat16)
    d2t = keep - torch.arange(c["Vd"])
    B, S = c["B"], c["S"]
    input_ids = torch.randint(0, c["Vt"], (B, S), generator=g)
    target = ih_normal((B, S, c["Ht"]), g, 1.0)
    hidden = ih_normal((B, S, 3 * c["Ht"]), g, 1.0)
    loss_mask = torch.ones(B, S, dtype=torch.long)
    attention_mask = torch.ones(B, S, dtype=torch.long)
    for b, L in enumerate(c.get("lengths") or [S] * B):
        loss_mask[b, L - 1:] = 0
        attention_mask[b, L:] = 0
        input_ids[b, L:] = 0
        target[b, L:] = 0
        hidden[b, L:] = 0
    if c.get("prompt"):
        loss_mask[:, : c["prompt"]] = 0
    batch = dict(input_ids=input_ids, attention_mask=attention_mask, loss_mask=loss_mask, hidden_state=hidden, target=target)
    return params, embed, head_w, t2d, d2t, batch


def case_checksums(params, embed, head_w, t2d, d2t, batch):
    out = {"param:" + k: checksum(v) for k, v in params.items()}
    out.update(embed=checksum(embed), head_w=checksum(head_w), t2d=checksum(t2d), d2t=checksum(d2t))
    out.update({"batch:" + k: checksum(v) for k, v in batch.items()})
    return out


def grad_probe_indices(shape, seed, n=4096):
    """flat indices of the sampled gradient entries a fixture stores (with replacement; deterministic)"""
    numel = 1
    for s in shape:
        numel *= s
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, numel, (min(n, numel),), generator=g)


def grad_summary(grad: torch.Tensor, idx: torch.Tensor):
    """what a fixture keeps of one gradient tensor: Frobenius norm, max |g|, row and column sums (fp64 accumulate), samples"""
    g = grad.detach().double().cpu()
    out = dict(fro=float(g.norm()), amax=float(g.abs().max()), samples=g.flatten()[idx].float())
    if g.dim() == 2:
        out["rowsum"] = g.sum(1).float()
        out["colsum"] = g.sum(0).float()
    return out
