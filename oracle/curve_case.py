"""Inputs of the real-dimension loss-curve fixture (tests/golden/loss_curve_cfg1_realdims.pt), regenerated from a seed.

Test infrastructure: imported by oracle/gen_curve_realdims.py (build container, runs the reference trainer on these inputs) and by
tests/test_loss_curve_realdims.py (runs the HIP path on the same inputs).  Integer draws only (oracle/seeded_case.py): bit-identical on
any host; the fixture's checksums are verified before every comparison.
"""
import os

import torch

from oracle import seeded_case as SC

CFG = dict(H=896, Ht=896, I=4864, nh=14, nkv=2, hd=64, Vt=151936, Vd=16000, B=1, S=8, lengths=[8], seed=201,      # (B, S: make_case's unused batch)
           steps=12, batch_size=1, ttt=7, lr=1e-3, max_grad_norm=0.5, warmup_ratio=0.1, n_files=4, max_len=256, weight_std=0.04, num_epochs=4)



def make_inputs(c):
    """-> params (bf16), embed, head_w, t2d, d2t, raws (list of the feature-file dicts), lengths"""
    params, embed, head_w, t2d, d2t, _ = SC.make_case(c, c["seed"])
    g = torch.Generator().manual_seed(c["seed"] + 1)
    for k in list(params):                       # livelier than the 0.02 init, so that 12 steps move the loss visibly
        if params[k].dim() == 2:
            params[k] = SC.ih_normal(tuple(params[k].shape), g, c["weight_std"])
    lengths = [int(x) for x in torch.randint(150, 300, (c["n_files"],), generator=g)]
    raws = []
    for L in lengths:
        prompt = int(torch.randint(5, 40, (1,), generator=g))
        lm = torch.ones(L, dtype=torch.long)
        lm[:prompt] = 0
        raws.append({"input_ids": torch.randint(0, c["Vt"], (L,), generator=g), "loss_mask": lm,
                     "hidden_state": SC.ih_normal((1, L, c["Ht"]), g, 1.0), "aux_hidden_state": SC.ih_normal((1, L, 3 * c["Ht"]), g, 1.0)})
    return params, embed, head_w, t2d, d2t, raws, lengths


def input_checksums(params, embed, head_w, t2d, d2t, raws):
    out = {"param:" + k: SC.checksum(v) for k, v in params.items()}
    out.update(embed=SC.checksum(embed), head_w=SC.checksum(head_w), t2d=SC.checksum(t2d), d2t=SC.checksum(d2t))
    for i, r in enumerate(raws):
        out[f"file{i}"] = "".join(SC.checksum(r[k]) for k in ("input_ids", "loss_mask", "hidden_state", "aux_hidden_state"))
    return out


def write_files(d, raws):
    os.makedirs(d, exist_ok=True)
    paths = []
    for i, raw in enumerate(raws):
        p = os.path.join(d, f"{i:04d}.ckpt")
        torch.save(raw, p)
        paths.append(p)
    return paths


def weight_summary(sd, seed):
    out = {}
    for k, v in sd.items():
        if v.dtype == torch.bfloat16 and "embed" not in k:
            idx = SC.grad_probe_indices(v.shape, seed + 9, n=2048)
            out[k] = dict(fro=float(v.double().norm()), samples=v.flatten()[idx].float().clone())
    return out
