"""Shared harness of the full-dimension parity tests: one EAGLE3 micro-step through the C-ABI on the GPU
vs the pinned oracle (oracle/eagle3_oracle.py) run ON THE SAME GPU in fp32 as the checker -- the CPU
cannot finish Llama-3-8B x seq 2048 in test time, the restatement is device-agnostic torch.

The checker runs the DRAFT in fp32 and the frozen teacher head in bf16 (as the reference trains: ``TargetHead`` is a
bf16 module, so bf16-rounded teacher logits and the argmax near-ties they create belong to the reference's semantics;
a first version with an fp32 teacher disagreed on 1-3 % of the argmax ids for the HIP path AND for the reference's own
bf16 run alike, which then dominated every gradient comparison at 8-17 %).

What is compared (BASELINE.json north_star tolerance for the bf16 path: 2e-2):
  * plosses / loss / acceptance rates / accuracy
  * acc_denoms: bit-exact; teacher argmax ids and position mask: bit-exact (== 100 %) at these real dimensions -- both sides
    round the SAME fp32-accumulated bf16 GEMM on the same GPU (measured 100 % in every case since round 2; the >= 99.5 % slack
    survives only in tests/test_configs.py's small cases, where the other side is a CPU bf16 GEMM with another summation order)
  * every parameter gradient: max-abs error relative to the tensor's max-abs, and relative Frobenius error
The same numbers are taken for the oracle run entirely in bf16 (= what the reference itself produces at that
precision with torch/hipBLASLt kernels), the yardstick each HIP number is printed beside.
Results go to gpurun_out/parity_<name>.json (committed under profiles/).
"""
import json
import os
import time

import torch

from oracle import eagle3_oracle as O
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg_kw(c):
    return dict(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                max_position_embeddings=c.get("max_pos", 2048), rms_norm_eps=c.get("eps", 1e-6), fc_norm=c.get("fc_norm", False),
                rope_theta=c.get("rope_theta", 10000.0), rope_scaling=c.get("rope_scaling"))


def make_case(c, seed=1):
    """bf16-representable weights / inputs shared by both sides (CPU tensors)."""
    oc = O.DraftConfig(**_cfg_kw(c))
    bf = torch.bfloat16
    params = {k: v.to(bf) for k, v in O.init_params(oc, seed=seed).items()}
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in params.items():  # non-trivial norm weights
        if v.dim() == 1:
            params[k] = (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
    embed = (torch.randn(c["Vt"], c["H"], generator=g) * 0.05).to(bf)
    head_w = (torch.randn(c["Vt"], c["Ht"], generator=g) * 0.05).to(bf)
    t2d, d2t = O.make_vocab_mapping(c["Vt"], c["Vd"], seed=seed + 2)
    batch = O.make_batch(oc, c["B"], c["S"], seed=seed + 3, dtype=bf, lengths=c.get("lengths"))
    if c.get("prompt"):
        batch["loss_mask"][:, :c["prompt"]] = 0          # a prompt region without loss
    return oc, params, embed, head_w, t2d, d2t, batch


def run_oracle(oc, params, embed, head_w, t2d, d2t, batch, ttt, dev, dtype):
    """draft in ``dtype``; the TEACHER always in bf16: the frozen ``TargetHead`` is a bf16 module in the reference's
    training runs (target_head.py:100-101 on bf16 features), so bf16-rounded teacher logits -- and the argmax ties they
    create -- are part of the reference's semantics, not an approximation of the HIP path."""
    p = {k: v.to(dev).to(dtype).requires_grad_(True) for k, v in params.items()}
    out = O.eagle3_forward(p, oc, embed_weight=embed.to(dev).to(dtype), target_head_weight=head_w.to(dev),
                           t2d=t2d.to(dev), d2t=d2t.to(dev), input_ids=batch["input_ids"].to(dev),
                           attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"].to(dev),
                           hidden_state=batch["hidden_state"].to(dev).to(dtype), target_hidden=batch["target"].to(dev),
                           ttt_length=ttt)
    out.loss.backward()
    res = dict(plosses=torch.stack([x.detach().float() for x in out.plosses]).cpu(), loss=out.loss.detach().float().cpu(),
               acces=torch.stack(out.acces).float().cpu(), acceptance=torch.stack(out.acceptance_rates).float().cpu(),
               acc_denoms=torch.stack(out.acc_denoms).float().cpu(), ids=out.target_token_ids.cpu(),
               pos_mask=out.position_mask.squeeze(-1).int().cpu(),
               grads={k: (v.grad.detach().float().cpu() if v.grad is not None else torch.zeros(v.shape)) for k, v in p.items()})
    del p, out
    torch.cuda.empty_cache()
    return res


def run_hip(c, params, embed, head_w, t2d, d2t, batch, ttt, dev):
    model = LlamaForCausalLMEagle3(DraftConfig(**_cfg_kw(c)), device=dev)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=ttt).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(dev)))
    out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                             loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"].to(dev),
                                             target=batch["target"].to(dev)), {"target_repr": "hidden_state"}))
    out.loss.backward()
    torch.cuda.synchronize()
    named = dict(model.named_parameters())
    res = dict(plosses=torch.stack(out.metrics["plosses"]).float().cpu(), loss=out.loss.detach().float().cpu(),
               acces=torch.stack(out.metrics["acces"]).float().cpu(),
               acceptance=torch.stack(out.metrics["acceptance_rates"]).float().cpu(),
               acc_denoms=torch.stack(out.metrics["acc_denoms"]).float().cpu(),
               ids=eagle.last_artifacts["target_token_ids"].cpu(), pos_mask=eagle.last_artifacts["position_mask"].int().cpu(),
               teacher_compacted=eagle.engine._teacher_compacted,
               grads={k: named[k].grad.float().cpu() for k in params})
    del model, eagle, strat, out
    torch.cuda.empty_cache()
    return res


def grad_errors(got, ref):
    rows = {}
    for k, g in ref.items():
        d = got[k] - g
        rows[k] = dict(max_rel=float(d.abs().max() / g.abs().max().clamp_min(1e-12)),
                       fro_rel=float(d.norm() / g.norm().clamp_min(1e-12)))
    return rows


def compare(name, c, *, loss_tol=5e-3, ids_min=1.0, grad_cap=6e-2, yardstick_factor=1.1):
    """runs both sides, writes the report, asserts the bars; returns the report dict."""
    dev = torch.device("cuda", 0)
    ttt = c["ttt"]
    case = make_case(c)
    t0 = time.time()
    ref = run_oracle(*case, ttt, dev, torch.float32)
    t_ref = time.time() - t0
    hip = run_hip(c, *case[1:], ttt, dev)
    rep = dict(case=name, dims={k: v for k, v in c.items()}, oracle_fp32_seconds=round(t_ref, 1))
    # sparse loss mask (ragged lengths): the engine ran the teacher only on positions that carry a loss (loss-row compaction) -- target
    # ids exist there, every other position holds id 0 and position mask 0 (which is what the reference's position mask has there too)
    on = case[6]["loss_mask"].bool() if hip["teacher_compacted"] else torch.ones_like(case[6]["loss_mask"], dtype=torch.bool)
    assert int(hip["ids"][~on].abs().sum()) == 0
    rep["teacher_compacted"], rep["ids_compared_on"] = bool(hip["teacher_compacted"]), float(on.float().mean())
    rep["hip_vs_fp32"] = dict(
        plosses=hip["plosses"].tolist(), plosses_ref=ref["plosses"].tolist(),
        ploss_max_abs=float((hip["plosses"] - ref["plosses"]).abs().max()), loss=float(hip["loss"]), loss_ref=float(ref["loss"]),
        acceptance_max_abs=float((hip["acceptance"] - ref["acceptance"]).abs().max()),
        acc_max_abs=float((hip["acces"] - ref["acces"]).abs().max()),
        ids_agree=float((hip["ids"] == ref["ids"])[on].float().mean()),
        pos_mask_agree=float((hip["pos_mask"] == ref["pos_mask"]).float().mean()),
        grads=grad_errors(hip["grads"], ref["grads"]))
    if True:
        bf = run_oracle(*case, ttt, dev, torch.bfloat16)
        rep["oracle_bf16_vs_fp32"] = dict(
            ploss_max_abs=float((bf["plosses"] - ref["plosses"]).abs().max()),
            ids_agree=float((bf["ids"] == ref["ids"]).float().mean()), grads=grad_errors(bf["grads"], ref["grads"]))
        rep["hip_vs_oracle_bf16"] = dict(ids_agree=float((hip["ids"] == bf["ids"])[on].float().mean()),
                                         pos_mask_agree=float((hip["pos_mask"] == bf["pos_mask"]).float().mean()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"parity_{name}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    h = rep["hip_vs_fp32"]
    worst = {k: (round(v["max_rel"], 4), round(v["fro_rel"], 4)) for k, v in h["grads"].items()}
    print(f"\n[parity {name}] ploss max|d| {h['ploss_max_abs']:.2e}  ids {h['ids_agree']:.5f}  grads (max_rel, fro_rel): {worst}")
    # ---- the bars
    assert torch.equal(hip["acc_denoms"], ref["acc_denoms"]), "acc_denoms must be bit-exact"
    torch.testing.assert_close(hip["plosses"], ref["plosses"], rtol=loss_tol, atol=loss_tol)
    torch.testing.assert_close(hip["loss"], ref["loss"], rtol=loss_tol, atol=loss_tol)
    torch.testing.assert_close(hip["acceptance"], ref["acceptance"], rtol=loss_tol, atol=loss_tol)
    torch.testing.assert_close(hip["acces"], ref["acces"], rtol=0, atol=loss_tol)
    assert h["ids_agree"] >= ids_min, h["ids_agree"]
    # position_mask = t2d[ids] * loss_mask must be self-consistent bit-exactly with the HIP path's own ids
    t2d = case[4]
    lm = torch.cat((case[6]["loss_mask"],), 0)
    assert torch.equal(hip["pos_mask"], t2d[hip["ids"]].int() * lm.int())
    # gradients: the bf16 rounding error of a step grows with the contraction depths (measured r2, Frobenius-relative:
    # 1.4 % at H 2048, 2.2 % at H 4096, 4.6 % at H 7168 / I 40960), so the bar is relative to the yardstick: per tensor the
    # HIP path must be no further from the fp32 truth than 1.1x what the reference's own bf16 run is (measured: 0.65-0.75x),
    # with an absolute cap of 6e-2 on both measures
    yard = rep["oracle_bf16_vs_fp32"]["grads"]
    bad = {}
    for k, v in h["grads"].items():
        if v["fro_rel"] > max(2e-2, yardstick_factor * yard[k]["fro_rel"]) or v["fro_rel"] > grad_cap or v["max_rel"] > max(
                2e-2, yardstick_factor * yard[k]["max_rel"], 2 * yard[k]["fro_rel"]) or v["max_rel"] > 1.5 * grad_cap:
            bad[k] = (v, yard[k])
    assert not bad, bad
    return rep
