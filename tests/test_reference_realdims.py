"""The HIP path against the REFERENCE ITSELF at real model dimensions -- no port in between.

``tests/golden/realdims_*.pt`` hold what the reference's own ``LlamaForCausalLMEagle3`` (sdpa) + ``OnlineEagle3Model`` produced in
the build container (``oracle/gen_golden_realdims.py``: imported from /root/reference, eager-loss shim, draft in fp32 = truth and in
bf16 = yardstick, teacher head in bf16) on inputs that ``oracle/seeded_case.py`` regenerates bit-identically from a seed (checksums in
the fixture are verified first).  Cases: cfg 1 at full dims and its own batch shape (Qwen2.5-0.5B, 1 x 256), cfg 2 dims (Llama-3-8B,
llama3 rope scaling, 2 x 512 ragged, prompt region), cfg 3 dims (Qwen3-8B, 1 x 640), cfg 4 dims (Qwen3-30B-A3B EAGLE3.1: fc_norm, 1 x 384), cfg 5 dims
(DeepSeek-V3 draft: H 7168, I 40960, 1 x 256) and a head_dim-256 recipe (Qwen3-Next-80B-A3B dims, 1 x 320).

Bars (BASELINE.json north_star: tree indices bit-exact, loss 2e-2 for the bf16 path -- held at 5e-3 here):
  * target ids / position mask: the teacher logits are one bf16 GEMM on either side (CPU fp32-accumulate vs MFMA): a 1-ulp rounding
    difference can flip an exact tie, so >= 99.5 % (measured: see profiles/); position mask == t2d[own ids] * loss_mask bit-exactly;
    ``acc_denoms`` bit-exact
  * plosses / loss / acceptance rates: 5e-3;  accuracies: differ on <= 2 tokens per step
  * every parameter gradient, on the fixture's 4096 sampled entries, its row sums and its column sums (these two relative to the tensor's norm): relative L2 error vs the fp32
    reference run no larger than max(2e-2, 1.15 x the reference's own bf16 run), capped at 8e-2; Frobenius norm within 3 %
``-m "not gpu"``: the fixture's checksums reproduce here, and the ORACLE PORT meets the same reference outputs at cfg 1 full dims (fp32,
1e-4) -- the port is pinned at real dimensions too, not only on the tiny goldens.
"""
import json
import os

import pytest
import torch

from oracle import eagle3_oracle as O
from oracle import seeded_case as SC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["cfg1_qwen2.5-0.5b_1x256", "cfg2_llama3-8b_2x512", "cfg3_qwen3-8b_1x640", "cfg4_qwen3-30b-a3b-eagle31_1x384", "cfg5_deepseek-v3_1x256",
         "qwen3-next-80b-a3b_1x320"]


def _load(golden_dir, name):
    blob = torch.load(os.path.join(golden_dir, f"realdims_{name}.pt"), weights_only=False)
    c = blob["dims"]
    case = SC.make_case(c, c["seed"])
    got = SC.case_checksums(*case)
    assert got == blob["checksums"], {k: (v, blob["checksums"][k]) for k, v in got.items() if v != blob["checksums"][k]}
    return blob, c, case


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def _grad_report(grads, blob, c):
    """per tensor: HIP (or port) vs the reference fp32 run beside the reference's own bf16 run vs its fp32 run"""
    rows = {}
    for k, g in grads.items():
        idx = SC.grad_probe_indices(g.shape, c["seed"] + 7)
        mine = SC.grad_summary(g, idx)
        t, y = blob["reference_fp32"]["grads"][k], blob["reference_bf16"]["grads"][k]
        r = dict(samples=_rel(mine["samples"], t["samples"]), samples_yard=_rel(y["samples"], t["samples"]),
                 sample_max=float((mine["samples"] - t["samples"]).abs().max() / t["amax"]),
                 fro=abs(mine["fro"] / t["fro"] - 1.0), fro_yard=abs(y["fro"] / t["fro"] - 1.0))
        if g.dim() == 2:
            # sums are judged against the TENSOR's norm: some are identically zero in exact arithmetic (the columns of d(lm_head) sum a softmax
            # minus a distribution over the vocabulary), and the error of a sum of n rounding errors is of the order of the tensor's own
            for s in ("rowsum", "colsum"):
                r[s] = float((mine[s].double() - t[s].double()).norm() / t["fro"])
                r[s + "_yard"] = float((y[s].double() - t[s].double()).norm() / t["fro"])
        rows[k] = r
    return rows


def _assert_grads(rows, floor=2e-2, factor=1.15, cap=8e-2):
    bad = {}
    for k, r in rows.items():
        for m in ("samples", "rowsum", "colsum"):
            if m in r and (r[m] > max(floor, factor * r[m + "_yard"]) or r[m] > cap):
                bad[(k, m)] = (r[m], r[m + "_yard"])
        if r["fro"] > 3e-2:
            bad[(k, "fro")] = r["fro"]
    assert not bad, bad


def test_fixture_inputs_regenerate_and_the_port_meets_the_reference_at_cfg1_full_dims(golden_dir):
    blob, c, case = _load(golden_dir, CASES[0])
    params, embed, head_w, t2d, d2t, batch = case
    ref = blob["reference_fp32"]
    oc = O.DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                       vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                       max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=bool(c.get("fc_norm")))
    p = {k: v.float().requires_grad_(True) for k, v in params.items()}
    out = O.eagle3_forward(p, oc, embed_weight=embed.float(), target_head_weight=head_w, t2d=t2d, d2t=d2t, input_ids=batch["input_ids"],
                           attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"].float(),
                           target_hidden=batch["target"], ttt_length=c["ttt"])
    out.loss.backward()
    assert torch.equal(out.target_token_ids, ref["target_token_ids"])
    assert torch.equal(out.position_mask.squeeze(-1).to(torch.int8), ref["position_mask"])
    torch.testing.assert_close(torch.stack([x.detach() for x in out.plosses]), ref["plosses"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(torch.stack(out.acces).float(), ref["acces"], rtol=0, atol=1e-6)
    torch.testing.assert_close(torch.stack(out.acceptance_rates).float(), ref["acceptance_rates"], rtol=1e-4, atol=1e-5)
    assert torch.equal(torch.stack(out.acc_denoms).float(), ref["acc_denoms"])
    rows = _grad_report({k: v.grad for k, v in p.items()}, blob, c)
    for k, r in rows.items():
        assert r["samples"] <= 1e-3 and r["fro"] <= 1e-4, (k, r)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_path_matches_the_reference_run_at_real_dims(golden_dir, name):
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3

    blob, c, case = _load(golden_dir, name)
    params, embed, head_w, t2d, d2t, batch = case
    ref, yard = blob["reference_fp32"], blob["reference_bf16"]
    dev = torch.device("cuda", 0)
    model = LlamaForCausalLMEagle3(DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"],
                                               num_key_value_heads=c["nkv"], vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"],
                                               target_hidden_size=c["Ht"], max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"],
                                               fc_norm=bool(c.get("fc_norm")), rope_theta=c.get("rope_theta", 10000.0),
                                               rope_scaling=c.get("rope_scaling")), device=dev)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(dev)))
    out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"],
                                             hidden_state=batch["hidden_state"].to(dev), target=batch["target"].to(dev)),
                                        {"target_repr": "hidden_state"}))
    out.loss.backward()
    torch.cuda.synchronize()
    ids = eagle.last_artifacts["target_token_ids"].cpu()
    pm = eagle.last_artifacts["position_mask"].cpu().to(torch.int8)
    on = batch["loss_mask"].bool() if eagle.engine._teacher_compacted else torch.ones_like(ids, dtype=torch.bool)
    st = lambda k: torch.stack(out.metrics[k]).float().cpu()
    named = dict(model.named_parameters())
    rows = _grad_report({k: named[k].grad.float().cpu() for k in params}, blob, c)
    B, S = batch["input_ids"].shape
    rep = dict(case=name, dims=c, reference="LlamaForCausalLMEagle3(sdpa) + OnlineEagle3Model, CPU, imported in the build container",
               ids_agree=float((ids == ref["target_token_ids"])[on].float().mean()), ids_compared_on=float(on.float().mean()),
               pos_mask_agree=float((pm == ref["position_mask"]).float().mean()),
               plosses=st("plosses").tolist(), plosses_ref=ref["plosses"].tolist(), plosses_ref_bf16=yard["plosses"].tolist(),
               ploss_max_abs=float((st("plosses") - ref["plosses"]).abs().max()),
               ploss_max_abs_ref_bf16=float((yard["plosses"] - ref["plosses"]).abs().max()),
               acceptance_max_abs=float((st("acceptance_rates") - ref["acceptance_rates"]).abs().max()),
               acc_tokens_max=float(((st("acces") - ref["acces"]).abs() * ref["acc_denoms"]).max()), grads=rows)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", f"parity_reference_{name}.json"), "w") as f:
        json.dump(rep, f, indent=1)
    print(f"\n[reference -> HIP {name}] ids {rep['ids_agree']:.5f} ploss max|d| {rep['ploss_max_abs']:.2e} (reference bf16 run "
          f"{rep['ploss_max_abs_ref_bf16']:.2e}) grads samples/yard: " + str({k: (round(r['samples'], 4), round(r['samples_yard'], 4)) for k, r in rows.items()}))
    assert rep["ids_agree"] >= 0.995 and rep["pos_mask_agree"] >= 0.995
    assert torch.equal(pm, (t2d[ids].to(torch.int8) * batch["loss_mask"].to(torch.int8)))
    assert torch.equal(st("acc_denoms"), ref["acc_denoms"])
    tol = 5e-3
    torch.testing.assert_close(st("plosses"), ref["plosses"], rtol=tol, atol=tol)
    torch.testing.assert_close(out.loss.detach().float().cpu(), ref["loss"], rtol=tol, atol=tol)
    torch.testing.assert_close(st("acceptance_rates"), ref["acceptance_rates"], rtol=tol, atol=tol)
    assert rep["acc_tokens_max"] <= 2.0 + 1e-3, rep["acc_tokens_max"]
    _assert_grads(rows)
