"""Loss curve at BASELINE.json configs[0]'s REAL dimensions against the REFERENCE TRAINER's own run (north_star: "loss curve within
stated tolerance of reference").

``tests/golden/loss_curve_cfg1_realdims.pt`` holds what the reference's trainer LOGGED (``build_offline_runtime -> Trainer.fit()``: its
reader, normaliser, collator, sampler, ``TrainerCore``, ``BF16Optimizer``; sdpa, CPU, bf16 parameters; oracle/gen_curve_realdims.py, build
container) over 12 optimizer steps on 4 ragged feature files, 3 epochs, Qwen2.5-0.5B draft JSON unmodified (H 896, 14 / 2 heads of 64,
Vt 151 936, Vd 16 000), ttt 7, batch 1, max_length 256 (two files are longer and get truncated, every file has a prompt prefix without
loss).  Inputs come back from the seed (oracle/curve_case.py, checksums verified).

``-m gpu``: the HIP side is the product's own chain -- ``HiddenStateIngest.epoch()`` on the same files (its shard order must equal the
order the reference consumed them in), ``Eagle3TrainStrategy`` + ``HipDPTrainingBackend`` + fused ``BF16Optimizer`` through
``TrainerCore.train_step`` -- and every logged value of every step must agree within the bf16 tolerance 2e-2 (relative for values above 1; the
gradient norm 5e-2),
the learning rate to 1e-6, the final weights' sampled entries to 2e-2 in relative L2 (entry-wise: within the summed learning rates).
``-m "not gpu"``: inputs regenerate bit-identically and the shard order matches (no compute).
"""
import json
import math
import os

import pytest
import torch

from oracle import curve_case as CC
from specforge_amd.training import distributed_sampler_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _blob(golden_dir):
    return torch.load(os.path.join(golden_dir, "loss_curve_cfg1_realdims.pt"), weights_only=False)


def _file_index(sample_id):      # 'curve-real:00000002' -> 2
    return int(str(sample_id).rsplit(":", 1)[1])


def test_fixture_inputs_regenerate_and_shard_order_matches(golden_dir):
    blob = _blob(golden_dir)
    c = blob["cfg"]
    params, embed, head_w, t2d, d2t, raws, lengths = CC.make_inputs(c)
    assert lengths == blob["lengths"]
    assert CC.input_checksums(params, embed, head_w, t2d, d2t, raws) == blob["checksums"]
    order = [_file_index(o[0]) for o in blob["order"]]
    mine = []
    for e in range(c["steps"] // c["n_files"]):
        mine += distributed_sampler_indices(c["n_files"], dp_rank=0, dp_size=1, seed=c["seed"], epoch=e, shuffle=True)
    assert mine == order, (mine, order)
    assert len(blob["logged"]) == c["steps"] and any(L > c["max_len"] for L in lengths)


@pytest.mark.gpu
def test_hip_training_chain_reproduces_the_reference_trainers_curve_at_cfg1_real_dims(golden_dir, tmp_path):
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead
    from specforge_amd.ingest import HiddenStateIngest
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend, TrainerCore

    blob = _blob(golden_dir)
    c = blob["cfg"]
    params, embed, head_w, t2d, d2t, raws, lengths = CC.make_inputs(c)
    assert CC.input_checksums(params, embed, head_w, t2d, d2t, raws) == blob["checksums"]
    files = CC.write_files(str(tmp_path / "features"), raws)
    dev = torch.device("cuda", 0)
    model = LlamaForCausalLMEagle3(DraftConfig.from_hf(blob["draft_config"]), device=dev)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(dev)))
    be = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(
        m, lr=c["lr"], max_grad_norm=c["max_grad_norm"], warmup_ratio=c["warmup_ratio"], total_steps=c["steps"]))
    be.prepare_model(eagle)
    core = TrainerCore(strat, be, accumulation_steps=1)
    ingest = HiddenStateIngest(files, batch_size=c["batch_size"], max_len=c["max_len"], target_hidden_size=c["Ht"], device=dev, seed=c["seed"])
    T = c["ttt"]
    worst, curve, step = {}, [], 0
    for epoch in range(c["steps"] // c["n_files"]):
        for batch in ingest.epoch(epoch):
            assert batch.metadata["sample_indices"] == [_file_index(blob["order"][step][0])]
            res = core.train_step(batch)
            want, m = blob["logged"][step], res.metrics
            got = {"loss": float(sum((0.8 ** i) * float(m["plosses"][i]) for i in range(T))), "grad_norm": float(res.grad_norm),
                   "lr": be.optimizer.get_learning_rate()}
            for i in range(T):
                got[f"ploss_{i}"] = float(m["plosses"][i])
                got[f"acc_{i}"] = float(m["acc_corrects"][i]) / max(float(m["acc_denoms"][i]), 1e-6)
                got[f"acceptance_rate_{i}"] = float(m["acceptance_rates"][i])
            curve.append(dict(step=step + 1, loss=got["loss"], loss_ref=want["loss"], grad_norm=got["grad_norm"], grad_norm_ref=want["grad_norm"]))
            for k, v in got.items():
                # (the gradient norm is the one logged value that amplifies: at the step where it spikes to 2.26 the two bf16 runs are 1.6e-2 apart)
                tol = 1e-9 + 1e-6 * abs(want[k]) if k == "lr" else (5e-2 if k == "grad_norm" else 2e-2) * max(1.0, abs(want[k]))
                err = abs(v - want[k])
                fam = k.rstrip("0123456789")
                worst[fam] = max(worst.get(fam, 0.0), err / max(1.0, abs(want[k])))
                assert err <= tol, (step + 1, k, v, want[k])
            step += 1
    assert step == c["steps"]
    torch.cuda.synchronize()
    final = CC.weight_summary({k: v.detach().cpu() for k, v in model.state_dict().items()}, c["seed"])
    # final weights: an AdamW update moves an entry by up to ~lr per step whatever the gradient's size, so two bf16 implementations drift
    # apart entry-wise by a few lr where a gradient entry is noise; the bars are the relative L2 distance over the 2048 sampled entries
    # (2e-2) and the largest entry-wise distance against the summed learning rates + one bf16 spacing at the tensor's largest entry (the
    # fp32 masters may differ by the former, the bf16 copies compared here then by one rounding step more: 2^-7 for a norm weight near 1)
    lr_sum = float(sum(m["lr"] for m in blob["logged"]))
    wdev = {}
    for k, w in blob["final"].items():
        d = final[k]["samples"].double() - w["samples"].double()
        ulp = 2.0 ** (math.floor(math.log2(float(w["samples"].abs().max()))) - 7)
        wdev[k] = dict(rel_l2=float(d.norm() / w["samples"].double().norm()), max_abs=float(d.abs().max()), max_abs_bar=lr_sum + ulp,
                       fro_ratio=final[k]["fro"] / w["fro"])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "loss_curve_cfg1_realdims.json"), "w") as f:
        json.dump(dict(reference="build_offline_runtime -> Trainer.fit(), CPU, bf16 parameters, imported in the build container", cfg=c,
                       lengths=lengths, worst_relative_deviation=worst, curve=curve, lr_sum=lr_sum, final_weights=wdev), f, indent=1)
    for k, r in wdev.items():
        assert r["rel_l2"] <= 2e-2 and r["max_abs"] <= r["max_abs_bar"] and abs(r["fro_ratio"] - 1.0) <= 5e-3, (k, r)
    print("\n[loss curve cfg 1 real dims] worst relative deviation over", step, "steps:", {k: f"{v:.2e}" for k, v in worst.items()},
          "final weights rel L2:", f"{max(r['rel_l2'] for r in wdev.values()):.2e}")
