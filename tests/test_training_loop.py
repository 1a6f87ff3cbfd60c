"""The path TRAINS: a short run of Eagle3TrainStrategy.forward_loss -> backend.backward -> backend.step on one fixed
batch drives every per-step loss down and the draft's accuracy on that batch up (functional check of fwd + bwd +
clip/AdamW + the parameter/weight-image refresh between steps, for the default CE objective and both LK objectives
of specforge/core/lk_loss.py:83-99).  Runs under the SIMT interpreter on CPU and on the GPU."""
import os

import pytest
import torch

from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend, TrainerCore


@pytest.mark.parametrize("lk", [None, "alpha", "lambda"])
def test_loss_decreases_on_a_fixed_batch(backend, golden_dir, lk):
    if lk is not None and str(backend) == "cpu":
        pytest.skip("LK objectives train on the GPU leg; the interpreter leg keeps the CPU suite short")
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"])
    model = LlamaForCausalLMEagle3(cfg, device=backend)
    sd = dict(blob["params"])
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"], blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    T = 3
    eagle = OnlineEagle3Model(model, length=T, lk_loss_type=lk, kl_scale=0.7, kl_decay=1.0).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"].to(torch.bfloat16).to(backend)))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # offload_master (optimizer.py:25-35) is accepted: a placement option, same update
        be = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=3e-3, max_grad_norm=1.0, total_steps=40,
                                                                            warmup_ratio=0.1, offload_master=lk is None))
        be.prepare_model(eagle)
    b = blob["batch"]
    batch = TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                            hidden_state=b["hidden_state"].to(backend), target=b["target"].to(backend)),
                       {"target_repr": "hidden_state"})
    steps = 8
    core = TrainerCore(strat, be, accumulation_steps=1)     # controller.py:328-363 protocol
    losses, accs, norms = [], [], []
    for _ in range(steps):
        res = core.train_step(batch)
        assert res.stepped and res.grad_norm is not None
        norms.append(float(res.grad_norm))
        losses.append(float(res.loss))
        accs.append(float(torch.stack(res.metrics["acces"]).mean()))
    assert core.global_step == steps
    assert all(torch.isfinite(torch.tensor(losses))) and all(n > 0 for n in norms)
    assert losses[-1] < 0.8 * losses[0], losses
    assert min(losses[steps // 2:]) < min(losses[: steps // 2]), losses
    assert accs[-1] >= accs[0], accs


def test_trainer_core_accumulation_boundaries(backend, golden_dir):
    """two micro-steps with accumulation_steps=2 == one optimizer step; the first one neither steps nor reduces"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"])
    model = LlamaForCausalLMEagle3(cfg, device=backend)
    sd = dict(blob["params"])
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"], blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=2).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"].to(torch.bfloat16).to(backend)))
    be = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=1e-3, total_steps=10))
    be.prepare_model(eagle)
    b = blob["batch"]
    batch = TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                            hidden_state=b["hidden_state"].to(backend), target=b["target"].to(backend)),
                       {"target_repr": "hidden_state"})
    core = TrainerCore(strat, be, accumulation_steps=2)
    w0 = eagle.engine.flat.data.clone()
    r1 = core.train_step(batch)
    assert not r1.stepped and r1.grad_norm is None and torch.equal(eagle.engine.flat.data, w0)
    r2 = core.train_step(batch)
    assert r2.stepped and float(r2.grad_norm) > 0 and not torch.equal(eagle.engine.flat.data, w0)
    assert core.global_step == 1
