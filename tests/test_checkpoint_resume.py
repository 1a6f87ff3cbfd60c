"""Checkpoint / resume contract of the training backend (reference: specforge/training/checkpoint.py,
controller.py:839-886, tests/test_runtime/test_checkpoint_resume.py): the state a run saves --
``{"model", "optimizer", "rng"}`` with the reference's draft state-dict key set and BF16Optimizer
layout -- restores a run that continues BIT-IDENTICALLY, and the filtered draft state dict is exactly
what ``export --to sglang`` consumes (specforge/export/to_sglang.py:33-54)."""
import os

import torch

from specforge_amd import _lib

REF_DRAFT_KEYS = {  # observed key set of the reference's draft checkpoint
    "d2t", "t2d", "fc.weight", "lm_head.weight", "norm.weight", "midlayer.hidden_norm.weight",
    "midlayer.input_layernorm.weight", "midlayer.post_attention_layernorm.weight", "midlayer.mlp.gate_proj.weight",
    "midlayer.mlp.up_proj.weight", "midlayer.mlp.down_proj.weight", "midlayer.self_attn.q_proj.weight",
    "midlayer.self_attn.k_proj.weight", "midlayer.self_attn.v_proj.weight", "midlayer.self_attn.o_proj.weight",
}


def _make(golden_dir):
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend

    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"])
    model = LlamaForCausalLMEagle3(cfg)
    sd = dict(blob["params"])
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"], blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=2).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"]))
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=5e-3, total_steps=20, warmup_ratio=0.1))
    backend.prepare_model(eagle)
    b = blob["batch"]
    batch = TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                            hidden_state=b["hidden_state"], target=b["target"]), {"target_repr": "hidden_state"})
    return eagle, strat, backend, batch


def _step(strat, backend, batch):
    out = strat.forward_loss(batch)
    backend.backward(out.loss, is_boundary=True)
    return backend.step()


def test_resume_is_bit_identical_and_keys_match_reference(golden_dir, emu_lib_path, tmp_path):
    _lib._inject_library_for_tests(emu_lib_path)
    try:
        eagle, strat, backend, batch = _make(golden_dir)
        for _ in range(2):
            _step(strat, backend, batch)
        state = backend.state_dict()
        torch.save(state, tmp_path / "training_state.pt")
        # (a) key contract: filtered draft state dict == the reference's checkpoint key set
        draft_sd = strat.checkpoint_state_filter(state["model"])
        assert set(draft_sd) == REF_DRAFT_KEYS
        opt = state["optimizer"]
        assert set(opt) == {"optimizer_state_dict", "scheduler_state_dict", "lr_scheduler_type", "max_grad_norm", "fp32_params"}
        assert set(opt["optimizer_state_dict"]) == {"state", "param_groups"}
        n_train = sum(1 for p in eagle.draft_model.parameters() if p.requires_grad)
        assert len(opt["fp32_params"]) == n_train and set(opt["optimizer_state_dict"]["state"]) == set(range(n_train))
        assert set(opt["optimizer_state_dict"]["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
        # torch.optim.AdamW accepts the optimizer_state_dict as-is (layout compatibility with the reference)
        ref_opt = torch.optim.AdamW([torch.nn.Parameter(t.clone()) for t in opt["fp32_params"]], lr=1e-3)
        ref_opt.load_state_dict(opt["optimizer_state_dict"])
        # (b) uninterrupted third step
        _step(strat, backend, batch)
        want = eagle.engine.flat.data.clone()
        want_lr = backend.optimizer.get_learning_rate()
        # (c) fresh process state + resume + third step
        eagle2, strat2, backend2, batch2 = _make(golden_dir)
        backend2.load_state_dict(torch.load(tmp_path / "training_state.pt", weights_only=False))
        _step(strat2, backend2, batch2)
        assert torch.equal(eagle2.engine.flat.data, want)
        assert backend2.optimizer.get_learning_rate() == want_lr
        assert backend2.optimizer.step_count == 3
    finally:
        _lib._inject_library_for_tests(None)
