"""Pin the CPU oracle against vectors produced by the REAL reference
(oracle/gen_golden.py) and against the reference tests' own golden artefacts."""
import os

import pytest
import torch

from oracle import eagle3_oracle as O


def _cfg(c):
    return O.DraftConfig(
        hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
        vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
        max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"], rope_scaling=c["rope_scaling"],
        norm_output=c.get("norm_output", True),
    )


def _run(blob):
    cfg = _cfg(blob["cfg"])
    p = {k: v.clone().requires_grad_(True) for k, v in blob["params"].items()}
    b = blob["batch"]
    out = O.eagle3_forward(
        p, cfg, embed_weight=blob["embed"], target_head_weight=blob["head_w"], t2d=blob["t2d"], d2t=blob["d2t"],
        input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
        hidden_state=b["hidden_state"], target_hidden=b["target"], ttt_length=blob["cfg"]["ttt"],
        lk_loss_type=blob["cfg"].get("lk_loss_type"), kl_scale=blob["cfg"].get("kl_scale", 1.0),
        kl_decay=blob["cfg"].get("kl_decay", 1.0), position_ids=b.get("position_ids"),
    )
    out.loss.backward()
    return p, out


@pytest.mark.parametrize("name,tol", [("eagle3_tiny_fp32", 1e-5), ("eagle31_gqa_fp32", 1e-5), ("eagle3_tiny_bf16", 2e-2),
                                      ("eagle3_lk_alpha_fp32", 1e-5), ("eagle3_lk_lambda_fp32", 1e-5),
                                      ("eagle3_nonorm_fp32", 1e-5), ("eagle3_rope_yarn_fp32", 1e-5),
                                      ("eagle3_rope_dynamic_fp32", 1e-5), ("eagle3_rope_linear_fp32", 1e-5), ("eagle3_hd256_fp32", 1e-5),
                                      ("eagle3_rope_mrope_fp32", 1e-5),
                                      # S = 44 against max_position_embeddings 16: the rotary cache is rebuilt per TTT step (RopeCache)
                                      ("eagle3_rope_grow_fp32", 1e-5), ("eagle3_rope_grow_dynamic_fp32", 1e-5)])
def test_oracle_matches_reference_run(golden_dir, name, tol):
    blob = torch.load(os.path.join(golden_dir, f"{name}.pt"), weights_only=False)
    p, out = _run(blob)
    # integer artefacts: bit-exact
    assert torch.equal(out.target_token_ids, blob["target_token_ids"])
    assert torch.equal(out.position_mask, blob["position_mask"])
    assert torch.equal(torch.stack(out.acc_corrects).float(), blob["acc_corrects"])
    assert torch.equal(torch.stack(out.acc_denoms).float(), blob["acc_denoms"])
    # floats
    torch.testing.assert_close(torch.stack([x.detach().float() for x in out.plosses]), blob["plosses"], rtol=tol, atol=tol)
    torch.testing.assert_close(torch.stack(out.acceptance_rates).float(), blob["acceptance_rates"], rtol=tol, atol=tol)
    torch.testing.assert_close(out.loss.detach().float(), blob["loss"], rtol=tol, atol=tol)
    for k, g in blob["grads"].items():
        key = k.replace("fc_norm.", "fc_norm.")
        got = p[key].grad if p[key].grad is not None else torch.zeros_like(p[key])  # norm_output=False: unused `norm`
        scale = g.float().abs().max().clamp_min(1e-8)
        err = (got.float() - g.float()).abs().max() / scale
        assert err < (1e-4 if tol < 1e-3 else 5e-2), (k, float(err))


def test_ttt_mask_matches_reference_golden_block_mask():
    """Reference golden: tests/test_utils/test_flex_attention.py:245-284 (Q_LEN=1024,
    KV_LEN=3072, seq_len=1024-256, 128x128 blocks)."""
    S, blk = 1024, 128
    dense = O.ttt_mask_dense(S - 256, S, 3)
    got = dense.view(S // blk, blk, 3 * S // blk, blk).amax(dim=(1, 3))
    expected = torch.tensor([
        [1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0],
        [1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0],
        [1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0],
        [1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0],
        [1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0],
        [1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 1, 0, 0],
        [0] * 24,
        [0] * 24,
    ], dtype=torch.int32)
    assert torch.equal(got, expected)


def test_ttt_mask_equals_sdpa_attention_support():
    """The integer mask is exactly the support of the oracle's sdpa-style attention."""
    S, nb = 12, 3
    m = O.ttt_mask_dense(S, S, nb)
    add = O.additive_attention_mask(torch.ones(1, S, dtype=torch.bool), S, torch.float32)
    assert torch.equal((add[0, 0] == 0).int(), m[:, :S])
    for i in range(1, nb):
        assert torch.equal(m[:, i * S:(i + 1) * S], torch.eye(S, dtype=torch.int32))


def test_optimizer_matches_reference_bf16optimizer(golden_dir):
    g = torch.load(os.path.join(golden_dir, "optimizer_bf16.pt"), weights_only=False)
    masters = [p.float().clone() for p in g["p0"]]
    m = [torch.zeros_like(x) for x in masters]
    v = [torch.zeros_like(x) for x in masters]
    for step in range(4):
        lr = O.cosine_warmup_lr(step, g["lr"], g["total_steps"], g["warmup_steps"])
        assert abs(lr - g["lrs"][step]) < 1e-9 * max(1, abs(lr)) + 1e-12
        norm = O.adamw_clip_step(masters, g["grads"][step], m, v, step + 1, lr=lr, max_grad_norm=g["max_grad_norm"])
        torch.testing.assert_close(norm, g["norms"][step], rtol=1e-5, atol=1e-6)
        for mp, ref in zip(masters, g["params_after"][step]):
            assert torch.equal(mp.to(torch.bfloat16), ref)
    for a, b in zip(masters, g["masters"]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(m, g["exp_avg"]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-8)
    for a, b in zip(v, g["exp_avg_sq"]):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-10)
    assert g["state_keys"] == ["fp32_params", "lr_scheduler_type", "max_grad_norm", "optimizer_state_dict", "scheduler_state_dict"]


def test_sampler_indices_match_reference_and_torch(golden_dir):
    from torch.utils.data import DistributedSampler

    cases = torch.load(os.path.join(golden_dir, "sampler_indices.pt"), weights_only=False)
    for c in cases:
        got = O.distributed_sampler_indices(c["size"], dp_rank=c["dp_rank"], dp_size=c["dp_size"], seed=c["seed"], epoch=c["epoch"])
        assert got == c["idx"]
        ds = DistributedSampler(list(range(c["size"])), num_replicas=c["dp_size"], rank=c["dp_rank"], shuffle=True, seed=c["seed"])
        ds.set_epoch(c["epoch"])
        assert got == list(iter(ds))
