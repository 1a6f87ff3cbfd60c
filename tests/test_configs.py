"""The other BASELINE.json configurations as parity cases.

* cfg 1 (Qwen2.5-0.5B EAGLE3, bs=1 seq=256 -- the reference's own CPU-runnable case) at FULL model
  dimensions: hd 64, 14/2 heads, H 896, I 4864, Vt 151936, Vd 16000.
* the shape families of cfg 3/4/5 (Qwen3-8B: I 12288; Qwen3-30B-A3B EAGLE3.1: fc_norm, nh*hd != H,
  32/4 heads; DeepSeek-V3 draft: H 7168, 56/8 heads, 3*Ht = 21504) with the vocabularies and the MLP
  width scaled down so the CPU oracle finishes in seconds -- what is exercised is every dimension
  RELATION the kernels branch on (GQA ratio, q width vs hidden, norm width, head_dim, fused widths).
Each case: one micro-step through the C-ABI on the GPU vs the pinned oracle in bf16 on the same inputs.
"""
import pytest
import torch

from oracle import eagle3_oracle as O
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3

CASES = {
    "cfg1_qwen2.5-0.5b_full": dict(H=896, Ht=896, I=4864, nh=14, nkv=2, hd=64, Vt=151936, Vd=16000, B=1, S=256, ttt=7,
                                   fc_norm=False, lengths=[256]),
    "cfg3_qwen3-8b_family": dict(H=4096, Ht=4096, I=1536, nh=32, nkv=8, hd=128, Vt=4096, Vd=1024, B=1, S=64, ttt=3,
                                 fc_norm=False, lengths=[57]),
    "cfg4_qwen3-30b-a3b_eagle3.1_family": dict(H=2048, Ht=2048, I=1536, nh=32, nkv=4, hd=128, Vt=4096, Vd=1024, B=2, S=40,
                                               ttt=3, fc_norm=True, lengths=[40, 17]),
    "cfg5_deepseek-v3_family": dict(H=7168, Ht=7168, I=2560, nh=56, nkv=8, hd=128, Vt=2048, Vd=512, B=1, S=48, ttt=2,
                                    fc_norm=False, lengths=[48]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_config_micro_step_matches_oracle(name):
    c = CASES[name]
    dev = "cuda"
    kw = dict(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
              vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
              max_position_embeddings=512, rms_norm_eps=1e-6, fc_norm=c["fc_norm"])
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    params = {k: v.to(bf) for k, v in O.init_params(oc, seed=1).items()}
    g = torch.Generator().manual_seed(2)
    for k, v in params.items():  # non-trivial norm weights
        if v.dim() == 1:
            params[k] = (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
    embed = (torch.randn(c["Vt"], c["H"], generator=g) * 0.05).to(bf)
    head_w = (torch.randn(c["Vt"], c["Ht"], generator=g) * 0.05).to(bf)
    t2d, d2t = O.make_vocab_mapping(c["Vt"], c["Vd"], seed=3)
    batch = O.make_batch(oc, c["B"], c["S"], seed=4, dtype=bf, lengths=c["lengths"])

    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.eagle3_forward(p, oc, embed_weight=embed, target_head_weight=head_w, t2d=t2d, d2t=d2t,
                           input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"],
                           hidden_state=batch["hidden_state"], target_hidden=batch["target"], ttt_length=c["ttt"])
    ref.loss.backward()

    model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=dev)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(dev)))
    out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                             loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"].to(dev),
                                             target=batch["target"].to(dev)), {"target_repr": "hidden_state"}))
    out.loss.backward()
    # integer artefacts: teacher argmax ids / position mask.  The teacher logits come from two different bf16 GEMMs
    # (torch CPU vs MFMA) whose last-ulp rounding can differ, so exact ties may resolve differently: require >= 99.5 %.
    ids = eagle.last_artifacts["target_token_ids"].cpu()
    on = batch["loss_mask"].bool() if eagle.engine._teacher_compacted else torch.ones_like(ids, dtype=torch.bool)   # (loss-row compaction)
    agree = float((ids == ref.target_token_ids)[on].float().mean())
    assert agree >= 0.995, agree
    tol = 2e-2
    pl = torch.stack(out.metrics["plosses"]).float().cpu()
    want = torch.stack([x.detach().float() for x in ref.plosses])
    torch.testing.assert_close(pl, want, rtol=tol, atol=tol)
    torch.testing.assert_close(torch.stack(out.metrics["acc_denoms"]).cpu(), torch.stack(ref.acc_denoms).float())
    named = dict(model.named_parameters())
    worst = {}
    for k, v in p.items():
        gref = v.grad.float()
        scale = float(gref.abs().max().clamp_min(1e-8))
        worst[k] = float((named[k].grad.float().cpu() - gref).abs().max()) / scale
    bad = {k: round(v, 4) for k, v in worst.items() if v > 6e-2}
    assert not bad, worst


# ------------------------------------------------------------------ dimension RELATIONS of the reference's other recipes (configs/*.json)
# Scaled-down cases (the CPU oracle and the SIMT interpreter finish them in seconds; [emu] in the CPU suite, [gpu] on hardware) that keep
# what the host code and the kernels branch on:
#   longcat-flash-eagle3.json      draft vocabulary == target vocabulary (no pruning: t2d all true, d2t zero), 64 / 16 heads
#   deepseek-v2-lite-eagle3.json   MHA (nkv == nh), yarn RoPE, I not a multiple of 128 / 256
#   qwen2.5-7b / qwen2-5-vl-7b     28 / 4 heads: 7 query heads per kv group (odd: head pairs of the diagonal kernel, head split of dK/dV)
#   gpt-oss-{20,120}B-eagle3.json  H = 2880 and I = 17280: multiples of 64 but not of 128 / 256 (edge tiles everywhere, unfused SwiGLU dgrad)
#   gemma3-1b-eagle3.json          head_dim 256 with 4 / 1 heads and nh * hd != H
ODD = {
    "longcat_vd_equals_vt": dict(H=128, Ht=128, I=256, nh=4, nkv=1, hd=64, Vt=384, Vd=384, B=2, S=24, ttt=3, lengths=[24, 15]),
    "deepseek_v2_lite_mha_yarn": dict(H=128, Ht=128, I=168, nh=2, nkv=2, hd=64, Vt=512, Vd=128, B=1, S=40, ttt=3, lengths=[37],
                                      rope_scaling=dict(rope_type="yarn", factor=4.0, beta_fast=32, beta_slow=1, mscale=1.0, mscale_all_dim=0.5,
                                                        original_max_position_embeddings=32)),
    "qwen2.5_7b_seven_heads_per_group": dict(H=192, Ht=192, I=256, nh=7, nkv=1, hd=64, Vt=512, Vd=128, B=2, S=32, ttt=4, lengths=[32, 19]),
    "gpt_oss_h_not_multiple_of_128": dict(H=192, Ht=192, I=320, nh=4, nkv=2, hd=64, Vt=640, Vd=192, B=2, S=20, ttt=3, lengths=[20, 11]),
    "gemma3_head_dim_256": dict(H=128, Ht=128, I=192, nh=2, nkv=1, hd=256, Vt=384, Vd=128, B=1, S=36, ttt=3, lengths=[33]),
}


@pytest.mark.parametrize("name", list(ODD))
def test_odd_dimension_relations_match_oracle(backend, name):
    run_small_case(backend, ODD[name])


def run_small_case(backend, c, *, seed=1):
    """one micro-step of a scaled-down configuration through the strategy vs the oracle (also the body of tools/engine_fuzz.py)"""
    kw = dict(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
              vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
              max_position_embeddings=c.get("max_pos", 128), rms_norm_eps=1e-6, rope_scaling=c.get("rope_scaling"),
              fc_norm=c.get("fc_norm", False), norm_output=c.get("norm_output", True))
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    params = {k: v.to(bf) for k, v in O.init_params(oc, seed=seed).items()}
    g = torch.Generator().manual_seed(seed + 1)
    for k, v in params.items():
        if v.dim() == 1:
            params[k] = (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
    embed = (torch.randn(c["Vt"], c["H"], generator=g) * 0.05).to(bf)
    head_w = (torch.randn(c["Vt"], c["Ht"], generator=g) * 0.05).to(bf)
    if c["Vd"] == c["Vt"]:
        t2d, d2t = torch.ones(c["Vt"], dtype=torch.bool), torch.zeros(c["Vd"], dtype=torch.int64)
    else:
        t2d, d2t = O.make_vocab_mapping(c["Vt"], c["Vd"], seed=seed + 2)
    batch = O.make_batch(oc, c["B"], c["S"], seed=seed + 3, dtype=bf, lengths=c["lengths"])
    if c.get("mask_keep", 1.0) < 1.0:      # prompt / user turns: only some positions carry a loss mask
        batch["loss_mask"] = batch["loss_mask"] * (torch.rand(c["B"], c["S"], generator=g) < c["mask_keep"]).long()
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.eagle3_forward(p, oc, embed_weight=embed, target_head_weight=head_w, t2d=t2d, d2t=d2t, input_ids=batch["input_ids"],
                           attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"],
                           target_hidden=batch["target"], ttt_length=c["ttt"])
    ref.loss.backward()
    model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(backend)))
    out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"],
                                             hidden_state=batch["hidden_state"].to(backend), target=batch["target"].to(backend)),
                                        {"target_repr": "hidden_state"}))
    out.loss.backward()
    ids = eagle.last_artifacts["target_token_ids"].cpu()
    on = batch["loss_mask"].bool() if eagle.engine._teacher_compacted else torch.ones_like(ids, dtype=torch.bool)
    assert float((ids == ref.target_token_ids)[on].float().mean()) >= 0.995         # (CPU bf16 GEMM on the other side: exact ties may flip)
    torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(), torch.stack([x.detach().float() for x in ref.plosses]),
                               rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(torch.stack(out.metrics["acc_denoms"]).cpu(), torch.stack(ref.acc_denoms).float())
    named = dict(model.named_parameters())
    for k, v in p.items():       # a parameter the forward never reads (the final norm with norm_output = false) has no gradient on either side
        if v.grad is None:
            assert named[k].grad is None or float(named[k].grad.float().abs().max()) == 0.0, k
    worst = {k: float((named[k].grad.float().cpu() - v.grad.float()).abs().max()) / float(v.grad.float().abs().max().clamp_min(1e-8))
             for k, v in p.items() if v.grad is not None}
    assert max(worst.values()) <= 6e-2, worst


# ------------------------------------------------------------------ the other configs at their REAL dimensions
REAL = {
    # configs/qwen3-8b-eagle3.json
    "cfg3_qwen3_8b": dict(H=4096, Ht=4096, I=12288, nh=32, nkv=8, hd=128, Vt=151936, Vd=32000, B=2, S=512, ttt=7, eps=1e-6,
                          max_pos=40960, rope_theta=1000000.0, lengths=[512, 301]),
    # configs/qwen3-30B-A3B-eagle3.1.json (fc_norm, nh*hd = 2H)
    "cfg4_qwen3_30b_a3b_eagle31": dict(H=2048, Ht=2048, I=12288, nh=32, nkv=4, hd=128, Vt=151936, Vd=32000, B=2, S=512, ttt=7,
                                       eps=1e-6, max_pos=2048, rope_theta=1000000.0, fc_norm=True, lengths=[512, 77]),
    # ... at its recipe's sequence length (bs 1 x 4096), a ragged pair
    "cfg4_qwen3_30b_a3b_eagle31_s4096": dict(H=2048, Ht=2048, I=12288, nh=32, nkv=4, hd=128, Vt=151936, Vd=32000, B=2, S=4096,
                                             ttt=7, eps=1e-6, max_pos=4096, rope_theta=1000000.0, fc_norm=True, lengths=[4096, 2931]),
    # configs/deepseek-v3-671b-eagle3.json (3*Ht = 21504 fusion input, I 40960, 129k vocabulary)
    "cfg5_deepseek_v3": dict(H=7168, Ht=7168, I=40960, nh=56, nkv=8, hd=128, Vt=129280, Vd=32000, B=1, S=512, ttt=7, eps=1e-5,
                             max_pos=163840, lengths=[509]),
    # round 5: the recipes' own sequence lengths.  cfg 3 at S 2048 (BASELINE.json: seq 2048), a ragged pair
    "cfg3_qwen3_8b_s2048": dict(H=4096, Ht=4096, I=12288, nh=32, nkv=8, hd=128, Vt=151936, Vd=32000, B=2, S=2048, ttt=7, eps=1e-6,
                                max_pos=40960, rope_theta=1000000.0, lengths=[2048, 1203]),
    # cfg 5 at its recipe's shape (examples/configs/deepseek-v3-671b-eagle3-offline.yaml: batch 1 x 2048): the H 7168 / I 40960 GEMMs
    # at N = 2048 rows (224-tile grids), a prompt region without loss (loss-row compaction + compact teacher)
    "cfg5_deepseek_v3_s2048": dict(H=7168, Ht=7168, I=40960, nh=56, nkv=8, hd=128, Vt=129280, Vd=32000, B=1, S=2048, ttt=7, eps=1e-5,
                                   max_pos=163840, lengths=[2048], prompt=300),
    # configs/qwen3-next-80b-a3b-eagle3.json (head_dim 256, 16 / 2 heads, nh * hd = 2H) at its recipe's shape
    # (examples/configs/qwen3-next-80b-a3b-eagle3-online.yaml: batch 1, max_length 4096)
    "qwen3_next_80b_a3b_s4096": dict(H=2048, Ht=2048, I=16384, nh=16, nkv=2, hd=256, Vt=151936, Vd=32000, B=1, S=4096, ttt=7, eps=1e-6,
                                     max_pos=8192, rope_theta=10000000.0, lengths=[4096], prompt=700),
    # configs/gemma3-1b-eagle3.json at its recipe's shape (examples/configs/gemma3-1b-eagle3-online.yaml: batch 1, max_length 4096): head_dim 256
    # with 4 / 1 heads (nh * hd = 1024 != H), H = 1152 and I = 6912 -- 4.5 and 27 tiles of 256: edge tiles in every GEMM --, 262k vocabulary
    "gemma3_1b_s4096": dict(H=1152, Ht=1152, I=6912, nh=4, nkv=1, hd=256, Vt=262144, Vd=32000, B=1, S=4096, ttt=7, eps=1e-6,
                            max_pos=32768, rope_theta=1000000.0, lengths=[4096], prompt=512),
    # configs/qwen3.5-35b-a3b-eagle3.json (head_dim 256, 248k target vocabulary) at examples/configs/qwen3.5-35b-a3b-eagle3-online.yaml's
    # max_length 8192: the longest sequence any shipped EAGLE3 recipe trains at
    "qwen3_5_35b_a3b_s8192": dict(H=2048, Ht=2048, I=16384, nh=16, nkv=2, hd=256, Vt=248320, Vd=32000, B=1, S=8192, ttt=7, eps=1e-6,
                                  max_pos=8192, rope_theta=10000000.0, lengths=[8192]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(REAL))
def test_real_dims_match_oracle(name):
    """cfg 3 / 4 / 5 (S 512, and at their recipes' sequence lengths) and the head_dim-256 recipes at the reference's own model
    dimensions vs the pinned oracle in fp32 on the GPU"""
    from tests._parity import compare

    compare(name, REAL[name])
