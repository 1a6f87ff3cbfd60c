"""Real-dimension goldens made by RUNNING THE REAL REFERENCE (imported from /root/reference) -- build container only.

    python oracle/gen_golden_realdims.py [case ...]        # writes tests/golden/realdims_<case>.pt  (tens of KB each)

Why: the GPU box has no /root/reference and a Python reference may not travel there in any form, so until round 6 every
real-dimension parity number on hardware went reference -> tiny golden -> oracle port -> HIP.  These fixtures remove the
port from that chain: the reference's OWN ``LlamaForCausalLMEagle3`` (sdpa backend, llama3_eagle.py:1653-1798) under its own
``OnlineEagle3Model.forward`` (eagle3/model.py:244-442), eager-loss shim (core/loss.py:15-21), run HERE on the CPU at the
models' real dimensions; the fixture keeps what the reference produced and the ``-m gpu`` test compares the HIP path with it
directly (tests/test_reference_realdims.py).  The inputs are 1-3 GB, so the fixture holds their SEED and checksums instead:
``oracle/seeded_case.py`` regenerates them bit-identically on any host (integer draws only).

Per case: the draft in fp32 ("truth") and in bf16 ("yardstick" = what the reference itself produces at the precision it
trains in); the frozen teacher head in bf16 both times (TargetHead is a bf16 module in the reference's runs).
Kept: plosses, accuracies, acceptance rates, counts, target ids, position mask, and of every parameter gradient its
Frobenius norm, max, row / column sums and 4096 sampled entries.
"""
import os
import sys
import time

os.environ["SPECFORGE_DEVICE"] = "cpu"
os.environ["TORCHDYNAMO_DISABLE"] = "1"
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import specforge.algorithms.eagle3.model as ref_model  # noqa: E402
import specforge.core.loss as ref_loss  # noqa: E402


class _EagerLoss:
    @staticmethod
    def apply(logits, target, mask):
        return ref_loss._compute_loss(logits, target, mask)


ref_model.LogSoftmaxLoss = _EagerLoss
from transformers import LlamaConfig  # noqa: E402

from specforge.modeling.draft.llama3_eagle import LlamaForCausalLMEagle3  # noqa: E402
from specforge.modeling.target.target_head import TargetHead  # noqa: E402

from oracle import seeded_case as SC  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# dims: SURVEY.md section 8 table (configs/*.json of the reference)
CASES = {
    # cfg 1 -- the reference's own CPU-runnable case, at FULL model dimensions and its own batch shape (1 x 256)
    "cfg1_qwen2.5-0.5b_1x256": dict(H=896, Ht=896, I=4864, nh=14, nkv=2, hd=64, Vt=151936, Vd=16000, B=1, S=256, ttt=7,
                                    lengths=[256], prompt=9, eps=1e-6, max_pos=2048, seed=101),
    # cfg 2 dims (Llama-3-8B draft, llama3 rope scaling as in configs/llama3-8B-eagle3.json), two ragged samples of 512
    "cfg2_llama3-8b_2x512": dict(H=4096, Ht=4096, I=14336, nh=32, nkv=8, hd=128, Vt=128256, Vd=32000, B=2, S=512, ttt=7,
                                 lengths=[512, 389], prompt=17, eps=1e-5, max_pos=2048, seed=102, rope_theta=500000.0,
                                 rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                                   original_max_position_embeddings=8192)),
    # cfg 3 dims (Qwen3-8B draft: I 12288, 152 k vocabulary, rope theta 1e6)
    "cfg3_qwen3-8b_1x640": dict(H=4096, Ht=4096, I=12288, nh=32, nkv=8, hd=128, Vt=151936, Vd=32000, B=1, S=640, ttt=7,
                                lengths=[640], prompt=11, eps=1e-6, max_pos=40960, seed=103, rope_theta=1000000.0),
    # cfg 4 dims (Qwen3-30B-A3B EAGLE3.1: fc_norm, nh * hd != H)
    "cfg4_qwen3-30b-a3b-eagle31_1x384": dict(H=2048, Ht=2048, I=12288, nh=32, nkv=4, hd=128, Vt=151936, Vd=32000, B=1, S=384,
                                             ttt=7, lengths=[384], prompt=5, eps=1e-6, max_pos=2048, fc_norm=True, seed=104),
    # cfg 5 dims (DeepSeek-V3 draft: H 7168, I 40960, 56 / 8 heads, 21504-wide hidden-state fusion)
    "cfg5_deepseek-v3_1x256": dict(H=7168, Ht=7168, I=40960, nh=56, nkv=8, hd=128, Vt=129280, Vd=32000, B=1, S=256, ttt=7,
                                   lengths=[256], prompt=7, eps=1e-5, max_pos=4096, seed=105),
    # head_dim 256 (configs/qwen3-next-80b-a3b-eagle3.json: 16 / 2 heads, nh * hd = 2 H)
    "qwen3-next-80b-a3b_1x320": dict(H=2048, Ht=2048, I=16384, nh=16, nkv=2, hd=256, Vt=151936, Vd=32000, B=1, S=320, ttt=7,
                                     lengths=[320], prompt=13, eps=1e-6, max_pos=8192, seed=106, rope_theta=10000000.0),
}


def run_reference(c, case, dtype):
    params, embed, head_w, t2d, d2t, batch = case
    cfg = LlamaConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      num_hidden_layers=1, vocab_size=c["Vt"], max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"],
                      pad_token_id=0, head_dim=c["hd"], rope_theta=c.get("rope_theta", 10000.0))
    if c.get("rope_scaling") is not None:
        cfg.rope_parameters = dict(c["rope_scaling"], rope_theta=c.get("rope_theta", 10000.0))
    cfg.draft_vocab_size = c["Vd"]
    cfg.target_hidden_size = c["Ht"]
    cfg.fc_norm = bool(c.get("fc_norm"))
    cfg.norm_output = True
    with torch.device("meta"):
        model = LlamaForCausalLMEagle3(cfg, attention_backend="sdpa")
    model = model.to_empty(device="cpu")
    # (to_empty leaves the non-persistent rotary buffers uninitialised: rebuild the rotary module on the CPU)
    fresh_attn = type(model.midlayer.self_attn)(cfg)
    model.midlayer.self_attn.rotary_emb = fresh_attn.rotary_emb
    del fresh_attn
    sd = {k: v.to(dtype) for k, v in params.items()}
    sd["embed_tokens.weight"] = embed.to(dtype)
    sd["t2d"], sd["d2t"] = t2d, d2t
    missing, unexpected = model.load_state_dict(sd, strict=False, assign=True)
    assert not unexpected and not missing, (missing, unexpected)
    model.freeze_embedding()
    for p in model.parameters():
        if p.requires_grad:
            assert p.dtype == dtype
    eagle = ref_model.OnlineEagle3Model(model, length=c["ttt"], attention_backend="sdpa")
    # Eagle3TrainStrategy.forward_loss glue (training/strategies/base.py:237-304); the head is bf16 like the reference's TargetHead
    input_ids, target, loss_mask = TargetHead.preprocess(None, batch["input_ids"], batch["target"], batch["loss_mask"])
    target_logits = F.linear(target, head_w)
    assert target_logits.dtype == torch.bfloat16
    outs = eagle(input_ids=input_ids, attention_mask=batch["attention_mask"], loss_mask=loss_mask, target=target_logits,
                 hidden_states=batch["hidden_state"].to(dtype), position_ids=None)
    plosses, acceptance_rates, acces, corrects, denoms, _, _ = outs
    loss = sum((0.8 ** i) * plosses[i] for i in range(len(plosses)))
    loss.backward()
    _, _, ids, pm = ref_model._compute_target_p(target=target_logits, t2d=model.t2d, loss_mask=loss_mask)
    grads = {n: p.grad for n, p in model.named_parameters() if p.requires_grad}
    assert set(grads) == set(params), set(grads) ^ set(params)
    st = lambda xs: torch.stack([x.detach().float() for x in xs])
    res = dict(plosses=st(plosses), acces=st(acces), acceptance_rates=st(acceptance_rates), acc_corrects=st(corrects),
               acc_denoms=st(denoms), loss=loss.detach().float(), target_token_ids=ids.clone(), position_mask=pm.squeeze(-1).to(torch.int8),
               grads={k: SC.grad_summary(g, SC.grad_probe_indices(g.shape, c["seed"] + 7)) for k, g in grads.items()})
    return res


def main(only):
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(os.cpu_count() or 8)
    for name, c in CASES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        case = SC.make_case(c, c["seed"])
        sums = SC.case_checksums(*case)
        t1 = time.time()
        truth = run_reference(c, case, torch.float32)
        t2 = time.time()
        yard = run_reference(c, case, torch.bfloat16)
        t3 = time.time()
        blob = dict(case=name, dims=c, checksums=sums, reference_fp32=truth, reference_bf16=yard,
                    made_by="oracle/gen_golden_realdims.py", torch=torch.__version__,
                    seconds=dict(inputs=round(t1 - t0, 1), fp32=round(t2 - t1, 1), bf16=round(t3 - t2, 1)))
        path = os.path.join(OUT, f"realdims_{name}.pt")
        torch.save(blob, path)
        print(name, "loss", float(truth["loss"]), "(bf16 run:", float(yard["loss"]), ") plosses", [round(float(p), 5) for p in truth["plosses"]],
              "ids agree fp32/bf16 runs", float((truth["target_token_ids"] == yard["target_token_ids"]).float().mean()),
              os.path.getsize(path) // 1024, "KiB", blob["seconds"], flush=True)


if __name__ == "__main__":
    main(set(sys.argv[1:]))
