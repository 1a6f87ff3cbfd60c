"""Generate golden vectors by RUNNING THE REAL REFERENCE (imported from /root/reference).

Run in the build container only (the GPU box has no /root/reference):

    python oracle/gen_golden.py            # writes tests/golden/*.pt

Harness shims (test-only): the Triton ``LogSoftmaxLoss`` is replaced
by the reference file's own eager ``_compute_loss`` (specforge/core/loss.py:15-21) because
Triton has no CPU driver; ``TORCHDYNAMO_DISABLE=1``; ``SPECFORGE_DEVICE=cpu``.
The reference objects exercised: ``LlamaForCausalLMEagle3`` (sdpa backend),
``OnlineEagle3Model.forward``, ``TargetHead.preprocess``, ``BF16Optimizer``,
``_distributed_sampler_indices``, ``compute_target_from_hidden``.
"""
import os
import sys

os.environ["SPECFORGE_DEVICE"] = "cpu"
os.environ["TORCHDYNAMO_DISABLE"] = "1"
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import specforge.core.loss as ref_loss  # noqa: E402


class _EagerLoss:
    @staticmethod
    def apply(logits, target, mask):
        return ref_loss._compute_loss(logits, target, mask)


import specforge.algorithms.eagle3.model as ref_model  # noqa: E402

ref_model.LogSoftmaxLoss = _EagerLoss
from transformers import LlamaConfig  # noqa: E402

from specforge.modeling.draft.llama3_eagle import LlamaForCausalLMEagle3  # noqa: E402
from specforge.modeling.target.target_head import TargetHead  # noqa: E402

from oracle.eagle3_oracle import make_batch, make_vocab_mapping, DraftConfig  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def run_case(name, *, H, Ht, I, nh, nkv, hd, Vt, Vd, B, S, lengths, ttt, dtype, fc_norm=False, rope_scaling=None, seed=0,
             lk_loss_type=None, kl_scale=1.0, kl_decay=1.0, norm_output=True, position_ids=None, max_pos=128):
    torch.manual_seed(seed)
    cfg = LlamaConfig(
        hidden_size=H, intermediate_size=I, num_attention_heads=nh, num_key_value_heads=nkv,
        num_hidden_layers=1, vocab_size=Vt, max_position_embeddings=max_pos, rms_norm_eps=1e-5,
        pad_token_id=0, head_dim=hd, rope_theta=10000.0,
    )
    if rope_scaling is not None:
        cfg.rope_parameters = dict(rope_scaling, rope_theta=10000.0)
    cfg.draft_vocab_size = Vd
    cfg.target_hidden_size = Ht
    cfg.fc_norm = fc_norm
    cfg.norm_output = norm_output
    model = LlamaForCausalLMEagle3(cfg, attention_backend="sdpa")
    # non-trivial norm weights so their gradients are exercised
    with torch.no_grad():
        for n, p in model.named_parameters():
            if p.dim() == 1:
                p.copy_(1.0 + 0.1 * torch.randn_like(p))
    t2d, d2t = make_vocab_mapping(Vt, Vd, seed=seed)
    model.t2d.copy_(t2d)
    model.d2t.copy_(d2t)
    model.freeze_embedding()
    model = model.to(dtype)
    head_w = torch.randn(Vt, Ht).to(dtype)
    ocfg = DraftConfig(hidden_size=H, intermediate_size=I, num_attention_heads=nh, num_key_value_heads=nkv,
                       vocab_size=Vt, draft_vocab_size=Vd, head_dim=hd, target_hidden_size=Ht,
                       max_position_embeddings=max_pos, rms_norm_eps=1e-5, fc_norm=fc_norm, rope_scaling=rope_scaling,
                       norm_output=norm_output)
    batch = make_batch(ocfg, B, S, seed=seed + 1, dtype=dtype, lengths=lengths)

    eagle = ref_model.OnlineEagle3Model(model, length=ttt, attention_backend="sdpa", lk_loss_type=lk_loss_type,
                                        kl_scale=kl_scale, kl_decay=kl_decay)
    # Eagle3TrainStrategy.forward_loss glue (training/strategies/base.py:237-304)
    input_ids, target, loss_mask = TargetHead.preprocess(None, batch["input_ids"], batch["target"], batch["loss_mask"])
    target_logits = F.linear(target, head_w)
    if position_ids is not None:      # [3, B, S] multimodal rotary positions (llama3_eagle.py:389-427, eagle3/model.py:228-242)
        batch["position_ids"] = position_ids
    outs = eagle(
        input_ids=input_ids, attention_mask=batch["attention_mask"], loss_mask=loss_mask,
        target=target_logits, hidden_states=batch["hidden_state"], position_ids=position_ids,
    )
    plosses, acceptance_rates, acces, corrects, denoms, _, _ = outs
    loss = sum((0.8 ** i) * plosses[i] for i in range(len(plosses)))
    loss.backward()
    # integer artefacts straight from the reference helper
    tp, tpod, ids, pm = ref_model._compute_target_p(target=target_logits, t2d=model.t2d, loss_mask=loss_mask)
    # compact teacher agrees (core/compact_teacher.py:107-150)
    from specforge.core.compact_teacher import compute_target_from_hidden
    ctp, ctpod, cids, cpm = compute_target_from_hidden(target, head_w, model.t2d, loss_mask, chunk_size=max(8, Vt // 3))
    assert torch.equal(cids, ids) and torch.equal(cpm, pm)

    params = {n: p.detach().clone() for n, p in model.named_parameters() if p.requires_grad}
    grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p))
             for n, p in model.named_parameters() if p.requires_grad}
    blob = dict(
        cfg=dict(H=H, Ht=Ht, I=I, nh=nh, nkv=nkv, hd=hd, Vt=Vt, Vd=Vd, ttt=ttt, eps=1e-5, fc_norm=fc_norm,
                 max_pos=max_pos, rope_scaling=rope_scaling, lk_loss_type=lk_loss_type, kl_scale=kl_scale, kl_decay=kl_decay,
                 norm_output=norm_output),
        dtype=str(dtype), params=params, grads=grads,
        embed=model.embed_tokens.weight.detach().clone(), head_w=head_w, t2d=t2d, d2t=d2t, batch=batch,
        plosses=torch.stack([p.detach().float() for p in plosses]),
        acces=torch.stack([a.detach().float() for a in acces]),
        acceptance_rates=torch.stack([a.detach().float() for a in acceptance_rates]),
        acc_corrects=torch.stack([c.detach().float() for c in corrects]),
        acc_denoms=torch.stack([d.detach().float() for d in denoms]),
        loss=loss.detach().float(), target_token_ids=ids, position_mask=pm,
        target_p_row0=tp[0, :4].clone(), target_pod_row0=tpod[0, :4].clone(),
    )
    path = os.path.join(OUT, f"{name}.pt")
    torch.save(blob, path)
    print(name, "loss", float(loss), "plosses", [round(float(p), 5) for p in plosses], os.path.getsize(path) // 1024, "KiB")


def run_optimizer_case():
    from specforge.optimizer import BF16Optimizer

    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(24, 16, bias=False), torch.nn.Linear(16, 8, bias=False)).to(torch.bfloat16)
    p0 = [p.detach().clone() for p in model.parameters()]
    opt = BF16Optimizer(model, lr=1e-2, max_grad_norm=0.5, total_steps=10, warmup_ratio=0.2)
    grads, params_after, norms, lrs = [], [], [], []
    g = torch.Generator().manual_seed(4)
    for step in range(4):
        gs = [(torch.randn(p.shape, generator=g) * (3.0 if step % 2 == 0 else 0.01)).to(torch.bfloat16) for p in model.parameters()]
        for p, gg in zip(model.parameters(), gs):
            p.grad = gg.clone()
        lrs.append(opt.get_learning_rate())
        norms.append(opt.step().float().clone())
        grads.append(gs)
        params_after.append([p.detach().clone() for p in model.parameters()])
    sd = opt.state_dict()
    torch.save(dict(p0=p0, grads=grads, params_after=params_after, norms=torch.stack(norms), lrs=lrs,
                    masters=[t.clone() for t in sd["fp32_params"]],
                    state_keys=sorted(sd.keys()),
                    opt_state_keys=sorted(sd["optimizer_state_dict"].keys()),
                    exp_avg=[sd["optimizer_state_dict"]["state"][i]["exp_avg"].clone() for i in range(2)],
                    exp_avg_sq=[sd["optimizer_state_dict"]["state"][i]["exp_avg_sq"].clone() for i in range(2)],
                    lr=1e-2, max_grad_norm=0.5, total_steps=10, warmup_steps=2),
               os.path.join(OUT, "optimizer_bf16.pt"))
    print("optimizer norms", [float(n) for n in norms], "lrs", lrs)


def run_sampler_case():
    from specforge.launch import _distributed_sampler_indices

    cases = []
    for size, dp, seed, epoch in [(10, 2, 0, 0), (11, 4, 3, 2), (64, 8, 0, 1), (5, 8, 1, 0), (7, 1, 0, 0)]:
        for r in range(dp):
            cases.append(dict(size=size, dp_size=dp, dp_rank=r, seed=seed, epoch=epoch,
                              idx=_distributed_sampler_indices(size, dp_rank=r, dp_size=dp, seed=seed, epoch=epoch)))
    torch.save(cases, os.path.join(OUT, "sampler_indices.pt"))
    print("sampler cases", len(cases))


if __name__ == "__main__":
    import sys
    os.makedirs(OUT, exist_ok=True)
    only = set(sys.argv[1:])   # e.g. `python oracle/gen_golden.py lk rope` regenerates just those groups
    want = lambda g: not only or g in only
    common = dict(Ht=128, I=256, nh=2, nkv=1, hd=128, Vt=512, Vd=128, B=2, S=32, lengths=[32, 23])
    if want("base"):
        run_case("eagle3_tiny_fp32", H=128, ttt=4, dtype=torch.float32, **common)
        run_case("eagle3_tiny_bf16", H=128, ttt=4, dtype=torch.bfloat16, **common)
        run_case("eagle31_gqa_fp32", H=128, Ht=96, I=192, nh=4, nkv=2, hd=64, Vt=640, Vd=256, B=1, S=48,
                 lengths=[48], ttt=7, dtype=torch.float32, fc_norm=True,
                 rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                   original_max_position_embeddings=64))
        run_optimizer_case()
        run_sampler_case()
    if want("lk"):   # LK objectives (core/lk_loss.py:83-99) and norm_output=False (llama3_eagle.py:1772-1777)
        run_case("eagle3_lk_alpha_fp32", H=128, ttt=3, dtype=torch.float32, lk_loss_type="alpha", seed=5, **common)
        run_case("eagle3_lk_lambda_fp32", H=128, ttt=3, dtype=torch.float32, lk_loss_type="lambda", kl_scale=0.7,
                 kl_decay=1.5, seed=6, **common)
        run_case("eagle3_nonorm_fp32", H=128, ttt=3, dtype=torch.float32, norm_output=False, seed=7, **common)
    if want("rope"):  # yarn / dynamic-NTK tables (llama3_eagle.py:347-386,430-540)
        small = dict(H=128, Ht=96, I=192, nh=4, nkv=2, hd=64, Vt=640, Vd=256, B=1, S=48, lengths=[40], ttt=2,
                     dtype=torch.float32)
        run_case("eagle3_rope_yarn_fp32", rope_scaling=dict(rope_type="yarn", factor=4.0, beta_fast=32, beta_slow=1,
                                                             mscale=1.0, mscale_all_dim=0.5,
                                                             original_max_position_embeddings=32), seed=8, **small)
        run_case("eagle3_rope_dynamic_fp32", rope_scaling=dict(rope_type="dynamic", factor=2.0), seed=9, **small)
    if want("rope2"):  # linear position scaling (llama3_eagle.py:315-344) and multimodal 3-axis rope (145-182, 389-427)
        small = dict(H=128, Ht=96, I=192, nh=4, nkv=2, hd=64, Vt=640, Vd=256, B=2, S=40, lengths=[40, 29], ttt=3,
                     dtype=torch.float32)
        run_case("eagle3_rope_linear_fp32", rope_scaling=dict(rope_type="linear", factor=2.5), seed=10, **small)
        g = torch.Generator().manual_seed(11)
        # text-like prefix (all three axes equal), then an "image" span where the height / width axes move on their own
        t = torch.arange(40).repeat(2, 1)
        hh = t.clone()
        ww = t.clone()
        hh[:, 12:28] = 12 + torch.arange(16).div(4, rounding_mode="floor")
        ww[:, 12:28] = 12 + torch.arange(16) % 4
        tt = t.clone()
        tt[:, 12:28] = 12
        pos3 = torch.stack([tt, hh, ww]) + torch.tensor([0, 3]).view(1, 2, 1)      # second sample offset
        run_case("eagle3_rope_mrope_fp32", rope_scaling=dict(rope_type="mrope", mrope_section=[8, 12, 12]), seed=12,
                 position_ids=pos3, **small)
    if want("ropegrow"):  # sequences longer than the draft JSON's max_position_embeddings + 20: the rotary module rebuilds its cache
        # per TTT step (llama3_eagle.py:303-306 via seq_len = q_len + lck at 733); dynamic NTK re-derives its base from each rebuilt length
        small = dict(H=128, Ht=96, I=192, nh=4, nkv=2, hd=64, Vt=640, Vd=256, B=2, S=44, lengths=[44, 31], ttt=4,
                     dtype=torch.float32, max_pos=16)
        run_case("eagle3_rope_grow_fp32", seed=14, **small)
        run_case("eagle3_rope_grow_dynamic_fp32", rope_scaling=dict(rope_type="dynamic", factor=2.0), seed=15, **small)
    if want("hd256"):  # head_dim 256 (configs/gemma3-1b-eagle3.json: 4 / 1 heads; qwen3-next-80b-a3b, qwen3.5-35b-a3b: 16 / 2):
        # nh * hd > H like those recipes; the reference's attention takes any head_dim (llama3_eagle.py:547-550)
        run_case("eagle3_hd256_fp32", H=128, Ht=64, I=192, nh=2, nkv=1, hd=256, Vt=384, Vd=128, B=2, S=40, lengths=[40, 27],
                 ttt=4, dtype=torch.float32, seed=13)
