"""BASELINE-size checks (Llama-3-8B draft, seq 2048) through size-independent properties -- the CPU
oracle cannot run this size in test time, so the path is held to invariants of the algorithm:
run-to-run bitwise determinism, linearity of the backward in the upstream gradient, accumulation
== 2x, soft targets are distributions, position_mask = t2d[argmax] * loss_mask, ploss >= 0 and close
to ln(Vd) for a random-init draft, teacher ids == argmax of an independent torch GEMM."""
import math

import pytest
import torch

from bench import LLAMA3_8B, make_batch
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3


@pytest.mark.gpu
def test_llama3_8b_seq2048_invariants():
    dev = torch.device("cuda", 0)
    cfg, B, S, T = LLAMA3_8B, 2, 2048, 7
    torch.manual_seed(0)
    model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = OnlineEagle3Model(model, length=T).train()
    head_w = (torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16)
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w))
    raw = make_batch(cfg, B, S, dev, 7)
    raw["loss_mask"][:, :100] = 0                       # a prompt region without loss
    raw["attention_mask"][1, 1500:] = 0                 # right padding in sample 1
    raw["loss_mask"][1, 1499:] = 0
    batch = TrainBatch(raw, {"target_repr": "hidden_state"})
    eng = eagle.engine

    def run(scale):
        eng.micro_in_window = 0
        out = strat.forward_loss(batch)
        (out.loss * scale).backward()
        torch.cuda.synchronize()
        return out, eng.flat.grad.clone()

    out1, g1 = run(1.0)
    out2, g2 = run(1.0)
    assert torch.equal(g1, g2), "backward is not run-to-run deterministic"
    assert torch.equal(torch.stack(out1.metrics["plosses"]), torch.stack(out2.metrics["plosses"]))
    _, g3 = run(2.0)                                    # linear in the upstream gradient
    torch.testing.assert_close(g3.float(), 2 * g1.float(), rtol=1e-2, atol=1e-2 * float(g1.float().abs().max()))
    eng.micro_in_window = 0                             # accumulation: second micro-step adds
    strat.forward_loss(batch).loss.backward()
    strat.forward_loss(batch).loss.backward()
    torch.testing.assert_close(eng.flat.grad.float(), 2 * g1.float(), rtol=2e-2, atol=2e-2 * float(g1.float().abs().max()))
    assert torch.isfinite(g1.float()).all()

    b = eng._buffers(B, S)
    assert eng._soft[0] == "zt"                      # the headline path never materialises [B, S, Vd] fp32 soft targets
    tp = eng.soft_targets(B, S)
    torch.testing.assert_close(tp.sum(-1), torch.ones(B, S, device=dev), rtol=1e-4, atol=1e-4)   # distributions
    assert float(tp.min()) >= 0.0
    torch.testing.assert_close(b["tsum"][:, :S], tp.sum(-1), rtol=1e-5, atol=1e-6)
    assert float((b["tsum"][:, S:] - float(torch.full((cfg["draft_vocab_size"],), 1.0 / cfg["draft_vocab_size"]).sum())).abs().max()) == 0.0   # padded tail
    tid, pm, lm = b["tids"][:, :S], b["pm"][:, :S], b["lm"][:, :S]
    assert torch.equal(pm, (t2d.to(dev)[tid].int() * lm))                                          # bit-exact
    assert int(pm[:, :100].sum()) == 0 and int(pm[1, 1499:].sum()) == 0
    # teacher ids vs an independent torch GEMM + argmax (bf16 logits like TargetHead.forward)
    th = torch.cat((raw["target"][:, 1:], torch.zeros_like(raw["target"][:, -1:])), dim=1)        # preprocess shift
    z = torch.matmul(th.reshape(B * S, -1), head_w.t())
    agree = float((z.float().argmax(-1).view(B, S) == tid).float().mean())
    assert agree >= 0.999, agree
    pl = torch.stack(out1.metrics["plosses"]).float().cpu()
    assert (pl >= 0).all()
    frac = float(pm.float().mean())
    assert abs(float(pl[0]) - frac * math.log(cfg["draft_vocab_size"])) < 0.35 * frac * math.log(cfg["draft_vocab_size"])
    acc = torch.stack(out1.metrics["acces"]).float().cpu()
    assert ((acc >= 0) & (acc <= 1)).all()


@pytest.mark.gpu
def test_llama3_8b_bs8_k114688_step():
    """The bench's own shape (bs 8 x 2048: the weight-gradient GEMMs contract over K = 7 * 16384 = 114 688 rows, which is
    where the split-K + pace-keeping path of sf_gemm_tn switches on) as four copies of the B = 2 batch: the gradient is
    run-to-run bit-identical, ploss_k equals the B = 2 run's (a mean over the same rows, four times), and the flat gradient
    equals the B = 2 gradient (sum over 4x the rows at 1/4 the weight) to bf16 rounding."""
    dev = torch.device("cuda", 0)
    cfg, S, T = LLAMA3_8B, 2048, 7
    torch.manual_seed(0)
    model = LlamaForCausalLMEagle3(DraftConfig(**cfg), device=dev)
    ids = torch.randperm(cfg["vocab_size"], generator=torch.Generator().manual_seed(0))[:cfg["draft_vocab_size"]].sort().values
    t2d = torch.zeros(cfg["vocab_size"], dtype=torch.bool)
    t2d[ids] = True
    model.load_vocab_mapping_tensors(t2d, ids - torch.arange(cfg["draft_vocab_size"]))
    eagle = OnlineEagle3Model(model, length=T).train()
    head_w = (torch.randn(cfg["vocab_size"], cfg["target_hidden_size"], device=dev) * 0.02).to(torch.bfloat16)
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w))
    raw2 = make_batch(cfg, 2, S, dev, 7)
    raw2["loss_mask"][:, :100] = 0
    raw2["attention_mask"][1, 1500:] = 0
    raw2["loss_mask"][1, 1499:] = 0
    raw8 = {k: v.repeat((4,) + (1,) * (v.dim() - 1)).contiguous() for k, v in raw2.items()}
    eng = eagle.engine

    def run(raw):
        eng.micro_in_window = 0
        out = strat.forward_loss(TrainBatch(raw, {"target_repr": "hidden_state"}))
        out.loss.backward()
        torch.cuda.synchronize()
        return torch.stack(out.metrics["plosses"]).float().cpu(), eng.flat.grad.clone()

    pl8, g8 = run(raw8)
    pl8b, g8b = run(raw8)
    assert torch.equal(g8, g8b), "bs 8 backward (split-K + paced weight gradients at K = 114 688) is not deterministic"
    assert torch.equal(pl8, pl8b)
    pl2, g2 = run(raw2)
    torch.testing.assert_close(pl8, pl2, rtol=2e-5, atol=1e-6)
    rel = float((g8.float() - g2.float()).norm() / g2.float().norm())
    print(f"\n[bs8 vs bs2 gradient] relative Frobenius difference {rel:.3e}")
    assert rel < 2e-2, rel


# ------------------------------------------------------------------ numeric parity at the headline dimensions
LLAMA3_8B_CASE = dict(H=4096, Ht=4096, I=14336, nh=32, nkv=8, hd=128, Vt=128256, Vd=32000, B=2, S=2048, ttt=7, eps=1e-5,
                      max_pos=2048, rope_theta=LLAMA3_8B["rope_theta"], rope_scaling=LLAMA3_8B["rope_scaling"],
                      lengths=[2048, 1500], prompt=100)


@pytest.mark.gpu
def test_llama3_8b_seq2048_matches_oracle():
    """BASELINE configs[1] dims (Llama-3-8B draft, seq 2048, ttt 7; B=2 with a right-padded sample and a prompt region)
    against the pinned oracle run in fp32 on the same GPU: losses / acceptance 2e-2, acc_denoms bit-exact, teacher ids
    >= 99.9 %, every parameter gradient (report: gpurun_out/parity_cfg2_llama3_8b.json -> profiles/)."""
    from tests._parity import compare

    compare("cfg2_llama3_8b", LLAMA3_8B_CASE)


@pytest.mark.gpu
def test_llama3_8b_three_optimizer_steps_match_oracle_adamw():
    """>= 3 OPTIMIZER steps at the headline dimensions (Llama-3-8B draft, S 2048, B 2, three different
    batches) -- the HIP strategy + HipDPTrainingBackend + fused grad-norm / clip / AdamW over the flat buffers against the
    pinned oracle's autograd + the reference optimizer's arithmetic (optimizer.py:104-168: global norm of the gradients, clip
    coefficient clamp(max_norm / (norm + 1e-6), max = 1), ``torch.optim.AdamW`` on fp32 masters, bf16 parameters re-cast from
    them) run on the same GPU.  Truth = the oracle with the draft computing in fp32 from the bf16-rounded parameters;
    yardstick = the same loop with the draft computing in bf16 (what the reference itself does at that precision).
    Compared at every step: loss, plosses, grad_norm, learning rate; after the last: every tensor's fp32 MASTER update
    (w_3 - w_0), whose error must stay within 1.25x the yardstick's (Adam's normalised update amplifies sign noise of
    near-zero gradients alike on both sides) -- flat-buffer aliasing or a drifting fused AdamW would show as O(1)."""
    from oracle import eagle3_oracle as O
    from specforge_amd.training import BF16Optimizer, HipDPTrainingBackend
    from tests._parity import _cfg_kw, make_case

    dev = torch.device("cuda", 0)
    c = dict(LLAMA3_8B_CASE)
    ttt, steps, lr, max_norm = c["ttt"], 3, 2e-4, 0.5
    oc, params, embed, head_w, t2d, d2t, _ = make_case(c)
    batches = [O.make_batch(oc, c["B"], c["S"], seed=40 + i, dtype=torch.bfloat16, lengths=[2048, 1500 + 100 * i]) for i in range(steps)]

    def oracle_loop(dtype):
        masters = {k: v.to(dev).float().requires_grad_(True) for k, v in params.items()}
        opt = torch.optim.AdamW(list(masters.values()), lr=lr, weight_decay=0.0)
        hist = []
        for b in batches:
            p = {k: m.detach().to(torch.bfloat16).to(dtype).requires_grad_(True) for k, m in masters.items()}   # bf16 parameters
            out = O.eagle3_forward(p, oc, embed_weight=embed.to(dev).to(dtype), target_head_weight=head_w.to(dev), t2d=t2d.to(dev),
                                   d2t=d2t.to(dev), input_ids=b["input_ids"].to(dev), attention_mask=b["attention_mask"],
                                   loss_mask=b["loss_mask"].to(dev), hidden_state=b["hidden_state"].to(dev).to(dtype),
                                   target_hidden=b["target"].to(dev), ttt_length=ttt)
            out.loss.backward()
            grads = {k: (v.grad.to(torch.bfloat16) if dtype == torch.bfloat16 else v.grad).float() for k, v in p.items()}
            norm = torch.stack([g.square().sum() for g in grads.values()]).sum().sqrt()
            coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
            for k, m in masters.items():
                m.grad = grads[k] * coef
            opt.step()
            opt.zero_grad()
            hist.append(dict(loss=float(out.loss), plosses=[float(x) for x in out.plosses], grad_norm=float(norm)))
            del p, out, grads
            torch.cuda.empty_cache()
        return hist, {k: m.detach().clone() for k, m in masters.items()}

    truth_hist, truth_w = oracle_loop(torch.float32)
    yard_hist, yard_w = oracle_loop(torch.bfloat16)

    model = LlamaForCausalLMEagle3(DraftConfig(**_cfg_kw(c)), device=dev)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=ttt).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(dev)))
    backend = HipDPTrainingBackend(optimizer_factory=lambda m: BF16Optimizer(m, lr=lr, max_grad_norm=max_norm, warmup_ratio=0.0,
                                                                             lr_scheduler="constant", total_steps=1000))
    backend.prepare_model(eagle)
    hip_hist = []
    for b in batches:
        out = strat.forward_loss(TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                                                 hidden_state=b["hidden_state"].to(dev), target=b["target"].to(dev)),
                                            {"target_repr": "hidden_state"}))
        backend.backward(out.loss, is_boundary=True)
        assert abs(backend.optimizer.get_learning_rate() - lr) < 1e-12
        gn = backend.step()
        hip_hist.append(dict(loss=float(out.loss), plosses=[float(x) for x in out.metrics["plosses"]], grad_norm=float(gn)))
    f, master = eagle.engine.flat, backend.optimizer.master
    rep = []
    for i, (h, t, y) in enumerate(zip(hip_hist, truth_hist, yard_hist)):
        rep.append((i, h["loss"], t["loss"], y["loss"], h["grad_norm"], t["grad_norm"], y["grad_norm"]))
        assert abs(h["loss"] - t["loss"]) <= 5e-3 * max(1.0, abs(t["loss"])), rep[-1]
        assert max(abs(a - b) for a, b in zip(h["plosses"], t["plosses"])) <= 5e-3, (i, h["plosses"], t["plosses"])
        assert abs(h["grad_norm"] - t["grad_norm"]) <= max(2e-2, 1.25 * abs(y["grad_norm"] - t["grad_norm"])) * t["grad_norm"], rep[-1]
    print("\n[3 optimizer steps, Llama-3-8B dims] (step, loss hip / fp32 / bf16, grad_norm hip / fp32 / bf16):", rep)
    worst = {}
    for k, w0 in params.items():
        lo, hi = f.slices[k]
        w0 = w0.to(dev).float()
        d_hip, d_true, d_yard = master[lo:hi].view(w0.shape) - w0, truth_w[k] - w0, yard_w[k] - w0
        e_hip = float((d_hip - d_true).norm() / d_true.norm().clamp_min(1e-20))
        e_yard = float((d_yard - d_true).norm() / d_true.norm().clamp_min(1e-20))
        worst[k] = (round(e_hip, 4), round(e_yard, 4))
        # the update has the right size everywhere (an aliasing / slicing bug moves the wrong elements)
        assert 0.5 <= float(d_hip.norm() / d_true.norm().clamp_min(1e-20)) <= 2.0, (k, worst[k])
        assert e_hip <= max(0.05, 1.25 * e_yard), (k, worst[k])
        # the bf16 parameters ARE the rounded masters (flat data aliases the module's parameters)
        assert torch.equal(dict(model.named_parameters())[k].detach(), master[lo:hi].view(w0.shape).to(torch.bfloat16)), k
    print("[3 optimizer steps] master update error vs fp32 truth, relative Frobenius (hip, bf16 yardstick):", worst)
