"""End-to-end parity of the HIP EAGLE3 micro-step against vectors produced by RUNNING THE REAL
REFERENCE (oracle/gen_golden.py -> tests/golden/*.pt): strategy.forward_loss -> loss.backward()
through the C-ABI, compared on losses, metrics, integer artefacts (bit-exact) and every
parameter gradient.  ``[emu]`` runs the kernels under the SIMT interpreter (CPU), ``[gpu]``
is the product library on an MI355X.

Tolerances (BASELINE.json north_star): integer artefacts bit-exact; bf16 path 2e-2 on
losses/metrics; gradients within 5e-2 of each tensor's max (same bar tests/test_oracle_golden.py
uses for the reference's own bf16 run vs the fp32 oracle).
"""
import os

import pytest
import torch

from oracle import eagle3_oracle as O
from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3


def _build(blob, dev):
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"],
                      rope_scaling=c["rope_scaling"], norm_output=c.get("norm_output", True))
    model = LlamaForCausalLMEagle3(cfg, device=dev)
    sd = {k: v.to(torch.bfloat16) for k, v in blob["params"].items()}
    sd["embed_tokens.weight"] = blob["embed"].to(torch.bfloat16)
    sd["t2d"], sd["d2t"] = blob["t2d"], blob["d2t"]
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    eagle = OnlineEagle3Model(model, length=c["ttt"], lk_loss_type=c.get("lk_loss_type"), kl_scale=c.get("kl_scale", 1.0),
                              kl_decay=c.get("kl_decay", 1.0))
    head = TargetHead(blob["head_w"].to(torch.bfloat16).to(dev))
    return cfg, model, eagle, Eagle3TrainStrategy(eagle, target_head=head)


def _batch(blob, dev):
    b = blob["batch"]
    return TrainBatch(tensors=dict(
        input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
        hidden_state=b["hidden_state"].to(torch.bfloat16).to(dev), target=b["target"].to(torch.bfloat16).to(dev),
        **({"position_ids": b["position_ids"]} if "position_ids" in b else {})),
        metadata={"target_repr": "hidden_state"})


def _oracle_bf16(blob):
    """The fp32 goldens hold fp32 weights/inputs; the product path computes in bf16.  Expected values for
    them = the pinned oracle (== the reference, tests/test_oracle_golden.py) run in bf16 on the bf16-cast
    golden inputs -- i.e. what the reference itself produces at that precision."""
    c = blob["cfg"]
    cfg = O.DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                        vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                        max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"],
                        rope_scaling=c["rope_scaling"], norm_output=c.get("norm_output", True))
    bf = torch.bfloat16
    p = {k: v.to(bf).clone().requires_grad_(True) for k, v in blob["params"].items()}
    b = blob["batch"]
    out = O.eagle3_forward(p, cfg, embed_weight=blob["embed"].to(bf), target_head_weight=blob["head_w"].to(bf), t2d=blob["t2d"],
                           d2t=blob["d2t"], input_ids=b["input_ids"], attention_mask=b["attention_mask"],
                           loss_mask=b["loss_mask"], hidden_state=b["hidden_state"].to(bf), target_hidden=b["target"].to(bf),
                           ttt_length=c["ttt"], lk_loss_type=c.get("lk_loss_type"), kl_scale=c.get("kl_scale", 1.0),
                           kl_decay=c.get("kl_decay", 1.0), position_ids=b.get("position_ids"))
    out.loss.backward()
    new = dict(blob)
    new.update(plosses=torch.stack([x.detach().float() for x in out.plosses]), loss=out.loss.detach().float(),
               acces=torch.stack(out.acces).float(), acceptance_rates=torch.stack(out.acceptance_rates).float(),
               acc_denoms=torch.stack(out.acc_denoms).float(), target_token_ids=out.target_token_ids,
               position_mask=out.position_mask,
               grads={k: (v.grad.detach() if v.grad is not None else torch.zeros_like(v)) for k, v in p.items()})
    return new


@pytest.mark.parametrize("name", ["eagle3_tiny_bf16", "eagle3_tiny_fp32", "eagle31_gqa_fp32", "eagle3_lk_alpha_fp32",
                                  "eagle3_lk_lambda_fp32", "eagle3_nonorm_fp32", "eagle3_rope_yarn_fp32",
                                  "eagle3_rope_dynamic_fp32", "eagle3_rope_linear_fp32", "eagle3_rope_mrope_fp32", "eagle3_hd256_fp32",
                                  # sequences longer than max_position_embeddings + 20 (the RoPE table grows like the reference's cache)
                                  "eagle3_rope_grow_fp32", "eagle3_rope_grow_dynamic_fp32"])
def test_micro_step_matches_reference_run(backend, golden_dir, name):
    blob = torch.load(os.path.join(golden_dir, f"{name}.pt"), weights_only=False)
    raw = blob
    if "fp32" in name:
        blob = _oracle_bf16(blob)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    out = strat.forward_loss(_batch(blob, backend))
    out.loss.backward()
    T = blob["cfg"]["ttt"]
    if raw is not blob:
        # first against the RAW values the reference itself produced in fp32 -- no port in between.  The HIP path ran on the bf16-ROUNDED
        # weights and inputs (the fp32 goldens' tensors are not bf16-representable), and rounding the teacher head breaks argmax near-ties
        # (2 of 64 positions in the tiny case); one flipped id moves a row's whole soft target, i.e. a `ploss` by ~0.1.  So: ids >= 95 %, and
        # where NO id flipped the losses meet the raw reference numbers at north_star's bf16 bar.  (bit-exact ids / 5e-3 losses against the
        # reference's own run on bf16-representable inputs at REAL dims: tests/test_reference_realdims.py; against this golden's inputs in
        # bf16: right below.)
        ids_raw = eagle.last_artifacts["target_token_ids"].cpu()
        assert float((ids_raw == raw["target_token_ids"]).float().mean()) >= 0.95
        if torch.equal(ids_raw, raw["target_token_ids"]):
            torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(), raw["plosses"], rtol=2e-2, atol=2e-2)
            torch.testing.assert_close(out.loss.detach().float().cpu(), raw["loss"], rtol=2e-2, atol=2e-2)
            assert torch.equal(torch.stack(out.metrics["acc_denoms"]).cpu(), raw["acc_denoms"])
    # integer artefacts: bit-exact
    ids = eagle.last_artifacts["target_token_ids"].cpu()
    pm = eagle.last_artifacts["position_mask"].cpu()
    assert torch.equal(ids, blob["target_token_ids"])
    assert torch.equal(pm[..., None].int(), blob["position_mask"].int())
    assert torch.equal(torch.stack(out.metrics["acc_denoms"]).cpu(), blob["acc_denoms"])
    tol = 2e-2
    pl = torch.stack(out.metrics["plosses"]).float().cpu()
    torch.testing.assert_close(pl, blob["plosses"], rtol=tol, atol=tol)
    torch.testing.assert_close(out.loss.detach().float().cpu(), blob["loss"], rtol=tol, atol=tol)
    acc = torch.stack(out.metrics["acces"]).float().cpu()
    # accuracy = correct / denominator: the draft's argmax may resolve a bf16 near-tie differently on at most ONE token per step
    assert float(((acc - blob["acces"]).abs() * blob["acc_denoms"]).max()) <= 1.0 + 1e-3, (acc, blob["acces"])
    ar = torch.stack(out.metrics["acceptance_rates"]).float().cpu()
    torch.testing.assert_close(ar, blob["acceptance_rates"], rtol=tol, atol=tol)
    assert torch.stack(out.metrics["metric_loss_denoms"]).cpu().tolist() == [float(blob["batch"]["input_ids"].numel())] * T
    named = dict(model.named_parameters())
    worst = {}
    for k, g in blob["grads"].items():
        if named[k].grad is None:       # norm_output=False: `norm` is never applied, its .grad stays None like the reference's
            assert k == "norm.weight" and not blob["cfg"].get("norm_output", True) and float(g.abs().max()) == 0.0
            continue
        got = named[k].grad.float().cpu()
        scale = float(g.float().abs().max().clamp_min(1e-8))
        worst[k] = float((got - g.float()).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if v > 5e-2}
    assert not bad, worst


@pytest.mark.parametrize("ttt", [10, 13])
def test_long_ttt_unroll_matches_pinned_oracle(backend, golden_dir, ttt):
    """ttt_length above the reference's default 7 at a native head_dim: the blocked diagonal backward then plans several blocks of
    branches, steps with more than 6 branches (dq in two or three launches) and streamed-step lists; expected values = the pinned
    oracle in bf16 on the golden's inputs with the longer unroll"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_fp32.pt"), weights_only=False)
    blob["cfg"] = dict(blob["cfg"], ttt=ttt)
    blob = _oracle_bf16(blob)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    out = strat.forward_loss(_batch(blob, backend))
    out.loss.backward()
    assert len(out.metrics["plosses"]) == ttt and max(len(v) for v in eagle.engine._diag_plan.values()) >= 2
    assert torch.equal(eagle.last_artifacts["target_token_ids"].cpu(), blob["target_token_ids"])
    torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(), blob["plosses"], rtol=2e-2, atol=2e-2)
    named = dict(model.named_parameters())
    worst = {k: float((named[k].grad.float().cpu() - g.float()).abs().max()) / float(g.float().abs().max().clamp_min(1e-8))
             for k, g in blob["grads"].items()}
    assert max(worst.values()) <= 5e-2, worst


def test_blocked_and_per_step_diagonal_backward_agree(backend, golden_dir):
    """engine.blocked_diag (round 4: sf_attn_bwd_diag over engine.diag_plan -- blocks of branches gathered at their top step, later
    steps streamed, first-touch sums) vs the per-step form (one sf_attn_bwd_pre per TTT step): the same gradient up to fp32
    summation order, on the ttt-7 golden (blocks {1..4} and {5, 6}) -- and the blocked run is bit-reproducible"""
    blob = torch.load(os.path.join(golden_dir, "eagle31_gqa_fp32.pt"), weights_only=False)
    grads = []
    for blocked in (True, False, True):
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        eagle.engine.blocked_diag = blocked
        strat.forward_loss(_batch(blob, backend)).loss.backward()
        grads.append(eagle.engine.flat.grad.float().cpu().clone())
    assert torch.equal(grads[0], grads[2])
    torch.testing.assert_close(grads[0], grads[1], rtol=2e-2, atol=4e-3 * float(grads[1].abs().max()))


@pytest.mark.parametrize("golden", ["eagle31_gqa_fp32.pt", "eagle3_nonorm_fp32.pt"])
def test_one_column_sum_per_norm_weight_equals_one_per_launch(backend, golden_dir, golden):
    """engine.norm_colsum_batched (round 5): the per-block partials of the three norm weights that every TTT step differentiates are kept side
    by side and reduced ONCE per weight after the sweep (3 column sums instead of 21) -- every other gradient bit-identical to the
    per-launch form, the three norm gradients equal up to fp32 summation order; two accumulated micro-steps; bit-reproducible.
    (second golden: norm_output = false, the final norm has no gradient and no partials)"""
    blob = torch.load(os.path.join(golden_dir, golden), weights_only=False)
    grads = []
    for batched in (True, False, True):
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        eagle.engine.norm_colsum_batched = batched
        for _ in range(2):
            strat.forward_loss(_batch(blob, backend)).loss.backward()
        grads.append(eagle.engine.flat.grad.float().cpu().clone())
        eagle.engine.end_window()
    assert torch.equal(grads[0], grads[2])
    f = eagle.engine.flat
    norm = torch.zeros(f.numel, dtype=torch.bool)
    for n in eagle.engine._NORMS_PER_STEP:
        norm[f.slices[n][0]:f.slices[n][1]] = True
    assert torch.equal(grads[0][~norm], grads[1][~norm])
    torch.testing.assert_close(grads[0][norm], grads[1][norm], rtol=2e-2, atol=1e-2 * float(grads[1][norm].abs().max()))     # (the flat gradient is bf16)


@pytest.mark.parametrize("mask", ["random", "head_only", "all_zero"])
def test_loss_row_compaction_equals_the_dense_form(backend, golden_dir, mask):
    """engine.compact_loss_rows (round 4): lm_head / CE / lm_head gradients over the rows with loss_mask[b, s + k] != 0 only, from
    host-side row counts -- same losses, metrics and gradients as the dense form (masked rows contribute exact zeros there).
    ``head_only``: the mask ends at position 2, so TTT steps 3.. have no row at all; wrong counts are refused in backward."""
    blob = torch.load(os.path.join(golden_dir, "eagle31_gqa_fp32.pt"), weights_only=False)
    lm = blob["batch"]["loss_mask"].clone()
    if mask == "random":
        lm = (torch.rand(lm.shape, generator=torch.Generator().manual_seed(5)) < 0.4).to(lm.dtype)
    elif mask == "head_only":
        lm.zero_()
        lm[:, :3] = 1
    elif mask == "all_zero":          # (a batch without any loss: every step compacts to zero rows, the loss and every gradient are zero)
        lm.zero_()
    blob["batch"] = dict(blob["batch"], loss_mask=lm)
    runs = []
    for compact in (True, False):
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        eagle.engine.compact_loss_rows = compact
        out = strat.forward_loss(_batch(blob, backend))
        assert (eagle.engine._lm_compact_K is not None) == (compact and float(lm.float().mean()) < 0.85)
        out.loss.backward()
        runs.append((out, eagle.engine.flat.grad.float().cpu().clone()))
    (a, ga), (d, gd) = runs
    if mask == "all_zero":
        assert float(ga.abs().max()) == 0.0 and float(gd.abs().max()) == 0.0 and float(a.loss.detach()) == 0.0
    for key in ("plosses", "acces", "acceptance_rates", "acc_corrects", "acc_denoms"):
        torch.testing.assert_close(torch.stack(a.metrics[key]).float().cpu(), torch.stack(d.metrics[key]).float().cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ga, gd, rtol=2e-2, atol=4e-3 * float(gd.abs().max()))
    if mask == "random":
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        batch = _batch(blob, backend)
        from specforge_amd.eagle3 import loss_mask_suffix_counts
        wrong = loss_mask_suffix_counts(lm)
        wrong[2] += 1
        batch.metadata["loss_mask_suffix_counts"] = wrong
        with pytest.raises(RuntimeError, match="loss_counts"):
            strat.forward_loss(batch).loss.backward()
        # a count of ZERO where rows do carry a loss (a step -- or the teacher -- would be skipped silently): refused as well
        for zeroed in ([1, 2], [0]):
            cfg, model, eagle, strat = _build(blob, backend)
            eagle.train()
            batch = _batch(blob, backend)
            wrong = loss_mask_suffix_counts(lm)
            for k in zeroed:
                assert wrong[k] > 0
                wrong[k] = 0
            batch.metadata["loss_mask_suffix_counts"] = wrong
            with pytest.raises(RuntimeError, match="loss_counts"):
                strat.forward_loss(batch).loss.backward()
        # ... and an eval forward (no backward to read the flag in) checks it itself
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        wrong = loss_mask_suffix_counts(lm)
        wrong[1] = 0
        b0 = _batch(blob, backend).tensors
        from specforge_amd.eagle3 import TargetHead as TH
        ids, tgt, lmask = TH.preprocess(b0["input_ids"], b0["target"], b0["loss_mask"])
        kw = dict(input_ids=ids.to(backend), attention_mask=b0["attention_mask"], loss_mask=lmask.to(backend),
                  hidden_states=b0["hidden_state"], target_hidden=tgt, target_head_weight=strat.target_head.fc.weight.data)
        eagle.engine.forward(train=False, loss_counts=loss_mask_suffix_counts(lm), **kw)       # eval never compacts: counts unused, fine
        with pytest.raises(RuntimeError, match="loss_counts"):
            eagle.engine.forward(train=True, loss_counts=wrong, **kw)
            eagle.engine.backward(lambda: eagle.engine.check_flags(eagle.engine._flags.tolist()) or 1.0)


def test_position_ids_outside_the_reference_range_are_refused(backend, golden_dir):
    """plain [B, S] position ids: the reference indexes cos[:S + k] at ids + k (llama3_eagle.py:303-311, 134-139), so an id >= S is an
    IndexError there.  Here: host ids are checked on the host (IndexError), device ids by a device flag read back at the end of the
    backward sweep (RuntimeError) -- never a rotation by a clamped angle.  In-range custom ids still run and equal the default ids."""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_fp32.pt"), weights_only=False)
    B, S = blob["batch"]["input_ids"].shape
    ok = torch.arange(S).repeat(B, 1)
    bad = ok.clone()
    bad[1, S - 1] = S                                   # one id past the end
    neg = ok.clone()
    neg[0, 0] = -1
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    base = strat.forward_loss(_batch(blob, backend))
    base.loss.backward()
    g0 = eagle.engine.flat.grad.clone()
    eagle.engine.end_window()                            # (what the optimizer does: the next backward overwrites flat.grad)
    # host tensors
    batch = _batch(blob, backend)
    batch.tensors["position_ids"] = ok
    out = strat.forward_loss(batch)
    out.loss.backward()
    assert torch.equal(eagle.engine.flat.grad, g0) and float(out.loss.detach()) == float(base.loss.detach())
    for ids in (bad, neg):
        batch = _batch(blob, backend)
        batch.tensors["position_ids"] = ids
        with pytest.raises(IndexError, match="position_ids"):
            strat.forward_loss(batch)
    # ids that are already on the engine's device when they reach it (no host copy to look at): the device flag
    b0 = _batch(blob, backend).tensors
    from specforge_amd.eagle3 import TargetHead as TH
    ids_in, tgt, lmask = TH.preprocess(b0["input_ids"], b0["target"], b0["loss_mask"])
    kw = dict(input_ids=ids_in.to(backend), attention_mask=b0["attention_mask"], loss_mask=lmask.to(backend),
              hidden_states=b0["hidden_state"], target_hidden=tgt, target_head_weight=strat.target_head.fc.weight.data)
    eng = eagle.engine
    if backend != "cpu":            # (under the interpreter every tensor is a host tensor: the host check above is what runs)
        eng.forward(train=True, position_ids=bad.to(backend), **kw)
        with pytest.raises(RuntimeError, match="position_ids"):
            eng.backward(lambda: eng.check_flags(eng._flags.tolist()) or 1.0)
        eng.forward(train=True, position_ids=ok.to(backend), **kw)       # the flag is cleared by the next forward
        eng.backward(lambda: eng.check_flags(eng._flags.tolist()) or 1.0)
        with pytest.raises(RuntimeError, match="position_ids"):
            eng.forward(train=False, position_ids=bad.to(backend), **kw)   # eval: checked at the end of the forward
    with pytest.raises(IndexError, match="position_ids"):
        eng.forward(train=True, position_ids=bad.to(backend), position_span=(0, S), **kw)   # a caller-supplied host span


@pytest.mark.parametrize("rope_scaling", [None, dict(rope_type="dynamic", factor=2.0), dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0,
                                                                                         high_freq_factor=4.0, original_max_position_embeddings=64)])
def test_rope_table_grows_like_the_reference_cache(backend, rope_scaling):
    """The rotary module of the reference rebuilds its cache when seq_len = S + k exceeds it and KEEPS the rebuilt one
    (llama3_eagle.py:303-306): the engine's per-step tables must equal the oracle's RopeCache (pinned to the reference by the
    eagle3_rope_grow* goldens) row for row, over a sequence of forwards with growing, shrinking and repeated lengths."""
    kw = dict(hidden_size=64, intermediate_size=64, num_attention_heads=2, num_key_value_heads=1, vocab_size=64, draft_vocab_size=32,
              head_dim=64, target_hidden_size=64, max_position_embeddings=16, rms_norm_eps=1e-5, rope_scaling=rope_scaling)
    from specforge_amd.engine import Eagle3Engine
    model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
    eng = Eagle3Engine(model, ttt_length=4)
    rc = O.RopeCache(O.DraftConfig(**kw), torch.bfloat16, "cpu")
    for S in (20, 33, 40, 38, 40, 120, 64, 3000, 20):
        tabs = eng._rope_steps(S)
        for k in range(4):
            cos, sin = rc.get(S + k)
            assert tabs[k][0].shape[0] >= S + k
            assert torch.equal(tabs[k][0][:S + k].cpu(), cos) and torch.equal(tabs[k][1][:S + k].cpu(), sin), (S, k)


@pytest.mark.parametrize("B,S,density", [(3, 50, 0.5), (1, 37, 0.3)])
def test_loss_row_compaction_on_odd_shapes(backend, golden_dir, B, S, density):
    """the compact form on batch shapes with nothing aligned (B * S not a multiple of 64, ragged lengths, a random mask): same step as dense"""
    blob = torch.load(os.path.join(golden_dir, "eagle31_gqa_fp32.pt"), weights_only=False)
    c = blob["cfg"]
    oc = O.DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                       vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                       max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"], rope_scaling=c["rope_scaling"],
                       norm_output=c.get("norm_output", True))
    g = torch.Generator().manual_seed(B * 100 + S)
    lens = [S] + [int(torch.randint(S // 2, S + 1, (1,), generator=g)) for _ in range(B - 1)]
    b = O.make_batch(oc, B, S, seed=B + S, dtype=torch.float32, lengths=lens)
    b["loss_mask"] = b["loss_mask"] * (torch.rand(B, S, generator=g) < density).to(b["loss_mask"].dtype)
    blob["batch"] = {k: b[k] for k in ("input_ids", "attention_mask", "loss_mask", "hidden_state", "target")}
    runs = []
    for compact in (True, False):
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        eagle.engine.compact_loss_rows = compact
        out = strat.forward_loss(_batch(blob, backend))
        assert (eagle.engine._lm_compact_K is not None) == compact
        out.loss.backward()
        runs.append((out, eagle.engine.flat.grad.float().cpu().clone(), eagle.last_artifacts["position_mask"].cpu()))
    (a, ga, pa), (d, gd, pd) = runs
    assert torch.equal(pa, pd)
    for key in ("plosses", "acces", "acceptance_rates", "acc_corrects", "acc_denoms"):
        torch.testing.assert_close(torch.stack(a.metrics[key]).float().cpu(), torch.stack(d.metrics[key]).float().cpu(), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(ga, gd, rtol=2e-2, atol=4e-3 * float(gd.abs().max()))


def test_accumulation_window_and_eval_mode(backend, golden_dir):
    """two micro-steps accumulate (DDP no_sync semantics, training/backend.py:310-320); eval forward leaves no state"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    (strat.forward_loss(_batch(blob, backend)).loss / 2).backward()
    g1 = eagle.engine.flat.grad.float().clone()
    (strat.forward_loss(_batch(blob, backend)).loss / 2).backward()
    g2 = eagle.engine.flat.grad.float()
    torch.testing.assert_close(g2, 2 * g1, rtol=2e-2, atol=2e-2 * float(g1.abs().max()))
    eagle.eval()
    with torch.no_grad():
        out = strat.forward_loss(_batch(blob, backend))
    assert not out.loss.requires_grad
    torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(), blob["plosses"], rtol=2e-2, atol=2e-2)


def test_foreign_ploss_weights_are_refused(backend, golden_dir):
    """the decay weights are baked into the fused CE gradients at forward time; a loss that combines the per-step losses
    with other weights must fail loudly (the check runs where the upstream gradient is read back: the end of the sweep)"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    t = _batch(blob, backend).tensors
    ids, th, lm = TargetHead.preprocess(t["input_ids"], t["target"], t["loss_mask"])
    plosses = eagle(input_ids=ids, attention_mask=t["attention_mask"], loss_mask=lm, target=None, hidden_states=t["hidden_state"],
                    target_hidden_for_compact=th, target_head_weight=strat.target_head.fc.weight.data)[0]
    wrong = sum((0.5 ** k) * p for k, p in enumerate(plosses))      # the model was built with ploss_decay 0.8
    with pytest.raises(RuntimeError, match="ploss weights"):
        wrong.backward()


def test_logits_teacher_path_equals_hidden_state_path(backend, golden_dir):
    """target_repr == logits: online capture has ALREADY shifted logits and input ids, so the strategy uses them exactly
    as delivered (_prepare_eagle_target, strategies/base.py:95-121).  Feeding the head's own bf16 logits of the shifted
    hidden state together with the shifted ids must give the same step as the offline hidden-state path (which shifts
    inside TargetHead.preprocess)."""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    batch = _batch(blob, backend)
    out_h = strat.forward_loss(batch)
    ids_h = eagle.last_artifacts["target_token_ids"].clone()
    from specforge_amd.eagle3 import padding_left_shift

    t = dict(batch.tensors)
    t["input_ids"] = padding_left_shift(t["input_ids"])
    t["target"] = torch.nn.functional.linear(padding_left_shift(t["target"].cpu().to(torch.bfloat16)),
                                             blob["head_w"].to(torch.bfloat16)).to(backend)
    out_l = strat.forward_loss(TrainBatch(t, {"target_repr": "logits"}))
    assert torch.equal(eagle.last_artifacts["target_token_ids"], ids_h)
    torch.testing.assert_close(torch.stack(out_l.metrics["plosses"]), torch.stack(out_h.metrics["plosses"]), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("B,S,lengths", [(2, 13, [13, 7]), (1, 27, [27])])
def test_sequence_length_not_a_multiple_of_8(backend, B, S, lengths):
    """The reference collator right-pads to the longest sample of the batch, whatever that length is
    (data/utils.py:122-142); ploss_k is a mean over ALL B*S rows (core/loss.py:20,201), so padding S any further
    would rescale every loss.  One micro-step at odd S vs the pinned oracle in bf16."""
    kw = dict(hidden_size=64, intermediate_size=96, num_attention_heads=2, num_key_value_heads=1, vocab_size=304,
              draft_vocab_size=72, head_dim=64, target_hidden_size=48, max_position_embeddings=64, rms_norm_eps=1e-5)
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    params = {k: v.to(bf) for k, v in O.init_params(oc, seed=5).items()}
    g = torch.Generator().manual_seed(6)
    for k, v in params.items():
        if v.dim() == 1:
            params[k] = (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
        else:
            params[k] = (v.float() * 4).to(bf)
    embed = (torch.randn(304, 64, generator=g) * 0.5).to(bf)
    head_w = (torch.randn(304, 48, generator=g) * 0.5).to(bf)
    t2d, d2t = O.make_vocab_mapping(304, 72, seed=3)
    batch = O.make_batch(oc, B, S, seed=4, dtype=bf, lengths=lengths)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.eagle3_forward(p, oc, embed_weight=embed, target_head_weight=head_w, t2d=t2d, d2t=d2t, input_ids=batch["input_ids"],
                           attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"],
                           hidden_state=batch["hidden_state"], target_hidden=batch["target"], ttt_length=3)
    ref.loss.backward()
    model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
    sd = dict(params)
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=3).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(backend)))
    out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                             loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"].to(backend),
                                             target=batch["target"].to(backend)), {"target_repr": "hidden_state"}))
    out.loss.backward()
    assert torch.equal(eagle.last_artifacts["target_token_ids"].cpu(), ref.target_token_ids)
    assert torch.stack(out.metrics["metric_loss_denoms"]).cpu().tolist() == [float(B * S)] * 3    # B*S, not a padded count
    torch.testing.assert_close(torch.stack(out.metrics["acc_denoms"]).cpu(), torch.stack(ref.acc_denoms).float())
    torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(),
                               torch.stack([x.detach().float() for x in ref.plosses]), rtol=2e-2, atol=2e-2)
    named = dict(model.named_parameters())
    worst = {k: float((named[k].grad.float().cpu() - v.grad.float()).abs().max() / v.grad.float().abs().max().clamp_min(1e-8))
             for k, v in p.items()}
    assert max(worst.values()) <= 5e-2, worst


def test_unmaterialised_soft_targets_equal_the_materialised_path(backend):
    """A shape the reduced teacher-head GEMM takes (>= 192 rows, target hidden size 512, Vt past roundup(Vd, 256)): the engine keeps
    the teacher's draft logits + (max, 1 / sum-exp) per row and the fused CE re-forms target_p from them; [B, S, Vd] fp32 never exists.
    Same micro-step with that path switched off (materialised target_p): ids bit for bit, metrics 2e-6, gradients equal up to rare last-ulp roundings; ids vs the oracle."""
    from specforge_amd import ops
    kw = dict(hidden_size=64, intermediate_size=96, num_attention_heads=2, num_key_value_heads=1, vocab_size=640,
              draft_vocab_size=256, head_dim=64, target_hidden_size=512, max_position_embeddings=128, rms_norm_eps=1e-5)
    B, S, T = 2, 96, 2
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(8)
    params = {k: (v.float() * 4).to(bf) if v.dim() > 1 else (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
              for k, v in O.init_params(oc, seed=7).items()}
    embed = (torch.randn(640, 64, generator=g) * 0.5).to(bf)
    head_w = (torch.randn(640, 512, generator=g) * 0.2).to(bf)
    t2d, d2t = O.make_vocab_mapping(640, 256, seed=5)
    batch = O.make_batch(oc, B, S, seed=9, dtype=bf, lengths=[S, 71])

    def run(materialise):
        model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
        sd = dict(params)
        sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
        model.load_state_dict(sd)
        eagle = OnlineEagle3Model(model, length=T).train()
        strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(backend)))
        orig = ops.gemm_nt_teacher_reduces
        if materialise:
            ops.gemm_nt_teacher_reduces = lambda *a: False
        try:
            out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"],
                                                     loss_mask=batch["loss_mask"], hidden_state=batch["hidden_state"].to(backend),
                                                     target=batch["target"].to(backend)), {"target_repr": "hidden_state"}))
            out.loss.backward()
        finally:
            ops.gemm_nt_teacher_reduces = orig
        kind = eagle.engine._soft[0]
        mets = {k: torch.stack(v).float().cpu() for k, v in out.metrics.items() if isinstance(v, list)}
        return kind, mets, eagle.engine.flat.grad.clone().cpu(), eagle.last_artifacts["target_token_ids"].cpu(), \
            eagle.engine.soft_targets(B, S).cpu()

    kind_a, met_a, grad_a, ids_a, tp_a = run(False)
    kind_b, met_b, grad_b, ids_b, tp_b = run(True)
    if str(backend) == "cpu":          # (the interpreter takes the reduced GEMM for every long-K shape; a GPU only for chip-filling ones)
        assert (kind_a, kind_b) == ("zt", "tp")
    # (round 6: the un-materialised form takes sum(target_p) as sd * (1 / sd) instead of re-summing Vd terms, so the two forms differ in the
    #  last bits of that row scalar: integer artefacts exact, metrics to 1e-6, bf16 gradients equal except for a rare last-ulp rounding)
    assert torch.equal(ids_a, ids_b)
    dg = (grad_a.float() - grad_b.float()).abs()
    assert float((dg > 0).float().mean()) <= 2e-3 and float(dg.max()) <= 2 ** -7 * float(grad_b.float().abs().max())
    for k in met_a:
        torch.testing.assert_close(met_a[k], met_b[k], rtol=2e-6, atol=1e-7, msg=k)
    torch.testing.assert_close(tp_a, tp_b, rtol=1e-6, atol=1e-9)         # (torch.exp vs the kernels' v_exp_f32 in the inspection helper)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    ref = O.eagle3_forward(p, oc, embed_weight=embed, target_head_weight=head_w, t2d=t2d, d2t=d2t, input_ids=batch["input_ids"],
                           attention_mask=batch["attention_mask"], loss_mask=batch["loss_mask"],
                           hidden_state=batch["hidden_state"], target_hidden=batch["target"], ttt_length=T)
    assert float((ids_a == ref.target_token_ids).float().mean()) >= 0.98          # (bf16 accumulation-order near-ties of the head GEMM)
    torch.testing.assert_close(met_a["plosses"], torch.stack([x.detach().float() for x in ref.plosses]), rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("B,S,teacher_rows", [(4, 250, 400)])     # (S does not divide teacher_rows: the chunk is capped by the partials' rows -- 4 chunks)
def test_teacher_runs_only_on_positions_with_a_loss_mask(backend, B, S, teacher_rows):
    """loss-row compaction, teacher side: with host-known row counts and a sparse loss mask the head GEMM + reduction run over the
    gathered positions with loss_mask != 0 only (two chunks here) and their stored logits / row scalars are scattered back.  Against the
    dense form of the same step: position mask everywhere, target ids where the loss mask is set, every metric bit for bit, gradients
    to summation order (the lm_head weight gradient contracts a compact stash)."""
    kw = dict(hidden_size=64, intermediate_size=96, num_attention_heads=2, num_key_value_heads=1, vocab_size=640,
              draft_vocab_size=256, head_dim=64, target_hidden_size=512, max_position_embeddings=512, rms_norm_eps=1e-5)
    T = 3
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(18)
    params = {k: (v.float() * 4).to(bf) if v.dim() > 1 else (1 + 0.1 * torch.randn(v.shape, generator=g)).to(bf)
              for k, v in O.init_params(oc, seed=17).items()}
    embed = (torch.randn(640, 64, generator=g) * 0.5).to(bf)
    head_w = (torch.randn(640, 512, generator=g) * 0.2).to(bf)
    t2d, d2t = O.make_vocab_mapping(640, 256, seed=15)
    batch = O.make_batch(oc, B, S, seed=19, dtype=bf, lengths=[S, S - 10, S - 29, S - 60])
    lm = batch["loss_mask"].clone()
    lm[:, :8] = 0                          # (a prompt prefix without loss)
    lm[1, 40:60] = 0

    def run(compact):
        model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
        sd = dict(params)
        sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = embed, t2d, d2t
        model.load_state_dict(sd)
        eagle = OnlineEagle3Model(model, length=T).train()
        eagle.engine.compact_loss_rows = compact
        eagle.engine.teacher_rows = teacher_rows                             # 2 chunks of the ~480 gathered positions / 4 of ~840
        strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(head_w.to(backend)))
        out = strat.forward_loss(TrainBatch(dict(input_ids=batch["input_ids"], attention_mask=batch["attention_mask"], loss_mask=lm,
                                                 hidden_state=batch["hidden_state"].to(backend), target=batch["target"].to(backend)),
                                            {"target_repr": "hidden_state"}))
        out.loss.backward()
        assert eagle.engine._soft[0] == "zt" or str(backend) != "cpu"
        mets = {k: torch.stack(v).float().cpu() for k, v in out.metrics.items() if isinstance(v, list)}
        return mets, eagle.engine.flat.grad.float().cpu().clone(), eagle.last_artifacts["target_token_ids"].cpu(), \
            eagle.last_artifacts["position_mask"].cpu(), eagle.engine._teacher_compacted

    met_c, grad_c, ids_c, pm_c, used_c = run(True)
    met_d, grad_d, ids_d, pm_d, used_d = run(False)
    if str(backend) == "cpu":
        assert used_c and not used_d
    on = lm.bool()
    assert torch.equal(pm_c, pm_d) and torch.equal(ids_c[on], ids_d[on])
    if used_c:       # (a GPU takes the reduced head GEMM -- and with it the compact teacher -- for chip-filling shapes only)
        assert int(ids_c[~on].abs().sum()) == 0
    for k in met_c:
        assert torch.equal(met_c[k], met_d[k]), k
    torch.testing.assert_close(grad_c, grad_d, rtol=2e-2, atol=4e-3 * float(grad_d.abs().max()))


def test_variable_length_batches_share_one_arena(backend, golden_dir):
    """the collator pads every batch to its own longest sample, so real data brings a new (B, S)
    almost every step.  All shapes run inside the storage reserved for the largest one -- HBM does not grow -- and a
    shape revisited after others gives bit-identical results (the constants it relies on are re-laid on every switch)."""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    eng = eagle.engine
    full = _batch(blob, backend)
    B, S = full.tensors["input_ids"].shape
    eng.reserve(B, S)

    def cut(L, nb=B):
        t = {k: v[:nb, :L].contiguous() for k, v in full.tensors.items()}
        t["attention_mask"] = t["attention_mask"].clone()
        return TrainBatch(t, {"target_repr": "hidden_state"})

    def run(batch):
        eng.micro_in_window = 0
        out = strat.forward_loss(batch)
        out.loss.backward()
        return torch.stack(out.metrics["plosses"]).clone(), eng.flat.grad.clone()

    p0, g0 = run(full)
    reserved = eng.arena_bytes()       # reserve() + the teacher-logit scratch chunk (sized at the first forward: needs Vt)
    for L in (S - 3, 9):
        run(cut(L) if L == 9 else cut(L, nb=1))
        assert eng.arena_bytes() == reserved, "a smaller batch shape grew the arena"
    p1, g1 = run(full)
    assert torch.equal(p0, p1) and torch.equal(g0, g1)
    assert eng.arena_bytes() == reserved


def test_mrope_ids_past_the_table_grow_it(backend, golden_dir):
    """three-axis ids far beyond max_position_embeddings + 20: the reference computes mrope angles analytically from the ids
    (llama3_eagle.py:389-427: no cache, no limit); the engine's plain table grows to the host-known span, and ids that are already on
    the device and do not fit are refused by the device flag instead of rotating by a clamped angle"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_rope_mrope_fp32.pt"), weights_only=False)
    blob["batch"] = dict(blob["batch"], position_ids=blob["batch"]["position_ids"] + 700)
    blob = _oracle_bf16(blob)
    cfg, model, eagle, strat = _build(blob, backend)
    eagle.train()
    rows0 = eagle.engine.cos.shape[0]
    out = strat.forward_loss(_batch(blob, backend))
    out.loss.backward()
    assert eagle.engine.cos.shape[0] > rows0 >= 128
    torch.testing.assert_close(torch.stack(out.metrics["plosses"]).float().cpu(), blob["plosses"], rtol=2e-2, atol=2e-2)
    named = dict(model.named_parameters())
    worst = {k: float((named[k].grad.float().cpu() - g.float()).abs().max()) / float(g.float().abs().max().clamp_min(1e-8))
             for k, g in blob["grads"].items()}
    assert max(worst.values()) <= 5e-2, worst
    neg = blob["batch"]["position_ids"].clone()
    neg[1, 0, 0] = -2
    b = _batch(blob, backend)
    b.tensors["position_ids"] = neg
    with pytest.raises(ValueError, match="mrope"):
        strat.forward_loss(b)
    if backend != "cpu":
        cfg, model, eagle, strat = _build(blob, backend)       # a fresh engine: table at max_position_embeddings + 20 rows
        eagle.train()
        b = _batch(blob, backend)
        b.tensors["position_ids"] = b.tensors["position_ids"].to(backend)
        with pytest.raises(RuntimeError, match="position_ids"):
            strat.forward_loss(b).loss.backward()


@pytest.mark.parametrize("rope_scaling", [None, dict(rope_type="dynamic", factor=2.0)])
def test_draft_backbone_method_grows_its_rope_table_too(backend, rope_scaling):
    """``backbone`` -- the forward-only draft method the reference's own ``OnlineEagle3Model`` calls on the plugin draft
    (modeling/draft/base.py:80-109) -- at a sequence longer than max_position_embeddings + 20, two TTT steps, against the pinned oracle's
    decoder layer with its ``RopeCache`` (= the reference's rotary-cache semantics, goldens ``eagle3_rope_grow*``)"""
    kw = dict(hidden_size=64, intermediate_size=128, num_attention_heads=2, num_key_value_heads=1, vocab_size=96, draft_vocab_size=32,
              head_dim=64, target_hidden_size=64, max_position_embeddings=16, rms_norm_eps=1e-5, rope_scaling=rope_scaling)
    oc = O.DraftConfig(**kw)
    bf = torch.bfloat16
    params = {k: v.to(bf) for k, v in O.init_params(oc, seed=3).items()}
    model = LlamaForCausalLMEagle3(DraftConfig(**kw), device=backend)
    sd = dict(params)
    g = torch.Generator().manual_seed(4)
    sd["embed_tokens.weight"] = (torch.randn(96, 64, generator=g) * 0.05).to(bf)
    sd["t2d"], sd["d2t"] = O.make_vocab_mapping(96, 32, seed=1)
    model.load_state_dict(sd)
    B, S = 2, 44
    emb = [(torch.randn(B, S, 64, generator=g) * 0.5).to(bf) for _ in range(2)]
    hid = (torch.randn(B, S, 64, generator=g) * 0.5).to(bf)
    am = torch.ones(B, S, dtype=torch.long)
    rope = O.RopeCache(oc, bf, "cpu")
    add_mask = O.additive_attention_mask(am.bool(), S, bf)
    pos = torch.arange(S).unsqueeze(0)
    cache_o, cache_h = [[], []], [[], []]
    ho, hh = hid, hid.to(backend)
    for k in range(2):
        ho = O.decoder_layer(params, oc, emb[k], ho, cache_o, add_mask, pos, rope)
        hh = model.backbone(emb[k].to(backend), hh, cache_h, am, None)
        torch.testing.assert_close(hh.float().cpu(), ho.float(), rtol=3e-2, atol=3e-2)
    assert model._rope[0].shape[0] >= S + 1 and rope.len == S + 1


def test_early_lm_head_weight_gradient_equals_the_deferred_one(backend, golden_dir):
    """engine.early_lm_head_wgrad (bench.py --dp-early-lm-head): the lm_head weight gradient taken BEFORE the data-gradient
    sweep -- its operands are complete after the forward -- is the same GEMM on the same operands: every gradient bit-identical, also over an
    accumulation window and with the loss-row compaction on; the bucket hook fires first for the lm_head range"""
    blob = torch.load(os.path.join(golden_dir, "eagle3_tiny_bf16.pt"), weights_only=False)
    res = {}
    for early in (False, True):
        cfg, model, eagle, strat = _build(blob, backend)
        eagle.train()
        eagle.engine.early_lm_head_wgrad = early
        seen = []
        eagle.engine.on_bucket_ready = lambda lo, hi: seen.append((lo, hi))
        for scale in (0.5, 0.25):           # two micro-steps of one window, different upstream gradients
            b = _batch(blob, backend)
            b.tensors["loss_mask"] = b.tensors["loss_mask"].clone()
            b.tensors["loss_mask"][:, :5] = 0
            out = strat.forward_loss(b)
            (out.loss * scale).backward()
        res[early] = (eagle.engine.flat.grad.clone().cpu(), list(seen))
        assert seen[0] == eagle.engine.bucket_bounds()[0] and len(seen) == 2 * len(eagle.engine.bucket_bounds())
    assert torch.equal(res[True][0], res[False][0])
    assert res[True][1] == res[False][1]
