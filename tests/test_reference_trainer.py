"""The HIP path driven by the REFERENCE's own code (build container only: needs /root/reference; skipped elsewhere).

* b1: ``AutoDraftModel.from_config`` constructs the registered HIP draft class; it IS an ``Eagle3DraftModel``; state
  dicts are key-for-key (order included) interchangeable with the reference's ``LlamaForCausalLMEagle3``; its four
  abstract methods agree with the reference class's own methods on the same weights.
* a18 / N1 / N5: the reference ``Trainer`` (``build_offline_runtime`` -> ``Trainer.fit()``: its data plane, collator,
  ``TrainerCore``, ``_reduce_eagle3_metrics``, checkpoint manager) runs the HIP strategy + backend + fused optimizer
  over feature files and logs, step for step, the values the PURE reference run logged on the same files
  (tests/golden/loss_curve_tiny.pt), mirroring tests/test_runtime/test_equiv_offline_eagle3.py:55-102.
* N1 / N2: ``specforge train``'s own path -- ``load_config`` -> ``resolve_run(cfg, registry=...)`` -> ``cli._train``
  (cli.py:113, application/composition.py:42-57) -- with the SAME run YAML and draft-config JSON as a reference run,
  then ``export_to_sglang`` (export/to_sglang.py:57-88) on the checkpoint it wrote and a strict reload into the
  reference's ``LlamaForCausalLMEagle3``.
The kernels run under the SIMT interpreter here (no GPU in the build container); the C-ABI calls are the product's.
"""
import json
import os

import pytest
import torch

from oracle import ref_harness as RH

pytestmark = pytest.mark.skipif(not RH.available(), reason="needs the reference checkout (/root/reference): build container only")


@pytest.fixture(scope="module")
def ref(emu_lib_path):
    RH.setup()
    RH.init_single_rank(29583)
    from specforge_amd import _lib
    from specforge_amd import reference_plugin as RP

    _lib._inject_library_for_tests(emu_lib_path)
    RP.draft_class()      # registers the HIP draft architecture (idempotent): the tests of this module may run in any order / worker
    yield RP
    RP.uninstall()
    _lib._inject_library_for_tests(None)


def _write_run_dir(work, blob, arch):
    from safetensors.torch import save_file

    dc = dict(blob["draft_config"], architectures=[arch])
    dj = os.path.join(work, "draft.json")
    json.dump(dc, open(dj, "w"))
    feat = os.path.join(work, "features")
    os.makedirs(feat)
    for i, raw in enumerate(blob["raws"]):
        torch.save(raw, os.path.join(feat, f"{i:04d}.ckpt"))
    td = os.path.join(work, "target")
    os.makedirs(td)
    H, V = dc["hidden_size"], dc["vocab_size"]
    json.dump({"architectures": ["LlamaForCausalLM"], "model_type": "llama", "hidden_size": H, "vocab_size": V,
               "num_hidden_layers": 1, "num_attention_heads": 4, "intermediate_size": 128}, open(os.path.join(td, "config.json"), "w"))
    w = {"lm_head.weight": blob["head_w"].float().contiguous(),
         "model.embed_tokens.weight": blob["init_state"]["embed_tokens.weight"].float().contiguous()}
    save_file(w, os.path.join(td, "model.safetensors"))
    json.dump({"metadata": {}, "weight_map": {k: "model.safetensors" for k in w}},
              open(os.path.join(td, "model.safetensors.index.json"), "w"))
    vp = os.path.join(work, "vm.pt")
    torch.save({"t2d": blob["t2d"], "d2t": blob["d2t"]}, vp)
    return dj, feat, td, vp


def test_draft_seam_is_the_reference_abc(ref, golden_dir, tmp_path):
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.draft.base import Eagle3DraftModel
    from specforge.modeling.draft.llama3_eagle import LlamaForCausalLMEagle3 as RefDraft

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    ref.draft_class()
    dj, feat, td, vp = _write_run_dir(str(tmp_path), blob, ref.DRAFT_ARCHITECTURE)
    hip = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), attention_backend="sdpa", torch_dtype=torch.bfloat16)
    assert isinstance(hip, Eagle3DraftModel) and type(hip).__name__ == ref.DRAFT_ARCHITECTURE
    rdj = os.path.join(str(tmp_path), "draft_ref.json")
    json.dump(blob["draft_config"], open(rdj, "w"))
    rm = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(rdj), attention_backend="sdpa", torch_dtype=torch.bfloat16)
    assert isinstance(rm, RefDraft)
    assert list(hip.state_dict()) == list(rm.state_dict())               # names AND order
    rm.load_state_dict(blob["init_state"], strict=True)
    hip.load_state_dict(rm.state_dict(), strict=True)                     # reference -> HIP
    rm.load_state_dict(hip.state_dict(), strict=True)                     # HIP -> reference
    hip.load_vocab_mapping(vp)
    hip.load_embedding(td, embedding_key="model.embed_tokens.weight")     # base.py:135 / :193
    assert hip.vocab_mapping_loaded and torch.equal(hip.t2d, blob["t2d"]) and torch.equal(hip.d2t, blob["d2t"])
    assert torch.equal(hip.embed_tokens.weight, blob["init_state"]["embed_tokens.weight"])
    hip.freeze_embedding()
    assert not hip.embed_tokens.weight.requires_grad
    # the four abstract methods vs the reference class's own (llama3_eagle.py:1702-1798), two TTT steps
    b = blob["batches"][0]
    B, S = b["input_ids"].shape
    close = lambda a, c: torch.testing.assert_close(a.float(), c.float(), rtol=2e-2, atol=2e-2)
    with torch.no_grad():
        assert torch.equal(hip.embed_input_ids(b["input_ids"]), rm.embed_input_ids(b["input_ids"]))
        h_h, h_r = hip.project_hidden_states(b["hidden_state"]), rm.project_hidden_states(b["hidden_state"])
        close(h_h, h_r)
        mask4 = rm.prepare_decoder_attention_mask(attention_mask=b["attention_mask"], hidden_states=h_r, batch_size=B,
                                                  seq_length=S, past_key_values_length=0)
        pos = torch.arange(S).unsqueeze(0)
        ch, cr = [[], []], [[], []]
        ids = b["input_ids"]
        for step in range(2):
            e = rm.embed_input_ids(ids)
            h_h = hip.backbone(input_embeds=e, hidden_states=h_h, cache_hidden=ch, attention_mask=mask4, position_ids=pos)
            h_r = rm.backbone(input_embeds=e, hidden_states=h_r, cache_hidden=cr, attention_mask=mask4, position_ids=pos,
                              use_cache=True)
            valid = b["attention_mask"].bool()
            close(h_h[valid], h_r[valid])
            close(hip.compute_logits(h_h)[valid], rm.compute_logits(h_r)[valid])
            ids = torch.cat((ids[:, 1:], torch.zeros_like(ids[:, -1:])), dim=1)
            h_h = h_r.clone()          # both continue from the same hidden state: errors do not compound over steps
        assert len(ch[0]) == 2 and len(ch[1]) == 2


def test_reference_trainer_fit_reproduces_the_reference_run(ref, golden_dir, tmp_path):
    """reference Trainer + data plane + HIP strategy/backend/optimizer == the pure reference run, step for step"""
    from specforge.launch import build_offline_runtime
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.target.target_head import TargetHead
    from specforge_amd.eagle3 import OnlineEagle3Model
    from specforge_amd.training import BF16Optimizer

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    c = blob["cfg"]
    dj, feat, td, vp = _write_run_dir(str(tmp_path), blob, ref.DRAFT_ARCHITECTURE)
    ref.install()
    draft = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), attention_backend="sdpa", torch_dtype=torch.bfloat16)
    draft.load_state_dict(blob["init_state"], strict=True)
    draft.freeze_embedding()
    head = TargetHead.from_pretrained(td, lm_head_key="lm_head.weight")
    model = OnlineEagle3Model(draft_model=draft, length=c["ttt"], attention_backend="sdpa")
    logged = []
    nsteps = 4
    trainer = build_offline_runtime(
        algorithm=ref.registry().resolve(ref.ALGORITHM_NAME), hidden_states_path=feat, draft_model=model, target_head=head,
        optimizer_factory=lambda m: BF16Optimizer(m, lr=c["lr"], max_grad_norm=c["max_grad_norm"], warmup_ratio=c["warmup_ratio"],
                                                  total_steps=c["steps"]),
        run_id="curve-hip", output_dir=os.path.join(str(tmp_path), "out"), ttt_length=c["ttt"], max_len=c["max_len"],
        batch_size=c["batch_size"], max_steps=nsteps, total_steps=c["steps"], num_epochs=2, seed=c["seed"],
        logger=lambda m, s: logged.append((s, m)), log_interval=1)
    assert type(trainer.backend).__name__ == "_HipBackendForTrainer"
    # f1: the trainer's batches come from the HIP ingest (reference_plugin.feature_loader_class), not the reference's loader
    assert type(trainer._loader) is ref.feature_loader_class() and trainer._loader._ingest is None
    assert trainer.fit() == nsteps
    assert trainer._loader._ingest is not None and trainer._loader._stager is None       # ... through the direct reader
    assert [s for s, _ in logged] == list(range(1, nsteps + 1))
    for (s, got), want in zip(logged, blob["logged"]):
        assert {"loss", "acc", "ploss_0", "acc_0", "acceptance_rate_0", "grad_norm", "lr"} <= set(got)   # _reduce_eagle3_metrics ran
        for k, v in want.items():
            tol = 1e-9 if k == "lr" else 2e-2 * max(1.0, abs(v))
            assert abs(got[k] - v) <= tol, (s, k, got[k], v)
    # the checkpoint the reference's manager wrote: reference key set, no embedding, strategy name of the algorithm
    state = torch.load(os.path.join(str(tmp_path), "out", "curve-hip-latest", "training_state.pt"), weights_only=False)
    assert state["strategy"] == ref.ALGORITHM_NAME and state["global_step"] == nsteps
    want_keys = [k for k in blob["init_state"] if "embed" not in k]
    assert sorted(state["draft_state_dict"]) == sorted(want_keys)
    assert "replicated_optimizer_state" in state and "fp32_params" in state["replicated_optimizer_state"]


def test_plugin_loader_yields_the_reference_loaders_batches(ref, golden_dir, tmp_path):
    """f1: ``HipFeatureDataLoader`` (what ``install()`` binds in place of ``FeatureDataLoader``, trainer.py:145) against the
    reference's own loader over the same refs, normaliser and collator -- batch for batch, tensor for tensor, ids and
    metadata included; ``seek`` / ``set_epoch`` / ``drop_last=False`` (the eval loader keeps the partial batch);
    a ``.ckpt.gz`` dataset takes the reference's materialisation (staged), with the same result."""
    import gzip
    import shutil

    from specforge.launch import _offline_io, _shard_offline_refs
    from specforge.runtime.data_plane import FeatureDataLoader, LocalFeatureStore

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    c = blob["cfg"]
    dj, feat, td, vp = _write_run_dir(str(tmp_path), blob, ref.DRAFT_ARCHITECTURE)
    alg = ref.registry().resolve(ref.ALGORITHM_NAME)
    collate, transform = _offline_io(alg, "text", c["max_len"], ttt_length=c["ttt"], use_usp_preprocess=False)
    Hip = ref.feature_loader_class()

    def batches(cls, refs, **kw):
        kw = dict(dict(batch_size=2, collate_fn=collate, per_sample_transform=transform, strategy=alg.name), **kw)
        return cls(LocalFeatureStore("t"), refs=refs, **kw)

    def same(a, b):
        from specforge_amd.eagle3 import loss_mask_suffix_counts

        # (one extra metadata key: the per-TTT-step loss-row counts, computed where the mask is in host memory -- the engine's
        #  loss-row compaction reads them; the reference ignores unknown keys)
        extra = {"loss_mask_suffix_counts": loss_mask_suffix_counts(b.tensors["loss_mask"])}
        assert a.sample_ids == b.sample_ids and a.strategy == b.strategy and a.metadata == dict(b.metadata, **extra)
        assert set(a.tensors) == set(b.tensors)
        for k, v in b.tensors.items():
            assert a.tensors[k].shape == v.shape and a.tensors[k].dtype == v.dtype and torch.equal(a.tensors[k].cpu(), v), k

    source = alg.providers.offline_for("text").build_reader(feat, run_id="t", ttt_length=c["ttt"], max_len=c["max_len"]).read()
    for epoch in (0, 1):
        refs = _shard_offline_refs(source, use_usp_preprocess=False, seed=3, epoch=epoch, dp_rank=0, dp_size=1)
        want = list(batches(FeatureDataLoader, refs))
        hip = batches(Hip, refs)
        got = []
        for b in hip:                       # a batch is valid until the next is requested: compare inside the loop
            same(b, want[len(got)])
            got.append(b.sample_ids)
        assert len(got) == len(want) > 1 and hip._ingest is not None
    # seek (resume) + drop_last=False (eval loader: an odd number of refs leaves a partial batch)
    odd = refs[:5]
    want = list(batches(FeatureDataLoader, odd, drop_last=False))
    hip = batches(Hip, odd, drop_last=False)
    n = 0
    for b in hip:
        same(b, want[n])
        n += 1
    assert n == 3 and want[-1].tensors["input_ids"].shape[0] == 1
    hip.seek(2)
    rest = [b.sample_ids for b in hip]
    assert rest == [want[2].sample_ids]
    with pytest.raises(ValueError):
        hip.seek(4)
    # .ckpt.gz: not the direct reader's format -> the reference's own _make_batch, same batches
    gz = os.path.join(str(tmp_path), "gz")
    os.makedirs(gz)
    for name in sorted(os.listdir(feat)):
        with open(os.path.join(feat, name), "rb") as fi, gzip.open(os.path.join(gz, name + ".gz"), "wb") as fo:
            shutil.copyfileobj(fi, fo)
    grefs = alg.providers.offline_for("text").build_reader(gz, run_id="t", ttt_length=c["ttt"], max_len=c["max_len"]).read()
    want = list(batches(FeatureDataLoader, grefs))
    hip = batches(Hip, grefs)
    n = 0
    for b in hip:
        same(b, want[n])
        n += 1
    assert n == len(want) and hip._ingest is None


def test_reference_evaluator_runs_over_the_hip_strategy(ref, golden_dir, tmp_path):
    """the reference's own eval pass (TrainerController.evaluate_configured -> eval/evaluator.py Evaluator.run, controller.py:779-817)
    over the HIP strategy in eval mode == the same pass over the pure reference model on the same feature files: every eval/* metric"""
    from specforge.algorithms.builtin import builtin_algorithm_registry
    from specforge.algorithms.eagle3.model import OnlineEagle3Model as RefOnline
    from specforge.launch import build_offline_runtime
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.target.target_head import TargetHead
    from specforge_amd.eagle3 import OnlineEagle3Model
    from specforge_amd.training import BF16Optimizer

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    c = blob["cfg"]
    dj, feat, td, vp = _write_run_dir(str(tmp_path), blob, ref.DRAFT_ARCHITECTURE)
    dj_ref = os.path.join(str(tmp_path), "draft_ref.json")
    cfgj = json.load(open(dj))
    cfgj["architectures"] = ["LlamaForCausalLMEagle3"]
    json.dump(cfgj, open(dj_ref, "w"))
    common = dict(hidden_states_path=feat, eval_hidden_states_path=feat, eval_interval=1000, ttt_length=c["ttt"], max_len=c["max_len"],
                  batch_size=c["batch_size"], max_steps=1, total_steps=c["steps"], num_epochs=1, seed=c["seed"], log_interval=1,
                  logger=lambda m, s: None)
    res = {}
    for kind in ("ref", "hip"):
        (ref.uninstall if kind == "ref" else ref.install)()     # the pure run must see the reference's own backend / optimizer
        ref.draft_class()                                        # (registers the HIP draft architecture; idempotent)
        draft = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj_ref if kind == "ref" else dj), attention_backend="sdpa",
                                           torch_dtype=torch.bfloat16)
        draft.load_state_dict(blob["init_state"], strict=True)
        draft.freeze_embedding()
        head = TargetHead.from_pretrained(td, lm_head_key="lm_head.weight")
        if kind == "ref":
            model = RefOnline(draft_model=draft, length=c["ttt"], attention_backend="sdpa")
            alg = builtin_algorithm_registry().resolve("eagle3")
            from specforge.optimizer import BF16Optimizer as RefOpt      # (bound after uninstall())
            opt = lambda m: RefOpt(m, lr=c["lr"], max_grad_norm=c["max_grad_norm"], warmup_ratio=c["warmup_ratio"], total_steps=c["steps"])
        else:
            model = OnlineEagle3Model(draft_model=draft, length=c["ttt"], attention_backend="sdpa")
            alg = ref.registry().resolve(ref.ALGORITHM_NAME)
            opt = lambda m: BF16Optimizer(m, lr=c["lr"], max_grad_norm=c["max_grad_norm"], warmup_ratio=c["warmup_ratio"], total_steps=c["steps"])
        trainer = build_offline_runtime(algorithm=alg, draft_model=model, target_head=head, optimizer_factory=opt, run_id=f"eval-{kind}",
                                        output_dir=os.path.join(str(tmp_path), f"out-{kind}"), **common)
        try:
            res[kind] = trainer._controller.evaluate_configured()
        finally:
            ref.uninstall()
        assert model.training                       # evaluate() restores the training mode it found
    want, got = res["ref"], res["hip"]
    assert want and set(got) == set(want) and any(k.startswith("eval/") for k in want)
    flat = lambda x: [float(y) for y in x] if isinstance(x, (list, tuple)) else [float(x)]
    for k, v in want.items():
        vs, gs = flat(v), flat(got[k])
        assert len(vs) == len(gs), k
        for a, g in zip(vs, gs):
            assert abs(g - a) <= 2e-2 * max(1.0, abs(a)), (k, g, a)


def test_cli_train_path_and_sglang_export_round_trip(ref, golden_dir, tmp_path):
    """the SAME run YAML / draft JSON a reference run would use (strategy ``eagle3``, ``LlamaForCausalLMEagle3``)"""
    import yaml

    from specforge import cli
    from specforge.application.composition import resolve_run
    from specforge.config import load_config
    from specforge.export.to_sglang import export_to_sglang
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    work = str(tmp_path)
    dj, feat, td, vp = _write_run_dir(work, blob, "LlamaForCausalLMEagle3")
    run = dict(model=dict(target_model_path=td, draft_model_config=dj, embedding_key="model.embed_tokens.weight",
                          vocab_mapping_path=vp, torch_dtype="bfloat16"),
               data=dict(hidden_states_path=feat, max_length=24),
               training=dict(strategy="eagle3", num_epochs=1, batch_size=2, learning_rate=1e-3, max_grad_norm=0.5, ttt_length=3,
                             attention_backend="sdpa", save_interval=0, log_interval=1, dist_timeout=5, seed=0, max_steps=2),
               run_id="hipcli", output_dir=os.path.join(work, "out"),
               deployment=dict(mode="local_colocated", trainer=dict(nnodes=1, nproc_per_node=1)))
    yp = os.path.join(work, "run.yaml")
    yaml.safe_dump(run, open(yp, "w"))
    ref.install(override=True)
    try:
        resolved = resolve_run(load_config(yp), registry=ref.registry(override=True))
        assert resolved.algorithm.name == "eagle3"
        import torch.distributed as dist

        if dist.is_initialized():          # cli._train owns init/destroy of the process group
            dist.destroy_process_group()
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            os.environ.pop(k, None)
        assert cli._train(resolved) == 2
    finally:
        ref.uninstall()
        RH.init_single_rank(29585)
    ckpt = os.path.join(work, "out")
    state = torch.load(os.path.join(ckpt, "hipcli-latest", "training_state.pt"), weights_only=False)
    assert state["strategy"] == "eagle3"
    out = export_to_sglang(ckpt, dj, os.path.join(work, "sglang"))       # materialises the REFERENCE class (plugin uninstalled)
    from safetensors.torch import load_file

    sd = load_file(os.path.join(out, "model.safetensors"))
    assert {"fc.weight", "norm.weight", "lm_head.weight", "t2d", "d2t"} <= set(sd) and not any("embed" in k for k in sd)
    for k, v in state["draft_state_dict"].items():
        assert torch.equal(sd[k], v), k
    rm = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), torch_dtype=torch.bfloat16)
    missing, unexpected = rm.load_state_dict(sd, strict=False)
    assert not unexpected and all("embed" in k for k in missing)
    assert json.load(open(os.path.join(out, "config.json")))["architectures"] == ["LlamaForCausalLMEagle3"]


def test_reference_tiny_fixture_head_dim_16_trains_unmodified(ref, tmp_path):
    """The reference's OWN test fixture (tests/test_runtime/_fixtures.py: TINY_DRAFT_CONFIG = hidden 64, 4 / 2 heads, i.e.
    head_dim 16; its target-head, vocab-map and feature-file writers) through ``build_offline_runtime -> Trainer.fit()``
    twice: the pure reference, then the same files and the same draft JSON with the HIP path installed.  head_dim 16 runs on
    the zero-padded-heads path of the engine (engine.py), ttt_length 9 on more than 8 diagonal branches."""
    import importlib.util

    ref.uninstall()        # (an earlier test of this process may have left the HIP backend / optimizer bound into the reference's modules:
    #                         the names imported below, and the first run, must be the PURE reference)
    from specforge.algorithms.builtin import builtin_algorithm_registry
    from specforge.algorithms.eagle3.model import OnlineEagle3Model as RefOnline
    from specforge.launch import build_offline_runtime
    from specforge.modeling.auto import AutoDraftModel, AutoDraftModelConfig
    from specforge.modeling.target.target_head import TargetHead
    from specforge.optimizer import BF16Optimizer as RefOpt
    from specforge_amd.eagle3 import OnlineEagle3Model
    from specforge_amd.training import BF16Optimizer

    spec = importlib.util.spec_from_file_location("ref_fixtures", os.path.join(RH.REFERENCE_ROOT, "tests", "test_runtime", "_fixtures.py"))
    fx = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fx)
    assert fx.TINY_DRAFT_CONFIG["hidden_size"] // fx.TINY_DRAFT_CONFIG["num_attention_heads"] == 16
    os.environ["FSDP_SHARDING"] = "NO_SHARD"      # the pure reference run: DDP over gloo (an earlier test's cli._train may have reset it)
    work = str(tmp_path)
    torch.manual_seed(0)
    dj = fx.write_draft_config(os.path.join(work, "draft.json")) if hasattr(fx, "write_draft_config") else None
    if dj is None:
        dj = os.path.join(work, "draft.json")
        json.dump(fx.TINY_DRAFT_CONFIG, open(dj, "w"))
    td = fx.write_target_head_dir(os.path.join(work, "target"))
    vp = fx.write_vocab_mapping(os.path.join(work, "vm.pt"))
    feat = fx.write_offline_files(os.path.join(work, "features"), n=8, seq=21)
    ttt, nsteps, kw = 9, 3, dict(lr=2e-3, max_grad_norm=0.5, warmup_ratio=0.0, total_steps=10)

    def run(hip: bool):
        torch.manual_seed(1)
        draft = AutoDraftModel.from_config(AutoDraftModelConfig.from_file(dj), attention_backend="sdpa", torch_dtype=torch.bfloat16)
        torch.manual_seed(2)
        with torch.no_grad():
            for p in draft.parameters():
                p.copy_(torch.randn(p.shape) * (0.08 if p.dim() > 1 else 0.1) + (1.0 if p.dim() == 1 else 0.0))
        draft.load_vocab_mapping(vp)
        draft.freeze_embedding()
        head = TargetHead.from_pretrained(td, lm_head_key="lm_head.weight")
        model = (OnlineEagle3Model if hip else RefOnline)(draft_model=draft, length=ttt, attention_backend="sdpa")
        logged = []
        trainer = build_offline_runtime(
            algorithm=(ref.registry(override=True) if hip else builtin_algorithm_registry()).resolve("eagle3"),
            hidden_states_path=feat, draft_model=model, target_head=head,
            optimizer_factory=lambda m: (BF16Optimizer if hip else RefOpt)(m, **kw),
            run_id="tiny-hip" if hip else "tiny-ref", output_dir=os.path.join(work, "out_hip" if hip else "out_ref"),
            ttt_length=ttt, max_len=32, batch_size=2, max_steps=nsteps, num_epochs=2, seed=0,
            logger=lambda m, s: logged.append({k: v for k, v in m.items() if not k.startswith("perf/")}), log_interval=1)
        assert trainer.fit() == nsteps
        return logged, type(draft).__name__, type(trainer.backend).__name__

    want, cls_ref, be_ref = run(False)
    ref.install(override=True)
    try:
        got, cls_hip, be_hip = run(True)
    finally:
        ref.uninstall()
    assert cls_ref == cls_hip == "LlamaForCausalLMEagle3" and be_ref == "FSDPTrainingBackend" and be_hip == "_HipBackendForTrainer"
    assert len(got) == len(want) == nsteps
    for s, (g, w) in enumerate(zip(got, want)):
        assert {f"ploss_{ttt - 1}", f"acc_{ttt - 1}", "grad_norm", "lr"} <= set(g)
        for k, v in w.items():
            tol = 1e-9 if k == "lr" else 3e-2 * max(1.0, abs(v))
            assert abs(g[k] - v) <= tol, (s, k, g[k], v)


def test_reference_optimizer_with_weight_decay_skips_the_gradless_norm(ref, golden_dir):
    """norm_output=False: the final `norm` is never applied, the reference leaves its .grad None and its BF16Optimizer skips
    the parameter (optimizer.py:139-142) -- also under weight decay.  The HIP path driven by the REFERENCE optimizer
    (HipDPTrainingBackend.set_optimizer) must do the same: no decay on norm.weight, no Adam state for it."""
    from specforge.optimizer import BF16Optimizer as RefOpt
    from specforge_amd.eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, TargetHead, TrainBatch
    from specforge_amd.model import DraftConfig, LlamaForCausalLMEagle3
    from specforge_amd.training import HipDPTrainingBackend

    blob = torch.load(os.path.join(golden_dir, "eagle3_nonorm_fp32.pt"), weights_only=False)
    c = blob["cfg"]
    cfg = DraftConfig(hidden_size=c["H"], intermediate_size=c["I"], num_attention_heads=c["nh"], num_key_value_heads=c["nkv"],
                      vocab_size=c["Vt"], draft_vocab_size=c["Vd"], head_dim=c["hd"], target_hidden_size=c["Ht"],
                      max_position_embeddings=c["max_pos"], rms_norm_eps=c["eps"], fc_norm=c["fc_norm"],
                      rope_scaling=c["rope_scaling"], norm_output=False)
    model = LlamaForCausalLMEagle3(cfg)
    sd = {k: v.to(torch.bfloat16) for k, v in blob["params"].items()}
    sd["embed_tokens.weight"], sd["t2d"], sd["d2t"] = blob["embed"].to(torch.bfloat16), blob["t2d"], blob["d2t"]
    model.load_state_dict(sd)
    eagle = OnlineEagle3Model(model, length=c["ttt"]).train()
    strat = Eagle3TrainStrategy(eagle, target_head=TargetHead(blob["head_w"].to(torch.bfloat16)))
    backend = HipDPTrainingBackend()
    backend.prepare_model(eagle)
    eagle.engine                                   # adopt the parameters before the optimizer clones its masters
    backend.set_optimizer(RefOpt(model, lr=1e-2, weight_decay=0.5, max_grad_norm=0.5, total_steps=10, warmup_ratio=0.0))
    b = blob["batch"]
    batch = TrainBatch(dict(input_ids=b["input_ids"], attention_mask=b["attention_mask"], loss_mask=b["loss_mask"],
                            hidden_state=b["hidden_state"].to(torch.bfloat16), target=b["target"].to(torch.bfloat16)),
                       {"target_repr": "hidden_state"})
    named = dict(model.named_parameters())
    before = {k: v.detach().clone() for k, v in named.items()}
    for _ in range(2):
        out = strat.forward_loss(batch)
        backend.backward(out.loss, is_boundary=True)
        assert named["norm.weight"].grad is None
        backend.step()
    assert torch.equal(named["norm.weight"], before["norm.weight"])          # untouched: no decay, no update
    moved = [k for k, v in named.items() if v.requires_grad and k != "norm.weight" and not torch.equal(v, before[k])]
    assert len(moved) == sum(1 for k, v in named.items() if v.requires_grad and k != "norm.weight")
    st = backend.optimizer.state_dict()["optimizer_state_dict"]["state"]
    trainable = [k for k, v in named.items() if v.requires_grad]
    assert trainable.index("norm.weight") not in st                            # no Adam state for the grad-less parameter


def test_drop_in_cli_entry_runs_specforge_train_on_the_hip_path(ref, golden_dir, tmp_path, capsys):
    """``python -m specforge_amd.reference_plugin train -c run.yaml`` = ``specforge train`` (cli.py:167-268) with the plugin installed in
    every process: the single-process plan trains through ``cli._train`` (the reference's own application / Trainer / checkpoint
    code, HIP model + step + backend + optimizer + loader); a 2-GPU topology renders a ``torch.distributed.run`` command whose
    workers are this module again (so each rank installs the plugin), not ``specforge.cli``."""
    import yaml

    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    work = str(tmp_path)
    dj, feat, td, vp = _write_run_dir(work, blob, "LlamaForCausalLMEagle3")

    def run_yaml(nproc, name):
        run = dict(model=dict(target_model_path=td, draft_model_config=dj, embedding_key="model.embed_tokens.weight",
                              vocab_mapping_path=vp, torch_dtype="bfloat16"),
                   data=dict(hidden_states_path=feat, max_length=24),
                   training=dict(strategy="eagle3", num_epochs=1, batch_size=2, learning_rate=1e-3, max_grad_norm=0.5, ttt_length=3,
                                 attention_backend="sdpa", save_interval=0, log_interval=1, dist_timeout=5, seed=0, max_steps=2),
                   run_id=name, output_dir=os.path.join(work, "out-" + name),
                   deployment=dict(mode="local_colocated", trainer=dict(nnodes=1, nproc_per_node=nproc)))
        yp = os.path.join(work, name + ".yaml")
        yaml.safe_dump(run, open(yp, "w"))
        return yp

    import torch.distributed as dist

    if dist.is_initialized():          # cli._train owns init / destroy of the process group
        dist.destroy_process_group()
    saved = {k: os.environ.pop(k, None) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    try:
        assert ref.main(["train", "-c", run_yaml(1, "hipmain")]) == 0
        state = torch.load(os.path.join(work, "out-hipmain", "hipmain-latest", "training_state.pt"), weights_only=False)
        assert state["strategy"] == "eagle3" and state["global_step"] == 2
        capsys.readouterr()
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):     # (the one-process run bootstrapped them, cli.py:81-106)
            os.environ.pop(k, None)
        assert ref.main(["train", "-c", run_yaml(2, "hipplan"), "--plan"]) == 0
        plan = json.loads(capsys.readouterr().out)
        assert plan["kind"] == "command" and len(plan["commands"]) == 1
        argv = plan["commands"][0]["argv"]
        i = argv.index("torch.distributed.run")
        assert argv[i + 1] == "--no-python" and argv[argv.index("--nproc_per_node") + 1] == "2"
        j = argv.index("specforge_amd.reference_plugin")
        assert argv[j - 1] == "-m" and argv[j + 1:j + 4] == ["train", "--config", os.path.join(work, "hipplan.yaml")]
        assert "specforge.cli" not in argv
    finally:
        ref.uninstall()
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            if saved[k] is None:
                os.environ.pop(k, None)
        RH.init_single_rank(29587)


def _world2_run_dir(golden_dir, work):
    """24 feature files of EQUAL length 16 (so that a micro-batch's row count -- the denominator of every ``ploss`` mean -- does not
    depend on how samples are grouped: the world-2 run and the single-process accumulation then log the same numbers)"""
    blob = torch.load(os.path.join(golden_dir, "loss_curve_tiny.pt"), weights_only=False)
    raws = [{k: (v[:16] if v.dim() == 1 else v[:, :16]).clone() for k, v in r.items()} for r in blob["raws"] if r["input_ids"].shape[0] >= 16][:24]
    assert len(raws) == 24
    return _write_run_dir(work, dict(blob, raws=raws), "LlamaForCausalLMEagle3")


def _launch_train(work, paths, name, *, nproc, accumulation, max_steps, batch_size=2, resume_from=None, timeout=600):
    """``python tests/_plugin_worker.py train -c <name>.yaml`` as a subprocess -> (logged {step: metrics}, probes by rank, checkpoint state)"""
    import ast
    import re
    import subprocess
    import sys

    import yaml

    dj, feat, td, vp = paths
    run = dict(model=dict(target_model_path=td, draft_model_config=dj, embedding_key="model.embed_tokens.weight", vocab_mapping_path=vp,
                          torch_dtype="bfloat16"),
               data=dict(hidden_states_path=feat, max_length=24),
               training=dict(strategy="eagle3", num_epochs=1, batch_size=batch_size, learning_rate=1e-3, max_grad_norm=0.5, ttt_length=3,
                             attention_backend="sdpa", save_interval=0, log_interval=1, dist_timeout=60, seed=0, max_steps=max_steps,
                             total_steps=3, accumulation_steps=accumulation),
               run_id=name, output_dir=os.path.join(work, "out-" + name),
               deployment=dict(mode="local_colocated", trainer=dict(nnodes=1, nproc_per_node=nproc)))
    if resume_from:
        run["training"]["resume_from"] = resume_from
    yp = os.path.join(work, name + ".yaml")
    yaml.safe_dump(run, open(yp, "w"))
    probe = os.path.join(work, "probe-" + name)
    os.makedirs(probe)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE",
                                                            "GROUP_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(SF_TEST_PROBE_DIR=probe, OMP_NUM_THREADS="2")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_plugin_worker.py"), "train", "-c", yp], env=env, cwd=root,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-4000:]
    logged = {}
    for m in re.finditer(r"^step (\d+): (\{.*\})\s*$", r.stdout, re.M):
        logged.setdefault(int(m.group(1)), ast.literal_eval(m.group(2)))
    probes = {}
    for f in sorted(os.listdir(probe)):
        rec = json.load(open(os.path.join(probe, f)))
        probes[rec["rank"]] = rec
    state = torch.load(os.path.join(work, "out-" + name, name + "-latest", "training_state.pt"), weights_only=False)
    return logged, probes, state, r.stdout


def test_specforge_train_at_world_size_2_is_really_launched(golden_dir, tmp_path):
    """``python -m specforge_amd.reference_plugin train -c run.yaml`` with ``deployment.trainer.nproc_per_node: 2`` EXECUTED
    (gloo, SIMT interpreter; round 5 stopped at ``--plan``): the reference's ``build_launch_plan -> run_commands`` spawns
    ``torch.distributed.run --standalone --nproc_per_node 2``, each worker installs the plugin and runs the reference's ``cli._train``
    -- its ``_shard_offline_refs`` (launch.py:174-239), ``TrainerCore`` with ``accumulation_steps: 2``, ``_reduce_eagle3_metrics``
    all-reduce (controller.py:257-282), rank-0 checkpoint (controller.py:839-886) -- over ``HipDPTrainingBackend``.  Mirrors
    /root/reference/tests/test_runtime/test_no_sync_equiv.py:132-172 (replicas equal, exactly ``acc - 1`` no_sync backwards per window)
    plus: logged metrics == the single-process run over the same 2 x refs per micro-step; a resumed run continues bit-identically."""
    work = str(tmp_path)
    paths = _world2_run_dir(golden_dir, work)
    steps, acc = 3, 2
    logged2, probes2, state2, out2 = _launch_train(work, paths, "w2", nproc=2, accumulation=acc, max_steps=steps)
    # --- both ranks trained, replicas bit-identical, no_sync protocol
    assert sorted(probes2) == [0, 1] and all(p["world"] == 2 for p in probes2.values()), probes2
    assert probes2[0]["weights_sha256"] == probes2[1]["weights_sha256"]
    for p in probes2.values():
        assert p["boundary_backwards"] == steps and p["skipped_backwards"] == steps * (acc - 1) == p["no_sync_backwards"], p
        assert p["bucket_allreduces"] > 0 and p["bucket_allreduces"] % steps == 0, p       # only boundary backwards reduce: same count per window
    assert sorted(logged2) == [1, 2, 3], out2[-3000:]
    # --- the rank-0 checkpoint: DDP convention, optimizer state stored once (controller.py:867-871)
    assert state2["global_step"] == steps and state2["strategy"] == "eagle3"
    assert "replicated_optimizer_state" in state2 and "fp32_params" in state2["replicated_optimizer_state"]
    assert os.path.exists(os.path.join(work, "out-w2", "w2-latest", "training_state_rank1.pt"))
    # --- == ONE process taking the same 2 x refs per micro-step (world 1, batch 4, same accumulation: micro-step m of window w is samples
    #     perm[8w + 4m .. 8w + 4m + 3] of the epoch permutation on both sides, launch.py:219-239 -- rank r holds perm[r::2]; equal-length
    #     samples, so every mean has the same denominator).  What a step LOGS is its boundary micro-step's counts summed over ranks
    #     (controller.py:200-304), i.e. exactly the 4-sample micro-batch of the single process.
    logged1, probes1, state1, out1 = _launch_train(work, paths, "w1", nproc=1, accumulation=acc, batch_size=4, max_steps=steps)
    assert probes1[0]["world"] == 1 and probes1[0]["bucket_allreduces"] == 0
    for s in (1, 2, 3):
        a, b = logged2[s], logged1[s]
        keys = [k for k in b if k == "loss" or k == "acc" or k.startswith(("ploss_", "acc_", "acceptance_rate_")) or k in ("grad_norm", "lr")]
        assert {"loss", "ploss_0", "ploss_2", "acc_0", "grad_norm", "lr"} <= set(keys)
        for k in keys:
            tol = 1e-9 if k == "lr" else (2e-3 if s == 1 else 2e-2) * max(1.0, abs(b[k]))      # step 1: same weights on both sides
            assert abs(a[k] - b[k]) <= tol, (s, k, a[k], b[k])
    # (weights: 3 Adam steps of lr 1e-3 move an entry by <= 1e-3 each whatever the gradient's size, so an entry whose tiny gradient rounds
    #  to the other sign differs by up to 2 * lr per step; the bulk agrees far below one step)
    for k, v in state1["draft_state_dict"].items():
        d = (state2["draft_state_dict"][k].float() - v.float()).abs()
        assert float(d.max()) <= 2 * 1e-3 * steps + 1e-4 and float(d.mean()) <= 2e-4, (k, float(d.max()), float(d.mean()))
    # --- resume: 2 steps, checkpoint, a NEW world-2 launch resumes and takes step 3 -> the weights of the uninterrupted run, bit for bit
    _, _, state_a, _ = _launch_train(work, paths, "w2a", nproc=2, accumulation=acc, max_steps=2)
    assert state_a["global_step"] == 2
    logged_b, probes_b, state_b, out_b = _launch_train(work, paths, "w2b", nproc=2, accumulation=acc, max_steps=steps,
                                                       resume_from=os.path.join(work, "out-w2a", "w2a-latest"))
    assert state_b["global_step"] == steps and sorted(logged_b) == [3], out_b[-3000:]
    assert probes_b[0]["weights_sha256"] == probes_b[1]["weights_sha256"] == probes2[0]["weights_sha256"]
    for k, v in state2["draft_state_dict"].items():
        assert torch.equal(state_b["draft_state_dict"][k], v), k
    for k in ("loss", "ploss_0", "acc_0", "grad_norm", "lr"):
        assert logged_b[3][k] == logged2[3][k], (k, logged_b[3][k], logged2[3][k])
