"""Binding of the HIP path into the reference's OWN extension seams (sgl-project/SpecForge).

Importable only where the ``specforge`` package is (this module is the reference-side stub of INTEGRATION.md made
executable; nothing else in ``specforge_amd`` imports it).  It provides, each at the seam SURVEY.md section 8b names:

* b1  ``HipLlamaForCausalLMEagle3``: a ``@register_draft`` subclass of the reference's ``Eagle3DraftModel``
  (modeling/draft/base.py:38-206, registry modeling/draft/registry.py:26-50) with the reference's parameter names; its
  four abstract methods, ``load_embedding`` and ``load_vocab_mapping`` are ``specforge_amd.model.Eagle3DraftMethods``
  (C-ABI calls).  ``AutoDraftModel.from_config(cfg, attention_backend=..., torch_dtype=...)``
  (algorithms/model_providers.py:91-112) constructs it from a draft-config JSON whose ``architectures[0]`` names it.
* b2  an ``AlgorithmRegistration`` ``eagle3_hip`` = the reference's own EAGLE3 spec and data providers (reader,
  normaliser, collator, feature contracts: algorithms/eagle3/providers.py) with the model / step factories swapped for
  the HIP ``OnlineEagle3Model`` and ``Eagle3TrainStrategy``; added to the built-in catalogue through
  ``AlgorithmRegistry.with_registration`` (algorithms/registry.py:86-90) and handed to ``resolve_run(cfg, registry=...)``
  (application/composition.py:42-57).
* b3  ``install_backend()``: ``Trainer.__init__`` hard-codes ``FSDPTrainingBackend(parallel, optimizer_factory=...)``
  (training/trainer.py:421) -- the one line a maintainer would make injectable; until then the name is rebound to
  ``HipDPTrainingBackend`` in that module.  The configured optimizer factory (training/assembly.py:246-275) is rebound
  to this package's fused ``BF16Optimizer`` with the same arguments.
* f1  ``feature_loader_class()``: ``Trainer.__init__`` also hard-codes ``FeatureDataLoader(store, refs=..., ...)``
  (training/trainer.py:145-155).  ``install()`` rebinds that name to a subclass whose refs-mode iteration hands the
  trainer DEVICE-RESIDENT ``TrainBatch``es staged by ``specforge_amd.ingest`` (pinned double buffer + HIP copy stream;
  the files' bytes read straight into the pinned slot) -- same refs, same order, same ``seek`` / ``set_epoch`` /
  ``drop_last`` behaviour (feature_dataloader.py:255-295), tensors bit-identical to the reference's normaliser +
  collator.  Anything the fast reader does not cover (``.ckpt.gz``, ``mem://`` refs, another algorithm's transform or
  collator, USP collation) is materialised by the reference's own ``_make_batch`` on a loader thread and staged through
  the same pinned slots.
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import torch

from .eagle3 import Eagle3TrainStrategy, OnlineEagle3Model, loss_mask_suffix_counts
from .model import DraftConfig, Eagle3DraftMethods
from .training import BF16Optimizer, HipDPTrainingBackend

ALGORITHM_NAME = "eagle3_hip"
DRAFT_ARCHITECTURE = "HipLlamaForCausalLMEagle3"
REFERENCE_ALGORITHM = "eagle3"
REFERENCE_ARCHITECTURE = "LlamaForCausalLMEagle3"

_draft_cls = {}
_installed = {}


def draft_class(name: str = DRAFT_ARCHITECTURE):
    """The HIP draft architecture under ``name`` (created once per name; needs ``specforge`` and ``transformers``).
    ``HipLlamaForCausalLMEagle3`` is registered with ``@register_draft`` side by side with the reference's class;
    the class named ``LlamaForCausalLMEagle3`` is what ``install(override=True)`` rebinds the registry entry to."""
    if name in _draft_cls:
        return _draft_cls[name]
    from transformers import LlamaConfig

    from specforge.modeling.draft.base import Eagle3DraftModel
    from specforge.modeling.draft.registry import register_draft

    class HipDraft(Eagle3DraftMethods, Eagle3DraftModel):
        config_class = LlamaConfig

        def __init__(self, config, quant_config=None, attention_backend: str = "sdpa") -> None:
            # ``attention_backend`` is the schema's closed literal (config/schema.py:523-525); the HIP TTT attention
            # implements the sdpa backend's semantics (the oracle) whichever value is configured
            Eagle3DraftModel.__init__(self, config)
            self.quant_config = quant_config
            self._build_parameters(DraftConfig.from_hf(config), attention_backend, torch.get_default_dtype(), None)

        def forward(self, *a, **kw):
            raise NotImplementedError("EAGLE3 drafts are driven through OnlineEagle3Model (TTT unroll), not forward()")

    HipDraft.__name__ = HipDraft.__qualname__ = name
    if name != REFERENCE_ARCHITECTURE:
        register_draft(HipDraft, name=name)
    _draft_cls[name] = HipDraft
    return HipDraft


class HipEagle3TrainStrategy(Eagle3TrainStrategy):
    name = ALGORITHM_NAME   # the checkpoint's ``strategy`` field must equal the algorithm name (training/trainer.py:288)


def _training_model_factory():
    import specforge.algorithms.model_providers as mp

    parts_cls = mp.AlgorithmModelParts

    def build(config, draft_model, draft_config, target_config, _tokenizer):
        """``build_eagle3_model`` (algorithms/model_providers.py:255-291) with the HIP composite"""
        t = config.training
        model = OnlineEagle3Model(draft_model=draft_model, length=t.ttt_length, attention_backend=t.attention_backend,
                                  lk_loss_type=t.lk_loss_type, kl_scale=t.kl_scale, kl_decay=t.kl_decay)
        model = model.to(device=mp._device(), dtype=mp._torch_dtype(config))
        target_head = None
        if config.mode == "offline" or (config.deployment.mode == "disaggregated" and t.role == "consumer"):
            from specforge.modeling.target.target_head import TargetHead

            target_head = TargetHead.from_pretrained(config.model.target_model_path, lm_head_key=config.model.lm_head_key,
                                                     cache_dir=config.model.cache_dir,
                                                     trust_remote_code=config.model.trust_remote_code)
        return parts_cls(model=model, target_head=target_head)

    return build


def _build_offline_reader(hidden_states_path, *, run_id, ttt_length, max_len):
    """``build_offline_reader`` (algorithms/eagle3/data.py:30-47) with this algorithm's name on the refs: the loader
    refuses refs whose strategy differs from the trainer's (feature_dataloader.py:148-157).  Same file format, same
    normaliser id."""
    from specforge.runtime.data_plane.offline_reader import OfflineManifestReader

    return OfflineManifestReader(hidden_states_path, run_id=run_id, strategy=ALGORITHM_NAME, ttt_length=ttt_length,
                                 max_len=max_len, target_repr="hidden_state")


def registration(override: bool = False):
    """``AlgorithmRegistration`` of the HIP path.

    ``override=False``: a NEW algorithm ``eagle3_hip`` on the NEW draft architecture ``HipLlamaForCausalLMEagle3`` --
    side by side with the reference's ``eagle3`` (``training.strategy: eagle3_hip``, ``architectures:
    ["HipLlamaForCausalLMEagle3"]``).
    ``override=True``: the registration NAMED ``eagle3`` on ``LlamaForCausalLMEagle3`` -- the reference's spec and data
    providers untouched, only the composite-model and step factories swapped -- so the SAME run YAML and draft-config
    JSON train on the HIP path and the checkpoints say ``strategy: eagle3`` (what ``export --to sglang`` requires,
    export/to_sglang.py:73-79).  Needs ``install(override=True)`` for the draft class."""
    from specforge.algorithms.common.providers import make_registration
    from specforge.algorithms.eagle3 import providers as ref

    spec, prov = ref.algorithm_spec(), ref.algorithm_providers()
    model = dataclasses.replace(prov.model, build_training_model=_training_model_factory())
    if override:
        strategy_cls = Eagle3TrainStrategy

        def build_step(wrapped_model, *, target_head=None, **options):
            return strategy_cls(wrapped_model, target_head=target_head, **options)

        prov = dataclasses.replace(prov, step=dataclasses.replace(prov.step, build=build_step), model=model)
        return make_registration(spec, prov)
    draft_class(DRAFT_ARCHITECTURE)
    spec = dataclasses.replace(
        spec, name=ALGORITHM_NAME,
        draft=dataclasses.replace(spec.draft, compatible_architectures=frozenset({DRAFT_ARCHITECTURE}),
                                  default_architecture=DRAFT_ARCHITECTURE))

    def build_step(wrapped_model, *, target_head=None, **options):
        return HipEagle3TrainStrategy(wrapped_model, target_head=target_head, **options)

    prov = dataclasses.replace(
        prov, algorithm_name=ALGORITHM_NAME,
        step=dataclasses.replace(prov.step, build=build_step),
        offline=tuple(dataclasses.replace(o, build_reader=_build_offline_reader) for o in prov.offline),
        model=dataclasses.replace(model, draft_config=dataclasses.replace(
            prov.model.draft_config, architecture=DRAFT_ARCHITECTURE, compatible_architectures=None)))
    return make_registration(spec, prov)


def registry(override: bool = False):
    """The catalogue to hand to ``resolve_run(cfg, registry=...)`` (application/composition.py:42-57):
    the built-in one + ``eagle3_hip`` (``AlgorithmRegistry.with_registration``, algorithms/registry.py:86-90), or --
    ``override=True`` -- the built-in one with its ``eagle3`` entry replaced by the HIP registration of that name."""
    from specforge.algorithms.builtin import builtin_algorithm_registry
    from specforge.algorithms.registry import AlgorithmRegistry

    base = builtin_algorithm_registry()
    if not override:
        return base.with_registration(registration())
    return AlgorithmRegistry([r for r in base if r.name != REFERENCE_ALGORITHM] + [registration(override=True)])


_loader_cls = {}


def _device_for_batches() -> torch.device:
    """where the trainer's batches must live: the GPU of this rank -- or, under the test suite's SIMT interpreter
    (which computes on host memory), the CPU"""
    from . import _lib

    _lib.lib()
    if _lib.is_emulated():
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def feature_loader_class():
    """``FeatureDataLoader`` (runtime/data_plane/feature_dataloader.py:92-320) with the refs-mode (offline) iteration on
    the HIP ingest.  Queue mode (online / streaming) is the reference's own code, untouched."""
    if "cls" in _loader_cls:
        return _loader_cls["cls"]
    import functools

    from specforge.algorithms.eagle3.data import normalize_offline_sample as ref_normalize
    from specforge.data.utils import DataCollatorWithPadding
    from specforge.runtime.contracts import TrainBatch as RefTrainBatch
    from specforge.runtime.data_plane.feature_dataloader import FeatureDataLoader

    from .ingest import HiddenStateIngest, PinnedStager

    class HipFeatureDataLoader(FeatureDataLoader):
        _ingest = None
        _stager = None

        def _fast_max_len(self, chunks):
            """max_len when every batch is plain EAGLE3 offline files through the reference's eagle3 normaliser and
            padding collator (what ``HiddenStateIngest`` is pinned to, tests/golden/ingest_collate.pt); else None"""
            t, c = self.per_sample_transform, self.collate_fn
            if not (isinstance(t, functools.partial) and t.func is ref_normalize and not t.args and set(t.keywords) == {"max_len"}):
                return None
            if type(c) is not DataCollatorWithPadding or getattr(c, "sp_degree", 1) != 1 or getattr(c, "ulysses_degree", 1) != 1:
                return None
            keys = {"input_ids", "loss_mask", "hidden_state", "aux_hidden_state"}
            for chunk in chunks:
                for r in chunk:
                    u = r.feature_store_uri
                    if not u.startswith("file://") or u.endswith(".gz") or set(r.feature_keys) != keys \
                            or any(k != v for k, v in r.feature_keys.items()):
                        return None
            return int(t.keywords["max_len"])

        def _iter_refs(self):
            skip, self._seek_batches = self._seek_batches, 0        # feature_dataloader.py:257-266
            chunks = []
            for start in range(skip * self.batch_size, len(self._refs), self.batch_size):
                chunk = self._refs[start:start + self.batch_size]
                if self.drop_last and len(chunk) < self.batch_size:
                    break
                chunks.append(chunk)
            if not chunks:
                return
            dev = _device_for_batches()
            max_len = self._fast_max_len(chunks)
            if max_len is None:
                yield from self._iter_staged(chunks, dev)
                return
            for chunk in chunks:
                self._validate_refs(chunk)
            ing = self._ingest
            if ing is None or ing.B != self.batch_size or ing.max_len != max_len or ing.device != dev:
                ing = self._ingest = HiddenStateIngest([], batch_size=self.batch_size, max_len=max_len, device=dev)
            groups = [[r.feature_store_uri[len("file://"):] for r in chunk] for chunk in chunks]
            for chunk, b in zip(chunks, ing.stream(groups)):
                yield RefTrainBatch(sample_ids=[r.sample_id for r in chunk], strategy=self.strategy, tensors=b.tensors,
                                    metadata={"target_repr": chunk[0].metadata.get("target_repr"),
                                              "ttt_length": chunk[0].metadata.get("ttt_length"),
                                              # (host-side row counts for the engine's loss-row compaction; the reference ignores extra keys)
                                              "loss_mask_suffix_counts": b.metadata.get("loss_mask_suffix_counts")})

        def _iter_staged(self, chunks, dev):
            """the reference's own materialisation (store.get + transform + collate) one batch ahead on a thread, then
            pinned slot + copy stream instead of the strategy's pageable ``.to(device)``"""
            from concurrent.futures import ThreadPoolExecutor

            def counted(batch):
                lm = batch.tensors.get("loss_mask")
                if lm is not None and not lm.is_cuda and isinstance(batch.metadata, dict):
                    batch.metadata.setdefault("loss_mask_suffix_counts", loss_mask_suffix_counts(lm))
                return batch

            if dev.type != "cuda":
                for chunk in chunks:
                    yield counted(self._make_batch(chunk))
                return
            if self._stager is None:
                self._stager = PinnedStager(dev)
            with ThreadPoolExecutor(max_workers=1, thread_name_prefix="sf-loader") as ex:
                fut = ex.submit(self._make_batch, chunks[0])
                for i in range(len(chunks)):
                    batch = fut.result()
                    if i + 1 < len(chunks):
                        fut = ex.submit(self._make_batch, chunks[i + 1])
                    counted(batch)
                    batch.tensors = self._stager.stage(batch.tensors)
                    yield batch

    _loader_cls["cls"] = HipFeatureDataLoader
    return HipFeatureDataLoader


class _HipBackendForTrainer(HipDPTrainingBackend):
    """constructor signature of ``FSDPTrainingBackend(parallel_config, *, optimizer_factory)`` (training/backend.py:158-165)"""

    def __init__(self, parallel_config, *, optimizer_factory=None):
        super().__init__(parallel_config, optimizer_factory=optimizer_factory,
                         process_group=getattr(parallel_config, "fsdp_process_group", None))


def install(override: bool = False) -> None:
    """Rebind the constructors the reference hard-codes (module docstring, b3 / f1): the trainer's backend, its feature
    loader and the configured optimizer factory's class; with ``override=True`` also the draft registry entry ``LlamaForCausalLMEagle3``.
    Idempotent; ``uninstall()`` restores everything (e.g. before ``export --to sglang`` materialises the reference's
    own class from the same draft config)."""
    import specforge.optimizer as ref_opt
    import specforge.training.trainer as ref_trainer
    from specforge.modeling.draft.registry import DRAFT_REGISTRY

    if "backend" not in _installed:
        _installed["backend"] = ref_trainer.FSDPTrainingBackend
        _installed["optimizer"] = ref_opt.BF16Optimizer
        _installed["loader"] = ref_trainer.FeatureDataLoader
        ref_trainer.FSDPTrainingBackend = _HipBackendForTrainer
        ref_opt.BF16Optimizer = BF16Optimizer   # _ConfiguredOptimizerFactory imports the name at call time (assembly.py:261)
        ref_trainer.FeatureDataLoader = feature_loader_class()   # trainer.py:145: the offline batches arrive device-resident
        import specforge.launch as ref_launch

        _installed["eval_loader"] = ref_launch.FeatureDataLoader
        ref_launch.FeatureDataLoader = feature_loader_class()    # launch.py:276: the offline eval loader
    if override and "draft" not in _installed:
        import specforge.modeling.draft.llama3_eagle  # noqa: F401  (makes sure the reference class is registered first)

        _installed["draft"] = DRAFT_REGISTRY[REFERENCE_ARCHITECTURE]
        DRAFT_REGISTRY[REFERENCE_ARCHITECTURE] = draft_class(REFERENCE_ARCHITECTURE)


install_backend = install


def uninstall() -> None:
    import specforge.optimizer as ref_opt
    import specforge.training.trainer as ref_trainer
    from specforge.modeling.draft.registry import DRAFT_REGISTRY

    if "backend" in _installed:
        ref_trainer.FSDPTrainingBackend = _installed.pop("backend")
        ref_opt.BF16Optimizer = _installed.pop("optimizer")
        ref_trainer.FeatureDataLoader = _installed.pop("loader")
        import specforge.launch as ref_launch

        ref_launch.FeatureDataLoader = _installed.pop("eval_loader")
    if "draft" in _installed:
        DRAFT_REGISTRY[REFERENCE_ARCHITECTURE] = _installed.pop("draft")


uninstall_backend = uninstall


def main(argv=None, *, worker_prefix=None) -> int:
    """``python -m specforge_amd.reference_plugin train -c run.yaml [--role ...] [--node-rank N] [--plan] [overrides ...]`` --
    ``specforge train`` (cli.py:167-268) on the HIP path with the SAME run YAML / draft JSON: the reference's own ``load_config``,
    ``resolve_run`` (handed the catalogue whose ``eagle3`` entry is the HIP registration), ``build_launch_plan`` and ``_train``,
    with ``install(override=True)`` in every process.  Multi-GPU topologies launch their workers through this module again
    (``worker_prefix`` / ``torchrun_prefix`` of ``build_launch_plan``, launch_plan.py:646-768), so each rank of
    ``torch.distributed.run`` installs the plugin before it builds its trainer.  Every other sub-command (``export``,
    ``benchmark``) is the reference's, untouched.  ``worker_prefix``: the argv prefix that re-enters this function in a spawned worker
    (default: ``python -m specforge_amd.reference_plugin``; the test suite's harness entry passes itself).  Single-node launches
    rendezvous on 127.0.0.1 (``--local-addr``: container host names need not resolve)."""
    import argparse
    import os
    import sys

    from specforge import cli

    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] != "train":
        return cli.main(argv)
    ap = argparse.ArgumentParser(prog="python -m specforge_amd.reference_plugin train")
    ap.add_argument("-c", "--config", required=True)
    ap.add_argument("--role", choices=("auto", "all", "producer", "consumer", "both"), default="auto")
    ap.add_argument("--node-rank", type=int, default=None)
    ap.add_argument("--plan", action="store_true")
    ap.add_argument("overrides", nargs="*")
    args = ap.parse_args(argv[1:])

    from specforge.application import bind_run, resolve_run
    from specforge.config import load_config
    from specforge.launch_plan import build_launch_plan, run_commands

    install(override=True)
    cfg = load_config(args.config, args.overrides)
    resolved = resolve_run(cfg, registry=registry(override=True))
    me = tuple(worker_prefix) if worker_prefix else (sys.executable, "-m", "specforge_amd.reference_plugin")
    plan = build_launch_plan(resolved.config, algorithm=resolved.algorithm, config_path=args.config, overrides=args.overrides,
                             requested_role=args.role, node_rank=args.node_rank, worker_prefix=me,
                             torchrun_prefix=(sys.executable, "-m", "torch.distributed.run", "--no-python", "--local-addr", "127.0.0.1"))
    if args.plan:
        print(plan.render())
        return 0
    if plan.kind == "worker":
        os.environ.update(plan.worker_env)
        role_config = cli._config_for_role(resolved.config, plan.role)
        try:
            with cli._worker_signal_unwind():
                cli._train(bind_run(role_config, resolved.algorithm))
        except cli._WorkerTermination as received:
            return 128 + received.signum
        return 0
    return run_commands(plan)


if __name__ == "__main__":
    raise SystemExit(main())
