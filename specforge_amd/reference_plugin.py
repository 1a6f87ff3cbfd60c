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
"""
from __future__ import annotations

import dataclasses
from typing import Any, Optional

import torch

from .eagle3 import Eagle3TrainStrategy, OnlineEagle3Model
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


class _HipBackendForTrainer(HipDPTrainingBackend):
    """constructor signature of ``FSDPTrainingBackend(parallel_config, *, optimizer_factory)`` (training/backend.py:158-165)"""

    def __init__(self, parallel_config, *, optimizer_factory=None):
        super().__init__(parallel_config, optimizer_factory=optimizer_factory,
                         process_group=getattr(parallel_config, "fsdp_process_group", None))


def install(override: bool = False) -> None:
    """Rebind the constructors the reference hard-codes (module docstring, b3): the trainer's backend and the configured
    optimizer factory's class; with ``override=True`` also the draft registry entry ``LlamaForCausalLMEagle3``.
    Idempotent; ``uninstall()`` restores everything (e.g. before ``export --to sglang`` materialises the reference's
    own class from the same draft config)."""
    import specforge.optimizer as ref_opt
    import specforge.training.trainer as ref_trainer
    from specforge.modeling.draft.registry import DRAFT_REGISTRY

    if "backend" not in _installed:
        _installed["backend"] = ref_trainer.FSDPTrainingBackend
        _installed["optimizer"] = ref_opt.BF16Optimizer
        ref_trainer.FSDPTrainingBackend = _HipBackendForTrainer
        ref_opt.BF16Optimizer = BF16Optimizer   # _ConfiguredOptimizerFactory imports the name at call time (assembly.py:261)
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
    if "draft" in _installed:
        DRAFT_REGISTRY[REFERENCE_ARCHITECTURE] = _installed.pop("draft")


uninstall_backend = uninstall
