"""Drop-in ``OnlineEagle3Model`` + ``Eagle3TrainStrategy`` on the HIP engine.

Same constructor arguments, ``forward`` keyword arguments and 7-tuple of per-TTT-step lists as
the reference (specforge/algorithms/eagle3/model.py:100-442) and the same
``forward_loss(batch) -> StepOutput`` contract (specforge/training/strategies/base.py:124-319),
so ``TrainerCore.train_step`` (training/controller.py:328-363) and ``_reduce_eagle3_metrics``
(controller.py:200-304) drive it unchanged.  The whole micro-step is ONE autograd node:
``loss.backward()`` triggers ``Eagle3Engine.backward`` which writes straight into the flat
gradient buffer the parameters' ``.grad`` alias.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from .engine import Eagle3Engine
from .model import LlamaForCausalLMEagle3


def loss_mask_suffix_counts(loss_mask: torch.Tensor, n: int = ops.MAX_DIAG + 1):
    """[sum(loss_mask[:, k:]) for k in range(n)] as python ints, from a HOST copy of the [B, S] loss mask: TTT step k scores row
    (b, s) against the mask at position s + k (eagle3/model.py:364-433), so these are the per-step numbers of rows that carry a loss.
    Computed where the mask still is in host memory (the ingest's pinned slot, a CPU batch): never a device read-back."""
    assert not loss_mask.is_cuda
    col = (loss_mask.reshape(loss_mask.shape[0], -1) != 0).sum(dim=0).flip(0).cumsum(0).flip(0).tolist()   # suffix sums over positions
    return [int(col[k]) if k < len(col) else 0 for k in range(n)]


def padding_left_shift(t: torch.Tensor) -> torch.Tensor:
    """``padding(tensor, left=False)`` (specforge/utils.py:128-135): shift left by one along dim 1, zero fill."""
    return torch.cat((t[:, 1:], torch.zeros_like(t[:, -1:])), dim=1)


class _TTTStep(torch.autograd.Function):
    """plosses[T] as one node; backward = the engine's hand-written sweep."""

    @staticmethod
    def forward(ctx, anchor, engine, plosses):
        ctx.engine = engine
        return plosses.clone()

    @staticmethod
    def backward(ctx, grad_plosses):
        eng: Eagle3Engine = ctx.engine
        # The upstream gradient lives on the device.  It is only needed as the alpha of the weight-gradient GEMMs at the
        # end of the sweep, so it is copied to pinned memory asynchronously and read back THERE: the host queues the whole
        # data-gradient sweep first and the read-back waits behind ~half a step of queued GPU work instead of stalling
        # the launch of the backward behind the forward.
        gp_dev = grad_plosses.detach().float()
        nt = gp_dev.numel()
        # the engine's device-side input checks (host-supplied row counts, position ids) ride along -- always, so the pinned
        # buffer keeps ONE shape whether or not a batch was checked
        gp_dev = torch.cat([gp_dev.view(-1), eng._flags.to(gp_dev.device)])
        if gp_dev.is_cuda:
            host = getattr(eng, "_upstream_host", None)      # one pinned buffer for the engine's lifetime: allocating
            if host is None or host.shape != gp_dev.shape:   # pinned memory per step costs milliseconds of idle GPU
                host = eng._upstream_host = torch.empty(gp_dev.shape, dtype=torch.float32, pin_memory=True)
            host.copy_(gp_dev, non_blocking=True)
            ready = torch.cuda.Event()
            ready.record()
        else:
            host, ready = gp_dev, None

        def resolve():
            if ready is not None:
                ready.synchronize()
            gp = host.tolist()
            eng.check_flags(gp[nt:])
            gp = gp[:nt]
            g = gp[0]
            # The decay weights are baked into the fused CE gradients; the caller must use the same ones.
            for k, v in enumerate(gp):
                want = g * (eng.decay ** k)
                if abs(v - want) > 1e-6 * max(1.0, abs(want)):
                    raise RuntimeError(
                        f"ploss weights seen in backward ({gp}) are not g*ploss_decay^k with ploss_decay={eng.decay}; "
                        "construct OnlineEagle3Model(ploss_decay=...) with the strategy's value")
            return g

        eng.backward(resolve)
        return None, None, None


class OnlineEagle3Model(nn.Module):
    """EAGLE3 TTT trainer module (reference: eagle3/model.py:100-442) on the HIP engine."""

    def __init__(self, draft_model: LlamaForCausalLMEagle3, length: int = 7, attention_backend: str = "hip",
                 lk_loss_type: Optional[str] = None, kl_scale: float = 1.0, kl_decay: float = 1.0,
                 ploss_decay: float = 0.8):
        super().__init__()
        if lk_loss_type not in (None, "alpha", "lambda"):
            raise ValueError(f"Unknown lk loss type: {lk_loss_type}")  # core/lk_loss.py:99
        if not 1 <= int(length) <= ops.MAX_DIAG + 1:
            raise ValueError(f"ttt_length must be in 1..{ops.MAX_DIAG + 1}")
        self.draft_model = draft_model
        self.length = length
        self.attention_backend = attention_backend
        self.lk_loss_type, self.kl_scale, self.kl_decay = lk_loss_type, kl_scale, kl_decay
        self.ploss_decay = ploss_decay
        self._engine: Optional[Eagle3Engine] = None
        dev = next(draft_model.parameters()).device
        self._anchor = nn.Parameter(torch.zeros((), device=dev), requires_grad=True)

    @property
    def engine(self) -> Eagle3Engine:
        """Built on first use: the reference's assembly moves the composite with ``.to(device, dtype)`` AFTER
        constructing it (algorithms/model_providers.py:268-275); adopting the parameters into the flat buffer any
        earlier would be undone by that move."""
        if self._engine is None:
            self._engine = Eagle3Engine(self.draft_model, ttt_length=self.length, ploss_decay=self.ploss_decay,
                                        lk_loss_type=self.lk_loss_type, kl_scale=self.kl_scale, kl_decay=self.kl_decay)
            object.__setattr__(self.draft_model, "_hip_engine", self._engine)   # found by BF16Optimizer(draft_model)
        return self._engine

    def _apply(self, fn, *a, **kw):
        if self._engine is not None:
            before = self._engine.flat.data.data_ptr()
            super()._apply(fn, *a, **kw)
            if any(p.data_ptr() != self._engine.flat.data[lo:hi].data_ptr()
                   for n, p in self._engine.flat.params.items() for lo, hi in [self._engine.flat.slices[n]]):
                raise RuntimeError("OnlineEagle3Model was moved/cast after its HIP engine adopted the parameters; move the "
                                   "model to its device and dtype before the first forward / optimizer construction")
            assert before == self._engine.flat.data.data_ptr()
            return self
        return super()._apply(fn, *a, **kw)

    def forward(self, input_ids, attention_mask, target=None, loss_mask=None, hidden_states=None, past_key_values=None,
                position_ids=None, target_hidden_for_compact=None, target_head_weight=None,
                compact_teacher_chunk_size: Optional[int] = None, loss_counts=None, position_span=None):
        """``loss_counts`` (optional, not in the reference's signature): loss_counts[k] = number of rows whose loss mask at position
        s + k is set, as HOST integers -- lets the engine run lm_head / CE on those rows only (``loss_mask_suffix_counts``).
        ``position_span`` (optional): (min, max) of ``position_ids`` as host integers (range check / RoPE table growth without a
        device read-back); CPU ``position_ids`` are measured here."""
        if past_key_values is not None:
            raise NotImplementedError("past_key_values is unused by EAGLE3 training (eagle3/model.py:262)")
        train = torch.is_grad_enabled() and self.training
        dev = self.engine.dev
        if position_span is None and position_ids is not None and not position_ids.is_cuda and position_ids.numel():
            position_span = (int(position_ids.min()), int(position_ids.max()))
        to = lambda t: None if t is None else t.to(dev)
        out = self.engine.forward(
            input_ids=to(input_ids), attention_mask=attention_mask, loss_mask=to(loss_mask), hidden_states=to(hidden_states),
            target_hidden=to(target_hidden_for_compact), target_head_weight=target_head_weight,
            target_logits=to(target) if target_hidden_for_compact is None else None, position_ids=to(position_ids), train=train,
            loss_counts=loss_counts, position_span=position_span)
        plosses = torch.stack(out["plosses"])
        if train:
            plosses = _TTTStep.apply(self._anchor, self.engine, plosses)
        self.last_artifacts = dict(target_token_ids=out["target_token_ids"], position_mask=out["position_mask"])
        return (list(plosses.unbind(0)), out["acceptance_rates"], out["acces"], out["acc_corrects"], out["acc_denoms"],
                out["metric_losses"], out["metric_loss_denoms"])


@dataclass(frozen=True)
class StepOutput:
    """training/strategies/base.py:29-42"""

    loss: torch.Tensor
    metrics: Dict[str, Any]
    ratio_metrics: Dict[str, Tuple[Any, Any]] = field(default_factory=dict)
    loss_terms: Optional[Tuple[torch.Tensor, torch.Tensor]] = None


class TargetHead(nn.Module):
    """Frozen teacher projection (specforge/modeling/target/target_head.py:15-108): ``fc`` H -> V_target."""

    def __init__(self, weight: torch.Tensor):
        super().__init__()
        self.fc = nn.Linear(weight.shape[1], weight.shape[0], bias=False, device=weight.device, dtype=weight.dtype)
        with torch.no_grad():
            self.fc.weight.copy_(weight)
        self.fc.weight.requires_grad = False

    @staticmethod
    def preprocess(input_ids, target, loss_mask):
        """target_head.py:103-108: shift target and input_ids left by one, loss_mask -> [B,S,1]"""
        return padding_left_shift(input_ids), padding_left_shift(target), loss_mask[..., None]


class Eagle3TrainStrategy:
    """training/strategies/base.py:124-319.  The teacher always runs the streaming path
    (target hidden state + frozen head weight -> soft targets; [B,S,V_target] is never kept)."""

    name = "eagle3"
    required_features = {"input_ids", "attention_mask", "loss_mask", "hidden_state", "target"}

    def __init__(self, eagle3_model: OnlineEagle3Model, *, target_head: Optional[TargetHead] = None,
                 ploss_decay: float = 0.8, compact_teacher: bool = True, compact_teacher_chunk_size: Optional[int] = None,
                 pinned_staging: bool = True):
        """``pinned_staging=False`` (A/B only, ``bench.py --feed cpu_batch_pageable``): CPU batches are shifted on the CPU and
        moved with the blocking pageable ``.to(device)`` the reference's strategy uses (strategies/base.py:256-284)."""
        self.pinned_staging = pinned_staging
        self.eagle3_model = eagle3_model
        self.target_head = target_head
        self.ploss_decay = ploss_decay
        self._stager = None
        if abs(eagle3_model.ploss_decay - ploss_decay) > 1e-12:
            raise ValueError("OnlineEagle3Model.ploss_decay and the strategy's ploss_decay must agree")

    def trainable_module(self) -> nn.Module:
        return self.eagle3_model

    def validate_batch(self, batch) -> None:
        missing = self.required_features - set(batch.tensors)
        if missing:
            raise ValueError(f"eagle3 strategy: batch lacks {sorted(missing)}")

    def _resident(self, tensors):
        """The batch on the engine's device.  Batches from the HIP ingest (``reference_plugin.feature_loader_class``,
        ``ingest.HiddenStateIngest``) already are; a CPU batch (the reference's own loader hands those over and moves them with
        a blocking pageable ``.to(device)``, strategies/base.py:270-289) goes through a pinned double buffer and a HIP copy
        stream -- the TTT shift of ``preprocess`` then runs on the device either way."""
        dev = self.eagle3_model.engine.dev
        if dev.type != "cuda" or not self.pinned_staging or all(v is None or v.is_cuda for v in tensors.values()):
            return tensors
        if self._stager is None:
            from .ingest import PinnedStager

            self._stager = PinnedStager(dev)
        return self._stager.stage(tensors)

    def forward_loss(self, batch, ctx=None) -> StepOutput:
        self.validate_batch(batch)
        # rows per TTT step that carry a loss (host integers): from the loader's metadata, or from the mask while it is still a CPU tensor
        md = getattr(batch, "metadata", None) or {}
        counts = md.get("loss_mask_suffix_counts")
        lm0 = batch.tensors.get("loss_mask")
        if counts is None and lm0 is not None and not lm0.is_cuda:
            counts = loss_mask_suffix_counts(lm0)
        pid0 = batch.tensors.get("position_ids")
        span = (int(pid0.min()), int(pid0.max())) if pid0 is not None and not pid0.is_cuda and pid0.numel() else None
        t = self._resident(batch.tensors)
        target_repr = md.get("target_repr")
        kwargs = {}
        if target_repr == "hidden_state":
            if self.target_head is None:
                raise ValueError("target_repr='hidden_state' requires a target_head to re-run the lm_head projection")
            input_ids, target_hidden, loss_mask = TargetHead.preprocess(t["input_ids"], t["target"], t["loss_mask"])
            kwargs = dict(target_hidden_for_compact=target_hidden, target_head_weight=self.target_head.fc.weight.data)
            target = None
        else:
            # logits (or any other / missing target_repr): online capture has already shifted logits and input ids, so
            # they are used exactly as delivered (_prepare_eagle_target, strategies/base.py:95-121)
            input_ids, target, loss_mask = t["input_ids"], t["target"], t["loss_mask"]
        plosses, acceptance_rates, acces, acc_corrects, acc_denoms, metric_losses, metric_loss_denoms = self.eagle3_model(
            input_ids=input_ids, attention_mask=t["attention_mask"], loss_mask=loss_mask, target=target,
            hidden_states=t["hidden_state"], position_ids=t.get("position_ids"), loss_counts=counts, position_span=span, **kwargs)
        weights = [self.ploss_decay ** i for i in range(len(plosses))]
        loss = sum(weights[i] * plosses[i] for i in range(len(plosses)))
        d = lambda xs: [x.detach() for x in xs]
        return StepOutput(loss=loss, metrics=dict(plosses=d(plosses), acces=d(acces), acceptance_rates=d(acceptance_rates),
                                                  acc_corrects=d(acc_corrects), acc_denoms=d(acc_denoms),
                                                  metric_losses=d(metric_losses), metric_loss_denoms=d(metric_loss_denoms)))

    def checkpoint_state_filter(self, state_dict: Dict[str, Any]) -> Dict[str, Any]:
        """strip ``draft_model.``, drop the frozen embedding and engine internals (strategies/base.py:306-319)"""
        return {k.replace("draft_model.", ""): v for k, v in state_dict.items()
                if "draft_model." in k and "embed" not in k.lower()}


@dataclass
class TrainBatch:
    """runtime/contracts.py:80-129 (the fields the strategy reads)"""

    tensors: Dict[str, torch.Tensor]
    metadata: Dict[str, Any] = field(default_factory=dict)
