"""Optimizer + data-parallel training backend of the HIP EAGLE3 path.

* ``BF16Optimizer``: same constructor arguments, ``step() -> grad_norm``, ``get_learning_rate()``
  and ``state_dict()`` keys as the reference (specforge/optimizer.py:12-232) -- fp32 masters +
  AdamW + global-L2 clip + warmup/cosine schedule -- but on the engine's flat buffers: ONE
  grad-norm reduction kernel and ONE fused clip+AdamW+bf16-cast kernel per step, no host sync
  (the clip coefficient is read from device memory by the AdamW kernel).
* ``HipDPTrainingBackend``: the ``TrainingBackend`` contract (specforge/training/backend.py:126-148)
  without FSDP/DDP: gradients of the replicated draft are all-reduced over the default process
  group (``backend="nccl"`` is RCCL over xGMI on ROCm; gloo in the CPU tests) bucket by bucket
  from inside the engine's backward sweep, so the collective of bucket i overlaps the wgrad
  GEMM of bucket i+1; non-boundary micro-steps skip the collective (DDP ``no_sync`` semantics,
  backend.py:310-320).  The 1/world of DDP's gradient averaging is folded into the grad-norm
  and AdamW kernels (no extra pass over the gradients).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from . import ops
from .eagle3 import OnlineEagle3Model


class _WarmupSchedule:
    """LR in effect after ``last_epoch`` scheduler steps of the reference's ``CosineAnnealingWarmupLR`` /
    ``ConstantWarmupLR`` (specforge/lr_scheduler.py:56-147): linear warmup ``(k+1)/W*lr``; the cosine branch
    chains torch's recursive CosineAnnealingLR started one step late, giving
    ``base*(1+cos(pi*e/T))/(1+cos(pi/T))`` with e = k-W (pinned by tests/golden/optimizer_bf16.pt).
    ``state_dict()`` / ``load_state_dict()`` use the reference scheduler's own layout (outer ``last_epoch`` stops at
    the end of the warmup, progress after that lives in ``after_scheduler_dict['last_epoch']``; lr_scheduler.py:13-52),
    so optimizer checkpoints move between the two trainers."""

    def __init__(self, kind: str, base_lr: float, total_steps: int, warmup_steps: int, eta_min: float = 0.0):
        if kind not in ("cosine", "constant"):
            raise ValueError(f"unsupported lr_scheduler={kind!r}; expected one of ['constant', 'cosine']")
        if total_steps <= 0:
            raise ValueError(f"total_steps must be positive, got {total_steps}")
        if not 0 <= warmup_steps < total_steps:
            raise ValueError(f"warmup_steps must be in [0, total_steps), got {warmup_steps} for total_steps={total_steps}")
        self.kind, self.base_lr, self.total_steps, self.warmup_steps, self.eta_min = kind, base_lr, total_steps, warmup_steps, eta_min
        self.last_epoch = 0

    def lr(self, k: Optional[int] = None) -> float:
        k = self.last_epoch if k is None else k
        W = self.warmup_steps
        if k < W:
            return (k + 1) / W * self.base_lr
        if self.kind == "constant":
            return self.base_lr
        e, T = k - W, self.total_steps - W
        if T == 1:   # torch's recursion has its own branch for T_max = 1: 2*base at the first step, then eta_min
            return self.eta_min + 2 * (self.base_lr - self.eta_min) if e == 0 else self.eta_min
        return self.eta_min + (self.base_lr - self.eta_min) * (1 + math.cos(math.pi * e / T)) / (1 + math.cos(math.pi / T))

    def step(self):
        self.last_epoch += 1

    def state_dict(self):
        k, W = self.last_epoch, self.warmup_steps
        common = {"_is_initial": False, "_get_lr_called_within_step": False}
        after = dict(base_lrs=[self.base_lr], last_epoch=max(0, k - W), _step_count=max(0, k - W) + 1,
                     _last_lr=[self.lr(k) if k > W else self.base_lr], **common)
        if self.kind == "cosine":
            after = dict(T_max=self.total_steps - W, eta_min=self.eta_min, **after)
        return dict(warmup_epochs=W, finished=k >= W, base_lrs=[self.base_lr], last_epoch=min(k, W), _step_count=min(k, W) + 1,
                    _last_lr=[self.lr(k)], after_scheduler_type="CosineAnnealingLR" if self.kind == "cosine" else "_FlatLR",
                    after_scheduler_dict=after, **common)

    def load_state_dict(self, sd):
        if "warmup_epochs" in sd and int(sd["warmup_epochs"]) != self.warmup_steps:
            raise ValueError(f"checkpoint scheduler has warmup_epochs={sd['warmup_epochs']} but this run has {self.warmup_steps}")
        k = int(sd["last_epoch"])
        if sd.get("finished") and "after_scheduler_dict" in sd:
            k = self.warmup_steps + int(sd["after_scheduler_dict"]["last_epoch"])
        self.last_epoch = k


class BF16Optimizer:
    def __init__(self, model: OnlineEagle3Model, lr, weight_decay=0.0, max_grad_norm=0.5, total_steps=800_000,
                 warmup_ratio=0.015, lr_scheduler="cosine", offload_master=False, betas=(0.9, 0.999), eps=1e-8):
        # ``offload_master`` (optimizer.py:25-35) moves the reference's fp32 masters and AdamW state to host memory to fit small
        # GPUs; it changes where the step runs, not what it computes.  On MI355X the 28 bytes per parameter stay in HBM (3.4 GB
        # of 288 at Llama-3-8B draft size), so the flag is accepted for config compatibility and has no effect on placement.
        self.offload_master = bool(offload_master)
        if self.offload_master:
            import warnings

            warnings.warn("offload_master=True: accepted for config compatibility; fp32 masters and AdamW state stay in HBM "
                          "on MI355X (same update, no host round trip)")
        self.model = model
        # ``model`` is the OnlineEagle3Model, or (the reference's convention: Trainer passes optimizer_target=
        # model.draft_model, training/trainer.py:425) its draft model, which carries a handle to the engine
        self.engine = model.engine if hasattr(model, "engine") else getattr(model, "_hip_engine", None)
        if self.engine is None:
            raise TypeError("BF16Optimizer needs a specforge_amd OnlineEagle3Model (or its draft model after the "
                            "composite was built): no HIP engine found")
        f = self.engine.flat
        # parameters that never receive a gradient are skipped like the reference skips ``p.grad is None``
        # (optimizer.py:139-142): with norm_output=False that is the final norm, laid out last in the flat buffer
        self.n_active = f.slices["norm.weight"][0] if not self.engine.cfg.norm_output else f.numel
        self.max_grad_norm = float(max_grad_norm)
        self.weight_decay, self.betas, self.eps = float(weight_decay), betas, eps
        self.master = f.data.float().clone()
        self.exp_avg = torch.zeros_like(self.master)
        self.exp_avg_sq = torch.zeros_like(self.master)
        self.norm = torch.zeros(1, device=f.data.device)
        self._ws = torch.empty(1024, device=f.data.device)
        self.step_count = 0
        self.lr_scheduler_type = lr_scheduler
        self.scheduler = _WarmupSchedule(lr_scheduler, float(lr), int(total_steps), int(warmup_ratio * total_steps))
        self.last_grad_norm = None
        self.grad_prescale = 1.0  # 1/world when the DP backend SUM-reduces (DDP averages)

    def get_learning_rate(self) -> float:
        return self.scheduler.lr()

    def step(self):
        f = self.engine.flat
        self.step_count += 1
        lr = self.scheduler.lr()
        n = self.n_active
        ops.grad_norm(f.grad[:n], self.norm, self._ws, self.grad_prescale)
        ops.adamw_step(f.grad[:n], self.master[:n], self.exp_avg[:n], self.exp_avg_sq[:n], f.data[:n], self.norm,
                       max_norm=self.max_grad_norm, lr=lr, beta1=self.betas[0], beta2=self.betas[1], eps=self.eps,
                       wd=self.weight_decay, step=self.step_count, grad_prescale=self.grad_prescale)
        self.scheduler.step()
        self.engine.end_window()
        self.last_grad_norm = self.norm[0].clone()
        return self.last_grad_norm

    # ---- reference-shaped state dict (optimizer.py:221-229; torch.optim.AdamW layout inside) ----
    def _per_param(self, flat: torch.Tensor) -> List[torch.Tensor]:
        f = self.engine.flat
        return [flat[f.slices[n][0]:f.slices[n][1]].view(f.params[n].shape) for n in f.module_order]

    def state_dict(self) -> Dict[str, Any]:
        f = self.engine.flat
        n = len(f.module_order)
        active = [f.slices[name][0] < self.n_active for name in f.module_order]   # grad-less params carry no Adam state
        state = {i: dict(step=torch.tensor(float(self.step_count)), exp_avg=m.detach().cpu().clone(),
                         exp_avg_sq=v.detach().cpu().clone())
                 for i, (m, v) in enumerate(zip(self._per_param(self.exp_avg), self._per_param(self.exp_avg_sq))) if active[i]}
        group = dict(lr=self.scheduler.lr(), betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay,
                     amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     initial_lr=self.scheduler.base_lr, params=list(range(n)))
        return {
            "optimizer_state_dict": {"state": state if self.step_count else {}, "param_groups": [group]},
            "scheduler_state_dict": self.scheduler.state_dict(),
            "lr_scheduler_type": self.lr_scheduler_type,
            "max_grad_norm": self.max_grad_norm,
            "fp32_params": [t.detach().cpu().clone() for t in self._per_param(self.master)],
        }

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        if sd.get("lr_scheduler_type", "cosine") != self.lr_scheduler_type:
            raise ValueError(f"checkpoint optimizer used lr_scheduler={sd.get('lr_scheduler_type')!r} but this run has "
                             f"lr_scheduler={self.lr_scheduler_type!r}")
        if sd.get("max_grad_norm") is not None and float(sd["max_grad_norm"]) != self.max_grad_norm:
            raise ValueError(f"checkpoint optimizer used max_grad_norm={sd['max_grad_norm']} but this run has "
                             f"max_grad_norm={self.max_grad_norm}")
        st = sd["optimizer_state_dict"]["state"]
        with torch.no_grad():
            for i, (m, v) in enumerate(zip(self._per_param(self.exp_avg), self._per_param(self.exp_avg_sq))):
                if i in st:
                    m.copy_(st[i]["exp_avg"])
                    v.copy_(st[i]["exp_avg_sq"])
                    self.step_count = int(st[i]["step"])
            if sd.get("fp32_params") is not None:
                for dst, src in zip(self._per_param(self.master), sd["fp32_params"]):
                    if dst.shape != src.shape:
                        raise ValueError(f"fp32 master param shape mismatch: checkpoint {tuple(src.shape)} vs {tuple(dst.shape)}")
                    dst.copy_(src)
            else:
                self.master.copy_(self.engine.flat.data.float())
        self.scheduler.load_state_dict(sd["scheduler_state_dict"])


@dataclass
class ParallelConfig:
    """the fields callers read (training/backend.py:30-123)"""

    world_size: int = 1
    tp_size: int = 1
    sp_size: int = 1
    dp_size: int = 1
    sharding_strategy: str = "NO_SHARD"
    fsdp_process_group: Any = None


class HipDPTrainingBackend:
    name = "hip_dp"

    def __init__(self, parallel_config: Optional[ParallelConfig] = None, *, optimizer_factory=None, process_group=None,
                 single_collective: bool = False, force_collectives: bool = False):
        """``single_collective``: one all-reduce of the whole flat gradient after the sweep instead of the overlapped
        per-bucket ones (A/B option for multi-GPU boxes).  ``force_collectives``: run the collectives even at world
        size 1 (exercises the RCCL path on a single-GPU box)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.parallel_config = parallel_config or ParallelConfig(world_size=self.world, dp_size=self.world,
                                                                 fsdp_process_group=process_group)
        self._optimizer_factory = optimizer_factory
        self.module: Optional[OnlineEagle3Model] = None
        self.optimizer: Optional[BF16Optimizer] = None
        self._handles: List[Any] = []
        self._single_collective = bool(single_collective)
        self._force_collectives = bool(force_collectives)
        self._pending_single = False
        self._sync_this_backward = True
        self.no_sync_backwards = 0  # telemetry: micro-steps that skipped the collective

    @property
    def optimizer_state_is_replicated(self) -> bool:
        return True

    def prepare_model(self, model: OnlineEagle3Model, *, wrap: bool = True, optimizer_target=None) -> nn.Module:
        self.module = model
        use_dp = self.world > 1 or (self._force_collectives and dist.is_available() and dist.is_initialized())
        model.engine.on_bucket_ready = self._bucket_ready if use_dp else None
        if use_dp:  # replicas must start identical (DDP broadcasts rank 0's parameters)
            dist.broadcast(model.engine.flat.data, src=0, group=self.group)
        if self._optimizer_factory is not None:
            # the reference's Trainer passes optimizer_target=model.draft_model (training/trainer.py:425)
            model.engine   # adopt the parameters into the flat buffers before any optimizer clones its masters
            self.set_optimizer(self._optimizer_factory(optimizer_target if optimizer_target is not None else model))
        return model

    def set_optimizer(self, optimizer) -> None:
        """``BF16Optimizer`` of this package (fused; folds DDP's 1/world into its kernels) or any optimizer with the
        reference's ``step() -> grad_norm`` / ``state_dict`` surface working on the draft's parameters and ``.grad``
        (e.g. the reference's own ``specforge.optimizer.BF16Optimizer``: slower, per-parameter, but equivalent)."""
        self.optimizer = optimizer
        self._native_optimizer = isinstance(optimizer, BF16Optimizer)
        if self._native_optimizer:
            optimizer.grad_prescale = 1.0 / self.world

    def _bucket_ready(self, lo: int, hi: int) -> None:
        if not self._sync_this_backward:
            return
        if self._single_collective:      # one all-reduce of the whole flat gradient after the sweep
            self._pending_single = True
            return
        # async SUM over RCCL: enqueued behind the wgrad GEMM that produced the bucket, runs on the
        # communicator's stream while the compute stream continues with the next GEMM
        ev = getattr(self, "bucket_events", None)     # telemetry (bench.py --gpus N): see bucket_timeline()
        if ev is not None and torch.cuda.is_available():
            ready = torch.cuda.Event(enable_timing=True)
            ready.record()                            # fires when the bucket's last weight-gradient GEMM has finished
        h = dist.all_reduce(self.module.engine.flat.grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._handles.append(h)
        if ev is not None and torch.cuda.is_available():
            # when did the collective END: a probe stream that waits for nothing but this Work, then stamps an event
            if getattr(self, "_probe_stream", None) is None:
                self._probe_stream = torch.cuda.Stream()
            done = torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(self._probe_stream):
                h.wait()
                done.record()
            ev.append((lo, hi, ready, done))

    def backward(self, loss: torch.Tensor, *, is_boundary: bool = True) -> None:
        self._sync_this_backward = bool(is_boundary)
        if not is_boundary and self.world > 1:
            self.no_sync_backwards += 1
        loss.backward()

    def scale_gradients(self, factor) -> None:
        raise NotImplementedError("loss_terms strategies (DFlash) are outside the EAGLE3 path")

    def synchronize_gradients(self) -> None:
        """wait for the bucket all-reduces launched by the last boundary backward"""
        if self._pending_single:
            dist.all_reduce(self.module.engine.flat.grad, op=dist.ReduceOp.SUM, group=self.group)
            self._pending_single = False
            self._handles.clear()
            return True
        reduced = bool(self._handles)
        ev = getattr(self, "comm_wait_events", None)     # telemetry (bench.py): how long the compute stream waits here
        if ev is not None and reduced and torch.cuda.is_available():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in self._handles:
            h.wait()
        if ev is not None and reduced and torch.cuda.is_available():
            e1.record()
            ev.append((e0, e1))
        self._handles.clear()
        return reduced

    def bucket_timeline(self, events=None):
        """[(mbytes, ready_to_done_ms)] of the bucket all-reduces recorded while ``bucket_events`` was a list: the time from "this bucket's
        weight gradient is complete" (an event right behind its last GEMM on the compute stream) to "its all-reduce has finished".  Beside the same all-reduce timed alone this is the wait for a CU (a persistent TN workgroup holds its
        CU for a whole tile) plus the slowdown from sharing the chip with the next weight-gradient GEMM."""
        out = []
        for lo, hi, ready, done in (events if events is not None else getattr(self, "bucket_events", None)) or []:
            done.synchronize()
            out.append(((hi - lo) * 2 / 1e6, ready.elapsed_time(done)))
        return out

    def step(self):
        reduced = self.synchronize_gradients()
        if not getattr(self, "_native_optimizer", True):
            eng = self.module.engine
            if self.world > 1 and reduced:
                eng.flat.grad.mul_(1.0 / self.world)       # DDP averages; a foreign optimizer sees plain mean gradients
            out = self.optimizer.step()
            eng.end_window()
            eng.flat.realias_grads()
            return out
        return self.optimizer.step()

    def state_dict(self) -> dict:
        sd = {k: v.detach().clone() for k, v in self.module.state_dict().items() if not k.startswith("_anchor")}
        dev = self.module.engine.dev
        rng = {"torch": torch.get_rng_state(), "device_type": "cuda" if dev.type == "cuda" else "cpu",   # backend.py:390-412
               "cuda": torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None, "npu": None}
        return {"model": sd, "optimizer": self.optimizer.state_dict() if self.optimizer else None, "rng": rng}

    def load_state_dict(self, state: dict) -> None:
        """restores whichever of module weights / optimizer / RNG the state carries (backend.py:352-360)"""
        if state.get("model") is not None:
            own = self.module.state_dict()
            with torch.no_grad():
                for k, v in state["model"].items():
                    own[k].copy_(v)
        if self.optimizer is not None and state.get("optimizer") is not None:
            self.optimizer.load_state_dict(state["optimizer"])
            if getattr(self, "_native_optimizer", True):
                n = self.optimizer.n_active
                self.module.engine.flat.data[:n].copy_(self.optimizer.master[:n])
        self.module.engine.weights_version += 1
        rng = state.get("rng") or {}
        cpu_state = rng.get("torch", rng.get("cpu"))
        if cpu_state is not None:
            torch.set_rng_state(cpu_state)
        if rng.get("cuda") is not None and self.module.engine.dev.type == "cuda":
            torch.cuda.set_rng_state(rng["cuda"], self.module.engine.dev)


@dataclass
class StepResult:
    """what ``TrainerCore.train_step`` hands back (specforge/training/controller.py:232-262)"""

    loss: torch.Tensor                    # un-divided micro-step loss (detached)
    metrics: Dict[str, Any]
    grad_norm: Optional[torch.Tensor]     # set on optimizer boundaries
    stepped: bool


class TrainerCore:
    """The micro-step protocol of the reference trainer (specforge/training/controller.py:328-363) for the EAGLE3
    strategy: loss / accumulation_steps -> ``backend.backward(is_boundary)`` (the gradient all-reduce is skipped on
    non-boundary micro-steps, backend.py:310-320) -> ``backend.step()`` on the boundary."""

    def __init__(self, strategy, backend: "HipDPTrainingBackend", *, accumulation_steps: int = 1):
        if accumulation_steps < 1:
            raise ValueError("accumulation_steps must be >= 1")
        self.strategy, self.backend, self.accumulation_steps = strategy, backend, int(accumulation_steps)
        self._micro = 0
        self.global_step = 0

    def train_step(self, batch, ctx=None) -> StepResult:
        out = self.strategy.forward_loss(batch, ctx)
        if out.loss_terms is not None:
            raise NotImplementedError("loss_terms strategies (DFlash) are outside the EAGLE3 path")
        loss = out.loss / self.accumulation_steps
        self._micro += 1
        stepped = self._micro % self.accumulation_steps == 0
        self.backend.backward(loss, is_boundary=stepped)
        grad_norm = self.backend.step() if stepped else None
        if stepped:
            self.global_step += 1
        return StepResult(loss=out.loss.detach(), metrics=out.metrics, grad_norm=grad_norm, stepped=stepped)


def distributed_sampler_indices(size: int, *, dp_rank: int, dp_size: int, seed: int, epoch: int, shuffle: bool = True):
    """``_distributed_sampler_indices`` (specforge/launch.py:219-239): == torch DistributedSampler."""
    if size <= 0:
        return []
    if shuffle:
        g = torch.Generator()
        g.manual_seed(int(seed) + int(epoch))
        idx = torch.randperm(size, generator=g).tolist()
    else:
        idx = list(range(size))
    total = math.ceil(size / dp_size) * dp_size
    pad = total - len(idx)
    if pad:
        reps = math.ceil(pad / len(idx))
        idx.extend((idx * reps)[:pad])
    return idx[dp_rank:total:dp_size]
