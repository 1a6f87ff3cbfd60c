"""Shims for RUNNING THE REAL REFERENCE in the build container (CPU, Python 3.10).  TEST INFRASTRUCTURE ONLY.

Used by ``oracle/gen_golden.py`` / ``oracle/gen_curve.py`` (fixture generators) and by the tests that drive the HIP
path from the reference's own trainer (``tests/test_reference_trainer.py``).  ``/root/reference`` does not exist on the
GPU box: everything importing this module is gated on ``available()``.

The shims are the three of SURVEY.md section 8c plus the device knobs:
  1. ``specforge.algorithms.eagle3.model.LogSoftmaxLoss`` -> the same file's eager ``_compute_loss``
     (core/loss.py:15-21): the Triton kernel has no CPU driver;
  2. a stub ``yunchang.globals`` (``PROCESS_GROUP.{ULYSSES_PG,RING_PG}``, ``set_seq_parallel_pg``) because
     ``init_distributed`` imports it unconditionally (distributed.py:54-59,151,176);
  3. ``sys.exception`` for Python 3.10 (training/trainer.py:542);
  env: ``SPECFORGE_DEVICE=cpu``, ``FSDP_SHARDING=NO_SHARD`` (DDP over gloo), ``TORCHDYNAMO_DISABLE=1``.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"
_ready = False


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "specforge"))


def setup():
    """idempotent; returns the imported ``specforge`` package"""
    global _ready
    if not available():
        raise RuntimeError("the reference checkout is not present (build container only)")
    if not _ready:
        os.environ["SPECFORGE_DEVICE"] = "cpu"
        os.environ["TORCHDYNAMO_DISABLE"] = "1"
        os.environ.setdefault("FSDP_SHARDING", "NO_SHARD")
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        if not hasattr(sys, "exception"):
            sys.exception = lambda: sys.exc_info()[1]
        if "yunchang" not in sys.modules:
            yc = types.ModuleType("yunchang")
            g = types.ModuleType("yunchang.globals")

            class _PG:
                ULYSSES_PG = None
                RING_PG = None

            g.PROCESS_GROUP = _PG
            g.set_seq_parallel_pg = lambda *a, **k: None
            yc.globals = g
            yc.set_seq_parallel_pg = g.set_seq_parallel_pg
            sys.modules["yunchang"] = yc
            sys.modules["yunchang.globals"] = g
        import specforge.algorithms.eagle3.model as ref_model
        import specforge.core.loss as ref_loss

        class _EagerLoss:
            @staticmethod
            def apply(logits, target, mask):
                return ref_loss._compute_loss(logits, target, mask)

        ref_model.LogSoftmaxLoss = _EagerLoss
        _ready = True
    import specforge

    return specforge


def init_single_rank(port: int = 29577):
    """``init_distributed`` at world size 1 over gloo (mirrors tests/test_runtime/_fixtures.py:40-55 without CUDA)"""
    import torch.distributed as dist

    setup()
    if dist.is_available() and dist.is_initialized():
        return
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("LOCAL_RANK", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    from specforge.distributed import init_distributed

    if "MASTER_PORT" in os.environ:
        init_distributed(timeout=10, tp_size=1)
        return
    # `port` is only a hint: parallel test workers (pytest -n) must not collide.  Probing a port and binding it later is a race
    # (another worker can take it in between), so the rendezvous itself is retried on a fresh ephemeral port.
    import socket

    last = None
    for attempt in range(5):
        with socket.socket() as sk:
            try:
                sk.bind(("127.0.0.1", port if attempt == 0 else 0))
            except OSError:
                sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        try:
            init_distributed(timeout=10, tp_size=1)
            return
        except Exception as e:      # EADDRINUSE surfaces as a RuntimeError / DistNetworkError from the TCP store
            last = e
            if dist.is_initialized():
                dist.destroy_process_group()
            if "address" not in str(e).lower() and "EADDRINUSE" not in str(e):
                raise
    raise last
