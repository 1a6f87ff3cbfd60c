"""TEST HARNESS process entry: ``python tests/_plugin_worker.py train -c run.yaml`` = ``python -m specforge_amd.reference_plugin
train -c run.yaml`` (i.e. ``specforge train`` on the HIP path) inside the build container -- the reference imported from
/root/reference with the CPU shims of ``oracle/ref_harness.py``, the kernels on the SIMT interpreter.  The launcher process AND every
``torch.distributed.run`` worker it spawns come through here (``worker_prefix``), so each rank installs shims + plugin + interpreter.

Probe: with ``SF_TEST_PROBE_DIR`` set, every trainer rank leaves ``probe_rank{r}.json`` -- sha256 of its flat bf16 weights after the
run, how many backward calls skipped the gradient collective (``no_sync``), how many did not, and how many all-reduces it issued.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main() -> int:
    from oracle import ref_harness as RH

    RH.setup()
    from specforge_amd import _lib, build
    from specforge_amd import reference_plugin as RP
    from specforge_amd import training as T

    _lib._inject_library_for_tests(build.build_emu())
    probe_dir = os.environ.get("SF_TEST_PROBE_DIR")
    backends = []
    if probe_dir:
        init0, backward0, ready0 = RP._HipBackendForTrainer.__init__, T.HipDPTrainingBackend.backward, T.HipDPTrainingBackend._bucket_ready

        def init(self, *a, **kw):
            init0(self, *a, **kw)
            self._probe = dict(boundary_backwards=0, skipped_backwards=0, bucket_allreduces=0)
            backends.append(self)

        def backward(self, loss, *, is_boundary=True):
            self._probe["boundary_backwards" if is_boundary else "skipped_backwards"] += 1
            return backward0(self, loss, is_boundary=is_boundary)

        def ready(self, lo, hi):
            n = len(self._handles)
            ready0(self, lo, hi)
            self._probe["bucket_allreduces"] += len(self._handles) - n

        RP._HipBackendForTrainer.__init__ = init
        T.HipDPTrainingBackend.backward = backward
        T.HipDPTrainingBackend._bucket_ready = ready
    rc = RP.main(sys.argv[1:], worker_prefix=(sys.executable, os.path.abspath(__file__)))
    if probe_dir and backends:
        be = backends[-1]
        flat = be.module.engine.flat.data.detach().cpu().contiguous()
        rank = int(os.environ.get("RANK", "0"))
        rec = dict(be._probe, rank=rank, world=be.world, no_sync_backwards=be.no_sync_backwards,
                   weights_sha256=hashlib.sha256(flat.view(__import__("torch").int16).numpy().tobytes()).hexdigest(),
                   lr=be.optimizer.get_learning_rate() if be.optimizer else None)
        with open(os.path.join(probe_dir, f"probe_rank{rank}.json"), "w") as f:
            json.dump(rec, f)
    return rc


if __name__ == "__main__":
    raise SystemExit(main())
