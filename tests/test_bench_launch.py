"""The bench's own launch paths, so that the driver's cold ``--gpus N`` run cannot fail for a reason a 1-GPU box can catch
(VERDICT r3 next #2).  All cases run ``bench.py`` as a SUBPROCESS exactly as the driver does and parse its ONE JSON line.

* bare ``python bench.py --gpus 2``: the self-launch (re-exec under torch.distributed.run on 127.0.0.1, free port), two ranks
  sharing the box's single GPU through gloo (``--share-gpu``: test only) -- both ranks' tokens counted, bucketed async
  all-reduces issued from inside the weight-gradient phase, their waits timed.
* ``python -m torch.distributed.run --nproc-per-node 1 ... bench.py --gpus 1 --dist-backend nccl --force-dp``: the driver's own
  command line with RCCL as the backend (world 1: the only RCCL world a 1-GPU box has).
* ``--feed ingest`` / the feed table, and ``--config`` for every BASELINE.json configuration name (tiny dims via ``--small``
  would hide the real dims, so those run in the measurement bundle, not here -- here only the argument plumbing).
Reference semantics being exercised: DDP bucketed all-reduce + no_sync (training/backend.py:233-253,310-320).
"""
import json
import math
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _run(cmd, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # rank 0 prints ONE JSON line, the other ranks none
    return json.loads(lines[0])


def _check_line(line, *, world, B, S, steps):
    assert line["n_gpus"] == world and line["steps"] == steps and line["scaling"] == "weak" and line["unit"] == "tokens/s"
    assert math.isfinite(line["final_loss"]) and line["final_loss"] > 0
    # whole-job aggregate: every rank's tokens over the max-over-ranks time
    want = world * B * S * steps / (line["ms_per_step"] * steps / 1e3)
    assert abs(line["value"] - want) <= 1e-6 * want
    assert line["config"]["global_batch"] == world * B and line["config"]["parallelism"] == f"dp{world}"
    assert 0 < line["roofline"]["frac"] < 1 and line["roofline"]["launches_per_step"] > 0


def test_bare_gpus2_self_launch_two_ranks_over_gloo():
    line = _run([sys.executable, "bench.py", "--gpus", "2", "--small", "--batch", "2", "--seq", "512", "--dist-backend", "gloo",
                 "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    _check_line(line, world=2, B=2, S=512, steps=2)
    r = line["rccl"]
    assert r["rccl_ranks"] == 2 and r["backend"] == "gloo" and not r["single_collective"]
    assert len(r["buckets"]) >= 2 and all(b["ms"] > 0 and b["mbytes"] > 0 for b in r["buckets"])
    assert r["exposed_wait_ms_per_step"] is not None and r["exposed_wait_ms_per_step"] >= 0     # the waits were timed inside the steps
    assert "dense_mask" not in line and "feeds" not in line and "cpu_baseline" not in line       # N > 1: the timed region only


def test_bare_gpus8_self_launch_eight_ranks_over_gloo():
    """world 8 -- the driver's largest scaling point -- without 8 GPUs: eight ranks share the box's one GPU through gloo.  The cold run must
    not die: one JSON line, all eight ranks' tokens counted, the 7 gradient buckets all-reduced and their waits timed, per-rank spread
    reported (a straggler would show in the scaling record)."""
    line = _run([sys.executable, "bench.py", "--gpus", "8", "--small", "--batch", "1", "--seq", "256", "--dist-backend", "gloo",
                 "--share-gpu", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], timeout=1500)
    _check_line(line, world=8, B=1, S=256, steps=2)
    r = line["rccl"]
    assert r["rccl_ranks"] == 8 and r["backend"] == "gloo" and len(r["buckets"]) == 7
    assert r["exposed_wait_ms_per_step"] is not None and r["exposed_wait_ms_per_step"] >= 0
    sp = line["rank_ms_per_step"]
    assert 0 < sp["min"] <= sp["max"] and abs(sp["max"] - line["ms_per_step"]) <= 1e-6 * sp["max"]
    assert "configs" not in line and "feeds" not in line and "cpu_baseline" not in line


def test_eight_concurrent_ingest_processes_smoke():
    """eight loader processes (one per would-be rank of a node) over shards of the same feature files into their own pinned slots and on
    to the GPU, at small size: what the ranks of an 8-GPU run do to one host (tools/ingest_procs.py is the measurement at full size)"""
    r = subprocess.run([sys.executable, "tools/ingest_procs.py", "--procs", "8", "--files", "64", "--seq", "128", "--hidden", "256"], cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    rep = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rep["procs"] == 8 and rep["device"]["aggregate_GBps"] > 0 and len(rep["device"]["per_proc_GBps"]) == 8


def test_default_line_carries_the_configs_legs():
    """plumbing of the `configs` object (cfg 3 / 4 / 5 + the batch-1 recipe shape beside the headline): with --small dims every leg
    runs the tiny model at its own (batch, seq); the real dims are what the default `python bench.py` line runs"""
    line = _run([sys.executable, "bench.py", "--small", "--configs", "--config-steps", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                 "--seq", "512", "--no-cpu-baseline", "--no-dense-mask", "--no-feeds"], timeout=1500)
    cf = line["configs"]
    sys.path.insert(0, ROOT)
    import bench

    assert {k for k, *_ in bench.CONFIG_LEGS} <= set(cf)
    for key, cname, B, S, what in bench.CONFIG_LEGS:
        leg = cf[key]
        assert leg.get("error") is None, leg
        assert leg["batch"] == B and leg["seq_len"] == S and leg["ms_per_step"] > 0 and 0 < leg["draft_frac"] < 1 and 0 < leg["nt_frac"] < 1
        assert abs(leg["tokens_per_s"] - B * S / (leg["ms_per_step"] / 1e3)) <= 1e-6 * leg["tokens_per_s"]
    # the short forms inside `roofline` (the object the driver's record keeps whole): every kernel of `kernels`, every leg of `configs`
    rf = line["roofline"]
    assert set(rf["by_kernel"]) == set(line["kernels"]) | {"gemm_nt"}
    for k, (frac, ms) in rf["by_kernel"].items():
        assert 0 < frac < 1.5 and ms > 0, (k, frac, ms)
    assert set(rf["by_config"]) == {k for k, *_ in bench.CONFIG_LEGS}
    for k, (ms, tps, dfrac, ntfrac, others) in rf["by_config"].items():
        assert abs(ms - cf[k]["ms_per_step"]) < 0.01 and abs(dfrac - cf[k]["draft_frac"]) < 1e-3 and isinstance(others, dict)


def test_driver_command_line_torchrun_rccl_world1():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    line = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                 "--master-port", str(port), "bench.py", "--gpus", "1", "--small", "--batch", "2", "--seq", "512", "--dist-backend", "nccl",
                 "--force-dp", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-dense-mask", "--no-feeds"])
    _check_line(line, world=1, B=2, S=512, steps=2)
    r = line["rccl"]
    assert r["rccl_ranks"] == 1 and r["backend"] == "nccl" and len(r["buckets"]) >= 2
    assert r["exposed_wait_ms_per_step"] is not None


def test_sparse_loss_mask_legs_and_option():
    """the default line carries the sparse-loss-mask pair (loss-row compaction on / off) beside the headline; ``--loss-mask-density``
    makes it the timed workload (labelled as not the headline), ``--no-compact`` its dense form: same loss up to bf16 summation order"""
    line = _run([sys.executable, "bench.py", "--small", "--batch", "2", "--seq", "512", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                 "--no-dense-mask"])
    sp = line["sparse_loss_mask"]
    assert 0.3 < sp["loss_mask_density"] < 0.7 and sp["compact"]["ms_per_step"] > 0 and sp["dense"]["ms_per_step"] > 0
    runs = []
    for extra in ([], ["--no-compact"]):
        l2 = _run([sys.executable, "bench.py", "--small", "--batch", "2", "--seq", "512", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--no-dense-mask", "--no-feeds", "--loss-mask-density", "0.5"] + extra)
        _check_line(l2, world=1, B=2, S=512, steps=2)
        assert "NOT the headline" in l2["config"]["workload"] and "sparse_loss_mask" not in l2
        runs.append(l2["final_loss"])
    assert abs(runs[0] - runs[1]) <= 2e-2 * abs(runs[1])


def test_feed_table_and_ingest_fed_timed_region():
    line = _run([sys.executable, "bench.py", "--small", "--batch", "2", "--seq", "512", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                 "--no-dense-mask", "--feed", "ingest"])
    _check_line(line, world=1, B=2, S=512, steps=3)
    assert line["feed"] == "ingest"
    f = line["feeds"]
    assert {"hbm", "ingest", "cpu_batch", "cpu_batch_pageable"} <= set(f)
    assert all(f[k]["ms_per_step"] > 0 for k in ("hbm", "ingest", "cpu_batch", "cpu_batch_pageable"))
    assert not any(n.startswith("sf_bench_feed_") for n in os.listdir("/tmp"))                   # the feature files were removed


@pytest.mark.parametrize("config", ["qwen3-8b", "qwen3-30b-a3b-eagle31", "deepseek-v3"])
def test_config_names_resolve(config):
    """argument plumbing of --config (dims are the reference's configs/*.json; the measured lines are profiles/r4_bench_<cfg>.json)"""
    sys.path.insert(0, ROOT)
    import bench

    dims, B, S, label = bench.CONFIGS[config]
    assert dims["hidden_size"] % 64 == 0 and dims["num_attention_heads"] % dims["num_key_value_heads"] == 0 and B >= 1 and S >= 2048 and label
