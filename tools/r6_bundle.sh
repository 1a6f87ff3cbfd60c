#!/bin/bash
# round 6: the ONE measurement bundle of the round (usage: bash tools/r6_bundle.sh <commit>), taken last: full GPU suite (serial, the driver's
# command line), the default bench line under the driver's own arguments (feeds, dense mask, sparse mask, configs legs, CPU baseline,
# roofline.by_kernel / by_config), rocprofv3 kernel stats of the same command, PMC passes (one counter group per run), per-shape GEMM table,
# attention bench, the DP telemetry of the RCCL world-1 path (per-bucket gradient-complete -> all-reduce-finished), smoke, short fuzz / soak.
# Every command under its own `timeout`.
export TMPDIR=/tmp
C=${1:-unknown}
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/r6_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/r6_gputests.log
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r6_bench_stdout.log 2>$O/r6_bench_stderr.log; tail -1 $O/r6_bench_stdout.log > $O/r6_bench_line.json; echo "bench rc=$?"
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds --no-configs > $O/prof_bench.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/r6_bench_kernel_stats.csv
python tools/summarize_stats.py /tmp/prof > $O/r6_bench_kernel_stats_summary.txt 2>&1
python tools/trace_gaps.py /tmp/prof > $O/r6_trace_gaps.txt 2>&1
timeout 120 python tools/attn_bench.py base > $O/r6_attn_bench.jsonl 2>/dev/null
timeout 120 python tools/attn_bench.py --shape=8,2048,16,2,256 base > $O/r6_attn_bench_hd256_8x2048.jsonl 2>/dev/null
timeout 150 python tools/gemm_table.py --rounds 3 > $O/r6_gemm_table.jsonl 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r6_smoke.log 2>&1
bash tools/final_pmc.sh r6 $C; echo "pmc rc=$?"
# DP telemetry on the only RCCL world a 1-GPU box has (world 1, forced collectives), default order and --dp-early-lm-head
P=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --dist-backend nccl --force-dp --steps 6 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds --no-configs 2>/dev/null | tail -1 > $O/r6_bench_rccl_world1_force_dp.json
P=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $P bench.py --gpus 1 --dist-backend nccl --force-dp --dp-early-lm-head --steps 6 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds --no-configs 2>/dev/null | tail -1 > $O/r6_bench_rccl_world1_force_dp_early_lm_head.json
timeout 200 python tools/engine_fuzz.py --cases 150 --seed 11 > $O/r6_fuzz_engine_final.jsonl 2>/dev/null; echo "engine fuzz rc=$?"
timeout 100 python tools/kernel_fuzz.py --cases 300 --seed 11 > $O/r6_fuzz_kernel_final.jsonl 2>/dev/null; echo "kernel fuzz rc=$?"
timeout 150 python tools/soak.py --steps 400 --seed 11 > $O/r6_soak_final.jsonl 2>/dev/null; echo "soak rc=$?"
tail -n 1 $O/r6_fuzz_engine_final.jsonl $O/r6_fuzz_kernel_final.jsonl $O/r6_soak_final.jsonl | cut -c1-300
python - <<'PY'
import json
l = json.load(open("gpurun_out/final/r6_bench_line.json"))
print({k: l[k] for k in ("value", "ms_per_step")}, l["roofline"]["frac"], l["roofline"].get("by_kernel"), l["roofline"].get("by_config"))
for f in ("r6_bench_rccl_world1_force_dp.json", "r6_bench_rccl_world1_force_dp_early_lm_head.json"):
    try:
        r = json.load(open("gpurun_out/final/" + f))
        print(f, r["ms_per_step"], r["rccl"].get("exposed_wait_ms_per_step"), [(round(b["mbytes"]), round(b.get("ready_to_done_ms_in_step", -1), 3), round(b["ms"], 3)) for b in r["rccl"]["buckets"]])
    except Exception as e:
        print(f, "failed", e)
PY
