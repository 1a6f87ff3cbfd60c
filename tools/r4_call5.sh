#!/bin/bash
# round 4, GPU call 5: clean kernel trace of the canonical command, odd-dimension cases on hardware, small-batch configs after the tile rule,
# ragged bench, the example training script end to end
export TMPDIR=/tmp
O=gpurun_out/r4c5; mkdir -p $O
timeout 900 python -m pytest tests/test_configs.py tests/test_engine_golden.py tests/test_bench_launch.py -m gpu -x -q -n 2 > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/gputests.log
rm -rf /tmp/prof; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds > $O/prof_bench.log 2>&1; echo "rocprof rc=$?"
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/r4_bench_kernel_stats.csv
python tools/summarize_stats.py /tmp/prof > $O/r4_bench_kernel_stats_summary.txt 2>&1
python tools/trace_gaps.py /tmp/prof > $O/r4_trace_gaps.txt 2>&1
for spec in "qwen3-30b-a3b-eagle31 1 4096" "deepseek-v3 1 2048"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$1_b$2_s$3.json 2> $O/bench_$1_b$2_s$3.err; echo "bench $spec rc=$?"
done
timeout 600 python tools/ragged_bench.py > $O/ragged_bench.jsonl 2> $O/ragged_bench.err; echo "ragged rc=$?"
timeout 600 python examples/train_offline_eagle3.py --draft-config examples/tiny_draft_config.json --synthetic 48 --batch-size 2 --max-len 64 --ttt-length 3 > $O/example.log 2>&1; echo "example rc=$?"; tail -n 3 $O/example.log
