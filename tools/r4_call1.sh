#!/bin/bash
# round 4, GPU call 1: new tests + feed table + per-config bench lines
export TMPDIR=/tmp
O=gpurun_out/r4c1
mkdir -p $O
timeout 900 python -m pytest tests/test_bench_launch.py tests/test_ingest.py -m gpu -x -q -s > $O/tests_launch.log 2>&1; echo "launch tests rc=$?"
timeout 1200 python -m pytest tests/test_full_size.py -m gpu -q -s -k "three_optimizer" > $O/tests_opt3.log 2>&1; echo "opt3 rc=$?"
timeout 1200 python -m pytest tests/test_kernels_fullsize.py -m gpu -q -s -k "ttt_attention_long or teacher_headline" > $O/tests_attn4096.log 2>&1; echo "attn4096 rc=$?"
timeout 1200 python -m pytest tests/test_configs.py -m gpu -q -s -k "real_dims" > $O/tests_realdims.log 2>&1; echo "realdims rc=$?"
cp gpurun_out/parity_*.json $O/ 2>/dev/null
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"
for spec in "qwen3-8b 8 2048" "qwen3-30b-a3b-eagle31 1 4096" "qwen3-30b-a3b-eagle31 4 4096" "deepseek-v3 1 2048" "deepseek-v3 4 2048" "deepseek-v3 8 2048"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$1_b$2_s$3.json 2> $O/bench_$1_b$2_s$3.err; echo "bench $spec rc=$?"
done
tail -3 $O/tests_*.log
