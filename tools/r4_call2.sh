#!/bin/bash
# round 4, GPU call 2: full -m gpu suite + headline bench (blocked diagonal backward) + A/B per-step form + small-batch configs + hd 256
export TMPDIR=/tmp
O=gpurun_out/r4c2
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q -n 2 > $O/gputests.log 2>&1; echo "gpu tests rc=$?"
tail -n 5 $O/gputests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline > $O/bench_blocked.json 2> $O/bench_blocked.err; echo "bench blocked rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --diag-per-step > $O/bench_perstep.json 2> $O/bench_perstep.err; echo "bench per-step rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_blocked2.json 2> $O/bench_blocked2.err; echo "bench blocked2 rc=$?"
for spec in "qwen3-30b-a3b-eagle31 1 4096" "deepseek-v3 1 2048" "qwen3-next-80b-a3b 8 2048" "qwen3-next-80b-a3b 1 4096"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$1_b$2_s$3.json 2> $O/bench_$1_b$2_s$3.err; echo "bench $spec rc=$?"
done
