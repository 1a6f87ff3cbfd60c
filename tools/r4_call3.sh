#!/bin/bash
# round 4, GPU call 3: pointwise ablation (N5), under-filled GEMM tile A/B, headline + small-batch benches after the diag / rs-split changes
export TMPDIR=/tmp
O=gpurun_out/r4c3
mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_engine_golden.py tests/test_kernels.py -m gpu -x -q -n 2 > $O/gputests_attn.log 2>&1; echo "gpu attn/kernels tests rc=$?"; tail -n 3 $O/gputests_attn.log
timeout 600 python tools/ablate_pointwise.py --steps 8 --rounds 3 > $O/ablate_pointwise.json 2> $O/ablate_pointwise.err; echo "ablate rc=$?"
timeout 300 python tools/small_tile_ab.py > $O/small_tile_product.jsonl 2> $O/small_tile_product.err; echo "small tile product rc=$?"
SF_GEMM_TILE=128 timeout 300 python tools/small_tile_ab.py > $O/small_tile_128.jsonl 2> $O/small_tile_128.err; echo "small tile 128 rc=$?"
SF_GEMM_TILE=256 timeout 300 python tools/small_tile_ab.py > $O/small_tile_256tools.jsonl 2> $O/small_tile_256tools.err; echo "small tile 256(tools) rc=$?"
timeout 600 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_headline.json 2> $O/bench_headline.err; echo "bench rc=$?"
for spec in "qwen3-next-80b-a3b 8 2048" "qwen3-next-80b-a3b 1 4096" "qwen3-30b-a3b-eagle31 1 4096"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$1_b$2_s$3.json 2> $O/bench_$1_b$2_s$3.err; echo "bench $spec rc=$?"
done
