#!/bin/bash
# round 5: the ONE measurement bundle of the round (usage: bash tools/r5_bundle.sh <commit>), taken last: full GPU suite (serial, the driver's command line), the default bench line
# (feeds, dense mask, sparse mask, configs legs, CPU baseline), rocprofv3 kernel stats of the same command, PMC passes (one counter group per
# run), per-shape GEMM table, attention A/B at head_dim 128 and 256, smoke.  Every command under its own `timeout`.
export TMPDIR=/tmp
C=${1:-unknown}
O=gpurun_out/final; mkdir -p $O
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/r5_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/r5_gputests.log      # serial, as the driver runs it
bash tools/final_profile.sh r5; echo "profile rc=$?"
timeout 150 python tools/attn_bench.py --shape=8,2048,16,2,256 base SF_ATTN_W1=0 > $O/r5_attn_bench_hd256_8x2048.jsonl 2>/dev/null
timeout 150 python tools/attn_bench.py --shape=1,4096,16,2,256 base SF_ATTN_W1=0 > $O/r5_attn_bench_hd256_1x4096.jsonl 2>/dev/null
timeout 200 python bench.py --config qwen3-next-80b-a3b --steps 6 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds > $O/r5_bench_qwen3-next-80b-a3b_b8_s2048.json 2>/dev/null
timeout 200 python bench.py --config qwen3-next-80b-a3b --batch 1 --seq 4096 --steps 6 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds > $O/r5_bench_qwen3-next-80b-a3b_b1_s4096.json 2>/dev/null
bash tools/final_pmc.sh r5 $C; echo "pmc rc=$?"
# randomised sweeps + soak on the same commit (short forms; the long sweeps of the round: profiles/r5_fuzz_summary.jsonl, r5_soak.jsonl)
timeout 200 python tools/engine_fuzz.py --cases 150 --seed 7 > $O/r5_fuzz_engine_final.jsonl 2>/dev/null; echo "engine fuzz rc=$?"
timeout 100 python tools/kernel_fuzz.py --cases 300 --seed 7 > $O/r5_fuzz_kernel_final.jsonl 2>/dev/null; echo "kernel fuzz rc=$?"
timeout 150 python tools/soak.py --steps 400 --seed 7 > $O/r5_soak_final.jsonl 2>/dev/null; echo "soak rc=$?"
tail -n 1 $O/r5_fuzz_engine_final.jsonl $O/r5_fuzz_kernel_final.jsonl $O/r5_soak_final.jsonl | cut -c1-300
tail -n 1 $O/r5_bench_line.json | cut -c1-400
