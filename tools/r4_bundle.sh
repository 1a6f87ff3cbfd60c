#!/bin/bash
# round 4: measurement bundle of a commit (usage: bash tools/r4_bundle.sh <commit>): full GPU suite, bench line (feeds, dense mask, CPU baseline),
# rocprofv3 kernel stats of the same command, PMC passes, per-shape GEMM table, attention A/B, smoke
export TMPDIR=/tmp
C=$1
O=gpurun_out/final; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -n 2 > $O/r4_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/r4_gputests.log
bash tools/final_profile.sh r4; echo "profile rc=$?"
bash tools/final_pmc.sh r4 $C; echo "pmc rc=$?"
tail -n 2 $O/r4_bench_line.json | cut -c1-600
