# Round-end measurement bundle, part 1 (GPU box): bench line, kernel stats, per-shape GEMM table, attention A/B, smoke.
# Every command runs under its own `timeout`: a wedged profiler must not eat the GPU budget.   usage: bash tools/final_profile.sh [rN]
export TMPDIR=/tmp
R=${1:-r5}; O=gpurun_out/final; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 3 > $O/${R}_bench_stdout.log 2>$O/${R}_bench_stderr.log; tail -1 $O/${R}_bench_stdout.log > $O/${R}_bench_line.json
rm -rf /tmp/prof; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-mask --no-feeds --no-configs > $O/prof_bench.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/${R}_bench_kernel_stats.csv
python tools/summarize_stats.py /tmp/prof > $O/${R}_bench_kernel_stats_summary.txt 2>&1
python tools/trace_gaps.py /tmp/prof > $O/${R}_trace_gaps.txt 2>&1
timeout 120 python tools/attn_bench.py base SF_ATTN_FWD_W4=1 SF_ATTN_DQ_W4=1 > $O/${R}_attn_bench.jsonl 2>/dev/null
timeout 150 python tools/gemm_table.py --rounds 3 > $O/${R}_gemm_table.jsonl 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/${R}_smoke.log 2>&1
