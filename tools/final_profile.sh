# Round-end measurement bundle, part 1 (GPU box): bench line, kernel stats, per-shape GEMM table, row kernels, smoke.
# Every command runs under its own `timeout`: a wedged profiler must not eat the GPU budget.
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout 200 python bench.py --steps 20 --warmup 3 > $O/r2_bench_stdout.log 2>$O/r2_bench_stderr.log; tail -1 $O/r2_bench_stdout.log > $O/r2_bench_line.json
timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense-mask > $O/prof_bench.log 2>&1
cp $(find /tmp/prof -name "*kernel_stats.csv" | head -1) $O/r2_bench_kernel_stats.csv
python tools/summarize_stats.py /tmp/prof > $O/r2_bench_kernel_stats_summary.txt 2>&1
python tools/trace_gaps.py /tmp/prof > $O/r2_trace_gaps.txt 2>&1
timeout 120 python tools/gemm_table.py --rounds 3 > $O/r2_gemm_table.jsonl 2>/dev/null
timeout 60 python tools/pointwise_bench.py > $O/r2_pointwise_bench.jsonl 2>/dev/null
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/r2_smoke.log 2>&1
