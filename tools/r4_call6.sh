#!/bin/bash
# round 4, GPU call 6: side-stream column sums A/B (alternating, same box) + the GPU tests that touch them
export TMPDIR=/tmp
O=gpurun_out/r4c6; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels.py tests/test_engine_golden.py tests/test_full_size.py -m gpu -x -q -n 2 > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/gputests.log
for i in 1 2; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_side_$i.json 2> $O/bench_side_$i.err; echo "side $i rc=$?"
  timeout 300 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask --inline-colsum > $O/bench_inline_$i.json 2> $O/bench_inline_$i.err; echo "inline $i rc=$?"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c6/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), d['final_loss'])
    except Exception as e: print(f, 'ERR', e)
PY
