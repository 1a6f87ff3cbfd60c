#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c8; mkdir -p $O
timeout 900 python -m pytest tests/test_attention.py tests/test_kernels_fullsize.py -m gpu -x -q -n 2 -k "attention or ttt" > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/gputests.log
timeout 200 python tools/attn_bench.py base > $O/attn_bench.jsonl 2>/dev/null; echo "attn bench rc=$?"
grep '"fwd"' $O/attn_bench.jsonl | cut -c1-160
for i in 1 2; do timeout 300 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$i.json 2> $O/bench_$i.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4c8/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), {k:(round(v['frac'],3), round(v['ms_per_step'],2)) for k,v in d['kernels'].items() if k.startswith('attn')})
PY
