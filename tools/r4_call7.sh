#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r4c7; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels.py tests/test_kernels_fullsize.py tests/test_configs.py -m gpu -x -q -n 2 > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -n 3 $O/gputests.log
timeout 300 python tools/small_tile_ab.py > $O/small_tile_product.jsonl 2> $O/p.err; echo "product rc=$?"
SF_NT_WS=1 timeout 300 python tools/small_tile_ab.py > $O/small_tile_splitk.jsonl 2> $O/s.err; echo "splitk rc=$?"
for spec in "qwen3-30b-a3b-eagle31 1 4096" "qwen3-next-80b-a3b 1 4096" "qwen3-30b-a3b-eagle31 2 4096"; do
  set -- $spec
  timeout 600 python bench.py --config $1 --batch $2 --seq $3 --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_$1_b$2_s$3.json 2> $O/bench_$1_b$2_s$3.err; echo "bench $spec rc=$?"
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-feeds --no-cpu-baseline --no-dense-mask > $O/bench_headline.json 2> $O/bench_headline.err; echo "headline rc=$?"
timeout 300 python bench.py --steps 5 --warmup 2 --no-feeds --no-cpu-baseline --no-dense-mask --force-dp --dist-backend nccl > $O/bench_rccl_world1.json 2> $O/bench_rccl_world1.err; echo "rccl world1 rc=$?"
python - <<'PY'
import json,glob
def load(f): return {(r['M'],r['N'],r['K']):r for r in map(json.loads, open(f))}
p,s=load('gpurun_out/r4c7/small_tile_product.jsonl'),load('gpurun_out/r4c7/small_tile_splitk.jsonl')
for k in p: print(k, p[k]['tiles256'], 'product', p[k]['ms'], p[k]['tflops'], '| ws', s[k]['ms'], s[k]['tflops'])
for f in sorted(glob.glob('gpurun_out/r4c7/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3))
    except Exception as e: print(f,'ERR',e)
PY
