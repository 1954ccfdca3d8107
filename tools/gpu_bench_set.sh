#!/bin/bash
# a set of bench lines in one box visit:  bash tools/gpu_bench_set.sh "tag|flags" ...  -> gpurun_out/bs_<tag>.json + one summary line each
mkdir -p gpurun_out
for item in "$@"; do
  tag=${item%%|*}; flags=${item#*|}
  python bench.py $flags --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > gpurun_out/bs_$tag.json 2>gpurun_out/bs_$tag.err
  python - <<PY
import json
try:
    d=json.load(open('gpurun_out/bs_$tag.json')); r=d.get('roofline',{})
    print('$tag', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms  frac', round(r.get('frac') or 0,4), 'mrf_share', round(r.get('mrf_share_of_step',0),3), 'launch_ms', round(r.get('avg_launch_ms',0) or 0,4))
except Exception as e:
    print('$tag FAILED', e); print(open('gpurun_out/bs_$tag.err').read()[-1500:])
PY
done
