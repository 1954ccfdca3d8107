#!/bin/bash
# the GPU suite + the headline line twice (a quick regression look between the full passes of tools/gpu_round.sh)
mkdir -p gpurun_out
bash tools/gpu_tests.sh
for i in 1 2; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/check_bench$i.json 2>gpurun_out/check_bench$i.err
  python -c "
import json; d=json.load(open('gpurun_out/check_bench$i.json')); r=d['roofline']; print('headline', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'iso', round(r.get('isolated',{}).get('frac',0),4), 'ms', round(d['ms_per_step'],2), 'hbm_view', round(r['hbm_view']['frac'],4))"
done
