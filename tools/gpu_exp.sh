#!/bin/bash
# experiment pass: narrow chain tiles (WETTS_CHAIN_NB3 = 0 | 3 | 2) on the micro-benchmark and the headline; tile shapes of
# the text encoder's FFN convs at B = 64 (configs[2])
mkdir -p gpurun_out
for nb in 0 3 2; do
  echo "== WETTS_CHAIN_NB3=$nb" | tee -a gpurun_out/exp_chain_nb.txt
  WETTS_CHAIN_NB3=$nb python tools/bench_resblock.py 32:3,32:7,32:11,64:3,64:7 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/exp_chain_nb.txt
done
for nb in 0 3 2 0 3; do
  WETTS_CHAIN_NB3=$nb python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/exp_bench_nb$nb.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/exp_bench_nb$nb.json')); r=d['roofline']; print('chain_nb3=$nb headline', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'iso', round(r.get('isolated',{}).get('frac',0),4), 'ms', round(d['ms_per_step'],2))" | tee -a gpurun_out/exp_chain_nb.txt
done
WETTS_BENCH_B=64 WETTS_XSHAPES=192:768:3:128,768:192:3:128,192:384:5:768 python tools/bench_conv.py 0,2,3,4,5,6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/exp_ffn_tiles_b64.txt
