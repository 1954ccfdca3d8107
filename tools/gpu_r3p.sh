#!/bin/bash
# round 3, pass P: whole ResBlock1 per launch at 16 bit (resblock1_chain16.hip)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -rf -k "f16 or bf16 or 16bit or reduced_precision or bit_identical" 2>&1 | tail -12 > gpurun_out/pytest_gpu_p.log
tail -6 gpurun_out/pytest_gpu_p.log
for cfg in "--decoder-dtype bf16" "--config stress48k"; do for e in "WETTS_TUNE=chain16_pct=0" "WETTS_TUNE=chain16_pct=15" "WETTS_TUNE=chain16_pct=30" "WETTS_TUNE=chain16_pct=0" "WETTS_TUNE=chain16_pct=15" "WETTS_TUNE=chain16_pct=10"; do env $e python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('[$cfg] [$e] ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), 'share', round(r['mrf_share_of_step'],3), 'launches', r.get('launches_per_step'))"; done; done 2>&1 | tee gpurun_out/chain16_ab.txt
