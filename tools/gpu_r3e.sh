#!/bin/bash
# round 3, pass E: grouped conv launches (f32)
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -rf -k "fused_resblock or reference_golden or v1_b16x128 or ragged or hifigan_standalone or small_launch" 2>&1 | tail -40 > gpurun_out/pytest_gpu_e.log
tail -4 gpurun_out/pytest_gpu_e.log
for t in "conv_groups=1" "conv_groups=0" "conv_groups=1" "conv_groups=0"; do WETTS_TUNE=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline $t ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4), 'launches', r['launches'], 'avg ms', round(r['avg_launch_ms'],3))"; done
WETTS_TUNE=conv_groups=1 python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('aishell3 ragged grouped ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4))"
