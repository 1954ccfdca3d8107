#!/bin/bash
# round 3, pass M: power / clock probe per workload; short-sequence attention kernel at batch 64
mkdir -p gpurun_out
timeout 600 python tools/power_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/power_probe.txt
for cfg in "--config multilingual" ""; do for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128"; do env $e python bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('[$cfg] [$e] ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4))"; done; done
