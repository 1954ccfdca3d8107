#!/bin/bash
# round 5, call 2: the split-bf16 decision measurement (conv_split.hip, bench library only), the RCCL start-up test with the
# device-bound process group, and the 16-bit lines with launch-granularity roofline bytes
mkdir -p gpurun_out
WETTS_BENCH_ITERS=10 timeout 600 python tools/bench_conv.py 0,106,109 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5b_split_microbench.txt
timeout 600 python -m pytest tests/test_gpu_rccl.py -m gpu -q -s --timeout 500 2>&1 | grep -v "^\[wetts" | tail -12 | tee gpurun_out/r5b_rccl.log
for spec in "cfg2_multilingual_bf16:--config multilingual" "bf16:--decoder-dtype bf16" "cfg4_stress48k_f16:--config stress48k"; do
  tag=${spec%%:*}; flags=${spec#*:}
  python bench.py $flags --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r5b_bench_$tag.json 2>gpurun_out/r5b_bench_$tag.err
  python -c "
import json; d=json.load(open('gpurun_out/r5b_bench_$tag.json')); r=d['roofline']
print('$tag', round(d['ms_per_step'],2),'ms frac', round(r['frac'],4), 'iso', round(r.get('isolated',{}).get('frac',0),4), 'perconv', round(r['perconv_view']['ratio_to_hbm_peak'],4), 'hbm_real', round(r.get('hbm_real',{}).get('frac',0),4), 'bytes/launch', r['bytes_per_launch'], 'traffic', r['traffic'], 'current', r['traffic_current'])" | tee -a gpurun_out/r5b_bench16.txt
done
