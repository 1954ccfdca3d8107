#!/bin/bash
# round 3, pass D: 16-bit pairs with pipelined B reads + mul/max leaky-relu, MAS without the band select, full suite
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -60 > gpurun_out/pytest_gpu_d.log
tail -4 gpurun_out/pytest_gpu_d.log
echo "--- pairs, MB=1 pipelined (two launches | fused)"; WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=128:3,128:7,128:11,64:3,64:7,64:11,32:3,32:7,32:11 python tools/bench_conv.py 32,16 2>/dev/null | tee gpurun_out/pair16_pipelined.txt
for cfgs in "--decoder-dtype bf16" "--config stress48k" "--config multilingual"; do
python bench.py $cfgs --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>gpurun_out/tmp.err; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('$cfgs ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'mfma', round(r.get('mfma_view',{}).get('frac',0),3), 'share', round(r['mrf_share_of_step'],3))" || tail -3 gpurun_out/tmp.err; done
python bench.py --mas > gpurun_out/mas.json 2> gpurun_out/mas.err; python -c "
import json; d=json.load(open('gpurun_out/mas.json'))
for c in d['cases']: print('mas', c['shape'], 'device ms', round(c['device_ms'],4), 'cpu ms', round(c['cpu_ms_1thread'],3), 'x', round(c['speedup'],1), c['bit_exact_vs_c_oracle'])"
WETTS_TUNE=mrf_streams=3 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline mrf_streams=3 ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4))"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4))"
