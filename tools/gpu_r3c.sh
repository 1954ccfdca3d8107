#!/bin/bash
# round 3, pass C: MB=2 16-bit kernels (A/B micro-benchmarks + parity), MAS ring, ragged XCD order
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rf -k "fused_resblock or 16bit or bf16 or f16 or reduced_precision or onnx or aishell3 or mas or ragged or uint8" 2>&1 | tail -120 > gpurun_out/pytest_gpu_c.log
tail -6 gpurun_out/pytest_gpu_c.log
echo "--- pairs, MB=2 (two launches | fused)"; WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=128:3,128:7,128:11,64:3,64:7,64:11 python tools/bench_conv.py 32,16 2>/dev/null | tee gpurun_out/pair16_mb2.txt
echo "--- pairs, MB=1 (round 2)"; WETTS_PAIR16_MB=1 WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=128:3,128:7,128:11,64:3,64:7,64:11 python tools/bench_conv.py 32,16 2>/dev/null | tee gpurun_out/pair16_mb1.txt
echo "--- C=256 singles, conv16_mb2"; WETTS_CONV_FLAGS=16 WETTS_SHAPES=256:3,256:7,256:11 python tools/bench_conv.py 0 2>/dev/null | tee gpurun_out/conv16_mb2.txt
echo "--- C=256 singles, round 2 kernel"; WETTS_CONV16_MB2=0 WETTS_CONV_FLAGS=16 WETTS_SHAPES=256:3,256:7,256:11 python tools/bench_conv.py 0 2>/dev/null | tee gpurun_out/conv16_mb1.txt
for cfgs in "--decoder-dtype bf16" "--config stress48k" "--config multilingual"; do
python bench.py $cfgs --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>gpurun_out/tmp.err; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('$cfgs ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'mfma', round(r.get('mfma_view',{}).get('frac',0),3), 'share', round(r['mrf_share_of_step'],3))" || tail -3 gpurun_out/tmp.err; done
for e in "WETTS_PAIR16_MB=1 WETTS_CONV16_MB2=0"; do env $e python bench.py --decoder-dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('bf16 round-2 kernels ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3))"; done
python bench.py --mas > gpurun_out/mas.json 2> gpurun_out/mas.err; python -c "
import json; d=json.load(open('gpurun_out/mas.json'))
for c in d['cases']: print('mas', c['shape'], 'device ms', round(c['device_ms'],4), 'cpu ms', round(c['cpu_ms_1thread'],3), 'x', round(c['speedup'],1), c['bit_exact_vs_c_oracle'])"
python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/aishell3_ragged.json 2>gpurun_out/aishell3_ragged.err
python -c "
import json; d=json.load(open('gpurun_out/aishell3_ragged.json')); print('aishell3 ragged ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms frac', round(d['roofline']['frac'],3), d['config']['sub_batch_plan']['sizes_rank0'])"
python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline --max-pad-frac 0.3 > gpurun_out/aishell3_ragged1.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/aishell3_ragged1.json')); print('aishell3 ragged pad<=0.3 ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms frac', round(d['roofline']['frac'],3), d['config']['sub_batch_plan']['sizes_rank0'])"
