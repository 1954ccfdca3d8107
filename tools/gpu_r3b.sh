#!/bin/bash
# round 3, pass B: full GPU suite, MAS, ragged decode on configs[3]
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q --timeout 900 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
tail -12 gpurun_out/pytest_gpu.log
python bench.py --mas > gpurun_out/mas.json 2> gpurun_out/mas.err; python -c "
import json; d=json.load(open('gpurun_out/mas.json'))
for c in d['cases']: print('mas', c['shape'], 'device ms', round(c['device_ms'],4), 'cpu ms', round(c['cpu_ms_1thread'],3), 'x', round(c['speedup'],1), c['bit_exact_vs_c_oracle'])"
for dec in ragged padded; do python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline --decode $dec > gpurun_out/aishell3_$dec.json 2>gpurun_out/aishell3_$dec.err
python -c "
import json; d=json.load(open('gpurun_out/aishell3_$dec.json')); print('aishell3 $dec ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms frac', round(d['roofline']['frac'],3), d['config']['sub_batch_plan']['sizes_rank0'])" || tail -3 gpurun_out/aishell3_$dec.err; done
python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline --decode ragged --max-batch 32 > gpurun_out/aishell3_ragged_mb32.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/aishell3_ragged_mb32.json')); print('aishell3 ragged max-batch 32 ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms')"
