#!/bin/bash
# round 3, pass A: the new parity pins + MAS + batching plan
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python bench.py --mas > gpurun_out/mas.json 2> gpurun_out/mas.err; cat gpurun_out/mas.json | cut -c1-1500
python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-600 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
for nb in 0 4 6 8 12; do python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline --buckets $nb > gpurun_out/aishell3_b$nb.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/aishell3_b$nb.json')); print('aishell3 buckets=$nb ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms frame_pad', round(d['config']['frame_pad_frac_rank0'],3), d['config']['sub_batch_plan']['sizes_rank0'])"; done
python bench.py --config aishell3 --steps 4 --warmup 2 --no-cpu-baseline --max-pad-frac 0.04 > gpurun_out/aishell3_p04.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/aishell3_p04.json')); print('aishell3 pad<=0.04 ->', d['config']['padded_sub_batches_per_step'], 'calls', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],1), 'ms frame_pad', round(d['config']['frame_pad_frac_rank0'],3))"
