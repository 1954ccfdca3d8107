#!/bin/bash
# quick regression look: headline, configs[2], streaming
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_bench.json 2>/dev/null; cut -c1-330 gpurun_out/q_bench.json; echo
python -c "
import json; d=json.load(open('gpurun_out/q_bench.json')); r=d['roofline']; print('frac', r['frac'], 'mrf_share', r.get('mrf_share_of_step'), 'ms', d['ms_per_step'])"
python bench.py --config multilingual --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_cfg2.json 2>/dev/null; cut -c1-200 gpurun_out/q_cfg2.json; echo
python bench.py --stream --model vits2_vocos_v1 > gpurun_out/q_stream_vits2.json 2>/dev/null; cut -c1-900 gpurun_out/q_stream_vits2.json; echo
python bench.py --stream --model v1 --stream-phonemes 16 > gpurun_out/q_stream_v1_16.json 2>/dev/null; cut -c1-900 gpurun_out/q_stream_v1_16.json; echo
