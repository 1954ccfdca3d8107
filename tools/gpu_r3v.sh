#!/bin/bash
# round 3, pass V: whole GPU suite with the 8-wave 1x1 small-launch convs as the default; B = 1 stream artefacts
mkdir -p gpurun_out
timeout 345 python -m pytest tests -m gpu -q --timeout 300 -rf 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
timeout 60 python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null
timeout 60 python bench.py --stream --model v1 --decoder-dtype bf16 > gpurun_out/stream_v1_bf16.json 2>/dev/null
for f in stream_v1 stream_v1_bf16; do python -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_graph'],2))"; done
