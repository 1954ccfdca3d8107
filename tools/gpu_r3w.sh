#!/bin/bash
# round 3, pass W: wave count of the small-launch convs with taps (k = 3 / 5: 12 stages) -- 4 waves today
mkdir -p gpurun_out
for e in "WETTS_SMALL_KG8_NS=16" "WETTS_SMALL_KG8_NS=8" "WETTS_SMALL_KG8_NS=16" "WETTS_SMALL_KG8_NS=8 WETTS_SMALL_KG16_NS=12"; do env $e timeout 50 python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_plain'],2))"; done 2>&1 | tee gpurun_out/small_kg_ab.txt
