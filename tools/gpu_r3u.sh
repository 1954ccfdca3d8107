#!/bin/bash
# round 3, pass U: 1x1 convs of small launches as one-chunk stages over 4 / 8 / 16 waves (WETTS_SMALL_1X1)
mkdir -p gpurun_out
for e in 0 1 2 3 0 3; do WETTS_SMALL_1X1=$e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [WETTS_SMALL_1X1=$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_plain'],2))"; done 2>&1 | tee gpurun_out/small_1x1_ab.txt
