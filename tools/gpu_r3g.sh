#!/bin/bash
# round 3, pass G: B = 1 latency -- fused short-sequence attention kernel, forked streams for small stages
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -rf -k "reference_golden or text_encoder or v1_b16x128 or random_small or ragged_and_degenerate or session or native or stream" 2>&1 | tail -30 > gpurun_out/pytest_gpu_g.log
tail -4 gpurun_out/pytest_gpu_g.log
for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128" "WETTS_ATTN_SMALL=128 WETTS_TUNE=small_fork=1" "WETTS_ATTN_SMALL=128 WETTS_TUNE=small_fork=1,conv_groups=0" "WETTS_ATTN_SMALL=0"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'mid', round(d['middle_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'total', round(d['stream_total_ms_plain'],2), 'graph win', round(d['first_window_ms_graph'],3))"; done
for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128"; do env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline [$e] ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4))"; done
env WETTS_ATTN_SMALL=128 WETTS_TUNE=small_fork=1 python bench.py --stream --model vits2_vocos_v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream vits2_vocos enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3))"
