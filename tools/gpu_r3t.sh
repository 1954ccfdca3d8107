#!/bin/bash
# round 3, pass T: WaveNet residual / skip update fused into the res_skip conv's epilogue (f32 flow)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rf -k "wn_update or reference_golden or flow or vits2 or preconv" 2>&1 | tail -6
for e in "WETTS_TUNE=wn_fuse=0" "WETTS_TUNE=wn_fuse=1" "WETTS_TUNE=wn_fuse=0" "WETTS_TUNE=wn_fuse=1"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_plain'],2))"; done 2>&1 | tee gpurun_out/wn_fuse_ab.txt
for e in "WETTS_TUNE=wn_fuse=0" "WETTS_TUNE=wn_fuse=1" "WETTS_TUNE=wn_fuse=0" "WETTS_TUNE=wn_fuse=1"; do env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline [$e] ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4))"; done 2>&1 | tee -a gpurun_out/wn_fuse_ab.txt
