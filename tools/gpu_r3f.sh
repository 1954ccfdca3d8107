#!/bin/bash
# round 3, pass F: phase timeline of the 16-bit pair kernel; cfg2 knobs
mkdir -p gpurun_out
for sh in 128:11 128:7 128:3 64:7 64:3 32:3; do WETTS_PAIR16_PROF=1 WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=$sh python tools/bench_conv.py 16 2>&1 | grep -v amdgpu.ids | grep "prof\]\|d=1" | tail -2; done | tee gpurun_out/pair16_phase_timeline.txt
for t in "fuse2_waste_pct=15" "fuse2_waste_pct=20" "fuse2_waste_pct=30"; do WETTS_TUNE=$t python bench.py --config multilingual --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('multilingual $t ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'share', round(r['mrf_share_of_step'],3), 'launches', r['launches'])"; done
python bench.py --decoder-dtype uint8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('uint8 ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'share', round(r['mrf_share_of_step'],3), 'launches', r['launches'])"
python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/stream_v1.json')); print('stream v1 enc', round(d['encoder_ms'],2), 'win', round(d['first_window_ms_plain'],2), 'first chunk', round(d['first_chunk_latency_ms_plain'],2))"
