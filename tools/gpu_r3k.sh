#!/bin/bash
# round 3, pass K: short-sequence attention with 16-byte LDS reads; bf16 loop calibration (tile shapes, operand
# paths, sustained clock); pair kernel phase timeline with the clock
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -rf -k "reference_golden or text_encoder or v1_b16x128 or random_small or ragged_and_degenerate or session or native or stream" 2>&1 | tail -30 > gpurun_out/pytest_gpu_k.log
tail -4 gpurun_out/pytest_gpu_k.log
for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'total', round(d['stream_total_ms_plain'],2))"; done
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/b1 -o b1 --output-format csv -- python tools/trace_b1.py --reps 5 > gpurun_out/b1_run.txt 2>&1
python tools/trace_b1.py --summarize /tmp/b1 > gpurun_out/b1_summary.txt 2>&1
grep -i "attn\|launches\|span" gpurun_out/b1_summary.txt | head -8
timeout 600 python tools/bench_mfma16_loop.py 272 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mfma16_loop_rs272.txt
timeout 300 python tools/bench_mfma16_loop.py 144 2>&1 | grep -v amdgpu.ids | tee gpurun_out/mfma16_loop_rs144.txt
WETTS_PAIR16_PROF=1 WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=128:11,128:3,64:7,32:3 timeout 300 python tools/bench_conv.py 16 2>&1 | grep -v amdgpu.ids | grep "d=1\|prof" | tee gpurun_out/pair16_phase_clock.txt
