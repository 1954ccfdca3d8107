#!/bin/bash
# round 3, pass L: pair kernel residency / clock per shape
mkdir -p gpurun_out; : > gpurun_out/pair16_phase_clock2.txt
for sh in 128:11 128:7 64:11 64:3; do
WETTS_PAIR16_PROF=1 WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=$sh timeout 200 python tools/bench_conv.py 16 2>&1 | grep -v amdgpu.ids | grep "d=1\|prof" | tail -3 | tee -a gpurun_out/pair16_phase_clock2.txt
done
