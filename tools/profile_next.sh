#!/bin/bash
# Kernel-time breakdowns that steer the next round: the Vocos / VITS2 family and configs[2] (v3 B=64 bf16)
mkdir -p gpurun_out; R=/root/repo; cd /tmp && export TMPDIR=/tmp
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vits2 -o r -- python $R/bench.py --model vits2_vocos_v1 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_vits2.log 2>&1
timeout 70 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_v3 -o r -- python $R/bench.py --model v3 --batch 64 --decoder-dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_v3.log 2>&1
rm -f $R/gpurun_out/prof_vits2/r_kernel_trace.csv $R/gpurun_out/prof_v3/r_kernel_trace.csv
ls $R/gpurun_out/prof_vits2 $R/gpurun_out/prof_v3
