#!/bin/bash
# Short closing pass after a kernel change: rocprofv3 kernel stats + the two HBM counter passes of the headline line, the
# default bench line (with the CPU baseline) and the B = 1 streaming line.  The full pass is tools/gpu_round.sh.
mkdir -p gpurun_out
R=/root/repo
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_baker -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats_baker.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_baker -o r -- python $R/bench.py --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline > $R/gpurun_out/pmc_fetch_baker.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_baker -o r -- python $R/bench.py --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline > $R/gpurun_out/pmc_write_baker.log 2>&1
find $R/gpurun_out/prof_stats_baker $R/gpurun_out/pmc_fetch_baker $R/gpurun_out/pmc_write_baker -name "*kernel_trace.csv" -delete 2>/dev/null
cd $R
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-300 gpurun_out/bench.json
timeout 120 python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null
cut -c1-400 gpurun_out/stream_v1.json
du -sh gpurun_out
