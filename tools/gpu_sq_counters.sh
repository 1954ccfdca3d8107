#!/bin/bash
# SQ counters of a kernel class: MFMA pipe busy share and where the waves wait.  Two passes (counter slots), kernel-trace only
# beside --pmc.  usage: gpu_sq_counters.sh [tag [filter [bench flags...]]]
#   tag     names the outputs (default mrf)                      -> gpurun_out/sq_counters_<tag>.txt
#   filter  tools/pmc_summary.py's class filter: `mrf` (the f32 MRF class, default) or `re:<regex on the kernel name>`
#   flags   bench.py flags (default --decoder-serial: the one-stream schedule, so that dispatches do not overlap)
mkdir -p gpurun_out
R=$(pwd)
TAG=${1:-mrf}; FILTER=${2:-mrf}
if [ $# -ge 3 ]; then shift 2; FLAGS="$@"; else FLAGS="--decoder-serial"; fi
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_a_$TAG -o r -- python $R/bench.py $FLAGS --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/pmc_sq_a_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_b_$TAG -o r -- python $R/bench.py $FLAGS --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/pmc_sq_b_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_c_$TAG -o r -- python $R/bench.py $FLAGS --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/pmc_sq_c_$TAG.log 2>&1
cd $R
find gpurun_out/pmc_sq_a_$TAG gpurun_out/pmc_sq_b_$TAG gpurun_out/pmc_sq_c_$TAG -name "*kernel_trace.csv" -delete 2>/dev/null
{ echo "== bench flags: $FLAGS ; class filter: $FILTER"; echo "== pass a: MFMA busy"; python tools/pmc_summary.py gpurun_out/pmc_sq_a_$TAG "$FILTER"; echo "== pass b: wave cycles / waits"; python tools/pmc_summary.py gpurun_out/pmc_sq_b_$TAG "$FILTER"; echo "== pass c: instruction mix"; python tools/pmc_summary.py gpurun_out/pmc_sq_c_$TAG "$FILTER"; } > gpurun_out/sq_counters_$TAG.txt 2>&1
tail -3 gpurun_out/pmc_sq_a_$TAG.log; head -40 gpurun_out/sq_counters_$TAG.txt
