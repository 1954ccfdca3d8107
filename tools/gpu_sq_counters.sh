#!/bin/bash
# SQ counters of the f32 MRF class (one-stream schedule, so that dispatches do not overlap): MFMA pipe busy share and where
# the waves wait.  Two passes (counter slots), kernel-trace only beside --pmc.  -> gpurun_out/sq_counters_mrf.txt
mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_a -o r -- python $R/bench.py --decoder-serial --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline > $R/gpurun_out/pmc_sq_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_b -o r -- python $R/bench.py --decoder-serial --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline > $R/gpurun_out/pmc_sq_b.log 2>&1
cd $R
find gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b -name "*kernel_trace.csv" -delete 2>/dev/null
{ echo "== pass a: MFMA busy"; python tools/pmc_summary.py gpurun_out/pmc_sq_a mrf; echo "== pass b: wave cycles / waits"; python tools/pmc_summary.py gpurun_out/pmc_sq_b mrf; } > gpurun_out/sq_counters_mrf.txt 2>&1
tail -5 gpurun_out/pmc_sq_a.log; head -60 gpurun_out/sq_counters_mrf.txt
