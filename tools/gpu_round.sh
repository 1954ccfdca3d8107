#!/bin/bash
# One GPU-box pass: parity tests, bench, rocprofv3 kernel stats + HBM counters. Outputs -> gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
ls -R $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch | head -20
du -sh $R/gpurun_out
