#!/bin/bash
# kernel stats of one bench configuration: CFG="--config multilingual" bash tools/gpu_prof_cfg.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o r -- python $R/bench.py $CFG --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_cfg.log 2>&1
cp $(find /tmp/pc -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_cfg_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_cfg_kernel_stats.csv")))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    n=r['Name'].replace('wetts::','').replace('void ','')[:80]
    print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {n}")
print('total', tot/1e6)
PY
