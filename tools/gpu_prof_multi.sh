#!/bin/bash
# kernel stats of several bench configurations in one box visit:
#   bash tools/gpu_prof_multi.sh "tag1|flags1" "tag2|flags2" ...   -> gpurun_out/prof_<tag>_kernel_stats.csv + a top-N print
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for item in "$@"; do
  tag=${item%%|*}; flags=${item#*|}
  rm -rf /tmp/pc_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$tag -o r -- python $R/bench.py $flags --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$tag.log 2>&1
  cp $(find /tmp/pc_$tag -name "*kernel_stats.csv" | head -1) $R/gpurun_out/prof_${tag}_kernel_stats.csv
  echo "== $tag ($flags)"; tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-200
  python - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/prof_${tag}_kernel_stats.csv")))
tot=sum(int(r['TotalDurationNs']) for r in rows)
for r in rows[:${TOPN:-24}]:
    n=r['Name'].replace('wetts::','').replace('void ','')[:100]
    print(f"{int(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):6.2f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {n}")
print('total', tot/1e6)
PY
done
