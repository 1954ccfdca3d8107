#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rf -k "uint8 or dynamic_quant" 2>&1 | tail -8
python bench.py --decoder-dtype uint8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('uint8 ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'share', round(r['mrf_share_of_step'],3))"
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_stats_uint8 -o r -- python /root/repo/bench.py --decoder-dtype uint8 --steps 2 --warmup 1 --presteps-s 0.5 --no-cpu-baseline > /root/repo/gpurun_out/prof_stats_uint8.log 2>&1)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_stats_uint8/r_kernel_stats.csv')))
for r in rows[:4]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
import collections
tr=list(csv.DictReader(open('gpurun_out/prof_stats_uint8/r_kernel_trace.csv')))
agg=collections.defaultdict(list)
for r in tr:
    if 'qconv_i8' in r['Kernel_Name']:
        agg[(r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X','?'))].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:8]: print('qconv grid', k, 'n', len(v), 'avg_us', round(sum(v)/len(v),1), 'total_ms', round(sum(v)/1e3,1))
PY
