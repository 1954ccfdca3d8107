#!/bin/bash
# round 3, pass H: where the uint8 decoder's time goes; B = 1 with the blocked attention kernel
mkdir -p gpurun_out
R=/root/repo
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_uint8 -o r -- python $R/bench.py --decoder-dtype uint8 --steps 2 --warmup 1 --presteps-s 0.5 --no-cpu-baseline > $R/gpurun_out/prof_stats_uint8.log 2>&1)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_stats_uint8/r_kernel_stats.csv')))
for r in rows[:12]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
PY
for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128" "WETTS_ATTN_SMALL=128 WETTS_TUNE=small_fork=0"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'total', round(d['stream_total_ms_plain'],2), 'graph==plain', d['graph_equals_plain'])"; done
timeout 600 python -m pytest tests -m gpu -q --timeout 600 -rf -k "graphed or stream or text_encoder or fused_resblock" 2>&1 | tail -4
