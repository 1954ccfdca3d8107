#!/bin/bash
# round 3, pass I: uint8 decoder with range bookkeeping + pipelined integer conv; attention staging fix
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rf -k "uint8 or dynamic_quant or text_encoder or reference_golden or random_small" 2>&1 | tail -30 > gpurun_out/pytest_gpu_i.log
tail -5 gpurun_out/pytest_gpu_i.log
python bench.py --decoder-dtype uint8 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('uint8 ->', round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],3), 'share', round(r['mrf_share_of_step'],3), 'launches', r['launches'])"
for e in "WETTS_ATTN_SMALL=0" "WETTS_ATTN_SMALL=128"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'total', round(d['stream_total_ms_plain'],2))"; done
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/prof_stats_uint8 -o r -- python /root/repo/bench.py --decoder-dtype uint8 --steps 2 --warmup 1 --presteps-s 0.5 --no-cpu-baseline > /root/repo/gpurun_out/prof_stats_uint8.log 2>&1)
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_stats_uint8/r_kernel_stats.csv')))
for r in rows[:6]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} pct {float(r['Percentage']):5.2f}")
PY
