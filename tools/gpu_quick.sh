#!/bin/bash
# quick regression look: tests, headline, configs[2], streaming
mkdir -p gpurun_out
if [ -n "$Q_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -8; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_bench.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/q_bench.json')); r=d['roofline']; print('headline', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'mrf_share', round(r.get('mrf_share_of_step',0),3), 'ms', round(d['ms_per_step'],2))"
for cfg in multilingual stress48k; do python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/q_$cfg.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/q_$cfg.json')); r=d['roofline']; print('$cfg', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'mrf_share', round(r.get('mrf_share_of_step',0),3), 'ms', round(d['ms_per_step'],2))"; done
python bench.py --stream --model v1 > gpurun_out/q_stream_v1.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/q_stream_v1.json')); print('stream v1 enc', round(d['encoder_ms'],2), 'win', round(d['first_window_ms_plain'],2), 'first chunk', round(d['first_chunk_latency_ms_plain'],2))"
