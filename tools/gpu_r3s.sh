#!/bin/bash
# round 3, pass S: whole GPU suite with the fused DDSConv on for small launches; B = 1 artefacts refreshed
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 900 -rf 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null
python bench.py --stream --model v1 --decoder-dtype bf16 > gpurun_out/stream_v1_bf16.json 2>/dev/null
python bench.py --stream --model vits2_vocos_v1 --stream-cpu > gpurun_out/stream_vits2_vocos.json 2>/dev/null
(export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/b1 -o b1 --output-format csv -- python tools/trace_b1.py --reps 5 > gpurun_out/b1_run.txt 2>&1; python tools/trace_b1.py --summarize /tmp/b1 > gpurun_out/b1_summary.txt 2>&1)
for f in stream_v1 stream_v1_bf16 stream_vits2_vocos; do python -c "
import json; d=json.load(open('gpurun_out/$f.json')); print('$f enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_graph'],2))"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('headline ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],2), 'ms frac', round(r['frac'],4))"
grep "dds\|call 5" gpurun_out/b1_summary.txt | head -4
