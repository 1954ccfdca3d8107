#!/bin/bash
# round 3, pass R: fused DDSConv kernel, second version (parameters in LDS, batched loads)
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -rf -k "dds_fused or reference_golden" 2>&1 | tail -6
for e in "WETTS_TUNE=dds_fused=0" "WETTS_TUNE=dds_fused=1" "WETTS_TUNE=dds_fused=0" "WETTS_TUNE=dds_fused=1"; do env $e python bench.py --stream --model v1 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph', round(d['first_chunk_latency_ms_graph'],3), 'total', round(d['stream_total_ms_plain'],2))"; done 2>&1 | tee gpurun_out/dds_fused_ab.txt
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/b1 -o b1 --output-format csv -- python tools/trace_b1.py --reps 5 > gpurun_out/b1_run.txt 2>&1
python tools/trace_b1.py --summarize /tmp/b1 > gpurun_out/b1_summary.txt 2>&1
grep "dds\|call 5\|call 6" gpurun_out/b1_summary.txt | head
