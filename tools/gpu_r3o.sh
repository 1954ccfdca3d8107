#!/bin/bash
# round 3, pass O: 16-bit decoder with the ResBlock chains of a stage on three streams
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q --timeout 900 -rf -k "f16 or bf16 or 16bit or reduced_precision or bit_identical or stream" 2>&1 | tail -12 > gpurun_out/pytest_gpu_o.log
tail -4 gpurun_out/pytest_gpu_o.log
for cfg in "--decoder-dtype bf16" "--config stress48k" "--decoder-dtype f16"; do for e in "WETTS_TUNE=mrf_streams16=1" "WETTS_TUNE=mrf_streams16=3" "WETTS_TUNE=mrf_streams16=1" "WETTS_TUNE=mrf_streams16=3"; do env $e python bench.py $cfg --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); r=d['roofline']; print('[$cfg] [$e] ->', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],3), 'ms frac', round(r['frac'],4), 'share', round(r['mrf_share_of_step'],3))"; done; done
for e in "WETTS_TUNE=mrf_streams16=1" "WETTS_TUNE=mrf_streams16=3"; do env $e python bench.py --stream --model v1 --decoder-dtype bf16 > gpurun_out/tmp.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/tmp.json')); print('stream v1 bf16 [$e] enc', round(d['encoder_ms'],3), 'win', round(d['first_window_ms_plain'],3), 'first chunk', round(d['first_chunk_latency_ms_plain'],3), 'graph win', round(d['first_window_ms_graph'],3))"; done
