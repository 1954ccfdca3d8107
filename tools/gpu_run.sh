#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "flow_16bit or v3_b3x128 or tiny_sdp_b3" 2>&1 | tail -12
for fd in f32 bf16; do python bench.py --config multilingual --flow-dtype $fd --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg2_flow_$fd.json 2>gpurun_out/bench_cfg2.err; python - <<PY
import json; d=json.load(open('gpurun_out/bench_cfg2_flow_$fd.json')); print("$fd", d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['mrf_share_of_step'], d['dtype'])
PY
done
