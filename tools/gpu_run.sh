#!/bin/bash
mkdir -p gpurun_out
python tools/bench_conv.py 0 > gpurun_out/conv_fast2.txt 2>&1; cut -c1-100 gpurun_out/conv_fast2.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "v1_b4x128 or v3_b3x128 or fused_resblock or v1_b2 or tiny_sdp_b3" 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c.json 2>gpurun_out/bench_c.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_c.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches'], d['roofline']['avg_launch_ms'], d['roofline']['mrf_share_of_step'])
PY
