#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_b.json 2>gpurun_out/bench_b.err; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_b.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches'], d['roofline']['avg_launch_ms'], d['roofline']['mrf_share_of_step'])
PY
python bench.py --config aishell3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg3_aishell3.json 2>/dev/null; python - <<'PY'
import json; d=json.load(open('gpurun_out/bench_cfg3_aishell3.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['workload'])
PY
WETTS_TUNE="bogus=1" python bench.py --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
