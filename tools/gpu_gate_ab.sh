#!/bin/bash
# A/B of the WaveNet gate in the in_layer conv's epilogue (WETTS_TUNE wn_gate): parity tests, then bench lines both ways.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -q -x --timeout 600 -k "wn_gate or wn_update" 2>&1 | grep -v "^\[wetts" | tail -5
python -m pytest tests/test_gpu_fullsize.py -q -x --timeout 600 -k "flow_at_b64 or v1_b16" 2>&1 | grep -v "^\[wetts" | tail -3
for g in 0 1 0 1; do
  WETTS_TUNE=wn_gate=$g python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline wn_gate=$g', round(d['ms_per_step'],3), 'ms', round(d['value']/1e6,2), 'M')"
done
for g in 0 1; do
  WETTS_TUNE=wn_gate=$g python bench.py --stream --model v1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stream wn_gate=$g', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k})"
done
WETTS_TUNE=wn_gate=0 python bench.py --config multilingual --decoder-dtype f32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 f32 B=64 wn_gate=0', round(d['ms_per_step'],3))"
python bench.py --config multilingual --decoder-dtype f32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 f32 B=64 wn_gate=1', round(d['ms_per_step'],3))"
