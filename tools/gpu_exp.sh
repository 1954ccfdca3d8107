#!/bin/bash
# experiment pass: per-stage three-stream fork of the ResBlock chains (WETTS_TUNE=mrf_fork_maxc=C) beside the grouped launches
mkdir -p gpurun_out
rm -f gpurun_out/exp_fork.txt
WETTS_TUNE=mrf_fork_maxc=64 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 500 -k "v1_b2 or v1_b4x128 or aishell3_b4x128 or v2_b4x128 or stress48k" 2>&1 | grep -v "^\[wetts" | tail -3 | tee -a gpurun_out/exp_fork.txt
WETTS_TUNE=mrf_fork_maxc=64 timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --timeout 500 -k "v1_b16x128 or ragged" 2>&1 | grep -v "^\[wetts" | tail -3 | tee -a gpurun_out/exp_fork.txt
run() {  # $1 = label, rest = env assignments
  lab=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/exp_b_$lab.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/exp_b_$lab.json')); r=d['roofline']; print('$lab', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'iso', round(r.get('isolated',{}).get('frac',0),4), 'ms', round(d['ms_per_step'],2), 'avg_launch', round(r['avg_launch_ms'],4), 'launches', r['launches'])" | tee -a gpurun_out/exp_fork.txt
}
run base A=1
run fork32 WETTS_TUNE=mrf_fork_maxc=32
run fork64 WETTS_TUNE=mrf_fork_maxc=64
run fork128 WETTS_TUNE=mrf_fork_maxc=128
run fork256 WETTS_TUNE=mrf_fork_maxc=256
run base2 A=1
run fork64b WETTS_TUNE=mrf_fork_maxc=64
run fork32b WETTS_TUNE=mrf_fork_maxc=32
