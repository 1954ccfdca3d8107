#!/bin/bash
# round 5, call 1: the four goldens that had not met the GPU (v2 at 8 / 128 phonemes, vits2_v1, stress48k f32), the
# parity file with the interior staging on, then the A/B of WETTS_CONV_INTERIOR on the MRF single convs and the headline
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -rf -k "v2_b2 or v2_b4x128 or vits2_v1_b2 or stress48k_b2" 2>&1 | grep -v "^\[wetts" | tail -25 | tee gpurun_out/r5a_new_goldens.log
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -rf 2>&1 | grep -v "^\[wetts" | tail -8 | tee gpurun_out/r5a_parity.log
for i in 0 1; do
  echo "== WETTS_CONV_INTERIOR=$i" | tee -a gpurun_out/r5a_interior_ab.txt
  WETTS_CONV_INTERIOR=$i WETTS_BENCH_ITERS=10 python tools/bench_conv.py 0 2>&1 | tee -a gpurun_out/r5a_interior_ab.txt
done
for i in 0 1 0 1; do
  WETTS_CONV_INTERIOR=$i python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5a_bench_interior$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/r5a_bench_interior$i.json')); r=d['roofline']; print('interior=$i headline', round(d['value']/1e6,2), 'M/s frac', round(r['frac'],4), 'iso', r.get('isolated',{}).get('frac'), 'ms', round(d['ms_per_step'],2))" | tee -a gpurun_out/r5a_interior_ab.txt
done
