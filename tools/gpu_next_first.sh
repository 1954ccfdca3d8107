#!/bin/bash
# What the first GPU call after round 4 should run (the round ended with its GPU minutes spent):
#  1. the two reference goldens that have only met the oracle so far (v2.json: 64 / 32 / 16 / 8-channel ResBlock stages;
#     vits2_v1.json) through the GPU parity test -- then move them from util.ORACLE_ONLY_CASES to util.INFER_CASES;
#  2. the full GPU suite on the final round-4 tree (the last change, four A-fragment register sets in
#     resblock_chain32_kernel, was checked by the f32 bit-identity tests and two full-size goldens only);
#  3. the headline line with the CPU baseline, for profiles/.
# The unmeasured staging change waits on the branch wip/interior-staging: check it out, build, and A/B it with
# tools/bench_conv.py 0 (single MRF convs) and bench.py before merging.
mkdir -p gpurun_out
WETTS_EXTRA_CASES=1 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 300 -k "v2_b2 or vits2_v1_b2" 2>&1 | grep -v "^\[wetts" | tail -5
bash tools/gpu_tests.sh
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-300 gpurun_out/bench.json
