#!/bin/bash
# Round-2 GPU pass A: parity tests, smoke, headline bench, N>1 control flow, other configs.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -x 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
tail -8 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-600 gpurun_out/bench.json; tail -2 gpurun_out/bench.err
# N>1 under the driver's command: must refuse on a 1-GPU box (exit 3) ...
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_gpus2_refused.json 2> gpurun_out/bench_gpus2_refused.err; echo "gpus2 exit=$?" | tee -a gpurun_out/bench_gpus2_refused.err
# ... and the same spawn code with both ranks sharing the GPU (dry run, RCCL can't share a device => gloo)
WETTS_BENCH_SINGLE_DEVICE=1 WETTS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_2rank_dryrun.json 2> gpurun_out/bench_2rank_dryrun.err; echo "dryrun exit=$?"
cut -c1-300 gpurun_out/bench_2rank_dryrun.json
python bench.py --config multilingual --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg2_multilingual_bf16.json 2>/dev/null
python bench.py --config aishell3 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cfg3_aishell3.json 2>/dev/null
python bench.py --config stress48k --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg4_stress48k_f16.json 2>/dev/null
for f in gpurun_out/bench_cfg*.json; do echo $f; cut -c1-330 $f; done
