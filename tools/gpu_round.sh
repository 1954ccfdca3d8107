#!/bin/bash
# One GPU-box pass: parity tests, bench, rocprofv3 kernel stats + HBM counters. Outputs -> gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 10 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
R=/root/repo
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
ls -R $R/gpurun_out/prof_stats $R/gpurun_out/pmc_fetch | head -20
du -sh $R/gpurun_out
# 16-bit decoder (BASELINE configs[2]/[4] precision): bench lines + kernel stats
cd $R
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder-dtype bf16 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder-dtype f16 > gpurun_out/bench_f16.json 2> gpurun_out/bench_f16.err
cat gpurun_out/bench_bf16.json | cut -c1-300
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_bf16 -o r -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --decoder-dtype bf16 > $R/gpurun_out/prof_stats_bf16.log 2>&1
cd $R
# fused-pair microbenchmarks (f32 and bf16): two launches (0x20) vs fused (0x10)
WETTS_PAIR=1 WETTS_SHAPES=128:3,128:7,128:11,64:3,64:7,64:11,32:3,32:7,32:11 python tools/bench_conv.py 32,16 > gpurun_out/conv32_fused_pair.txt 2>&1
WETTS_PAIR=1 WETTS_CONV_FLAGS=16 WETTS_SHAPES=128:3,128:7,128:11,64:3,64:7,64:11,32:3,32:7,32:11 python tools/bench_conv.py 32,16 > gpurun_out/conv16_fused_pair.txt 2>&1
python tools/bench_conv.py 0 > gpurun_out/conv_microbench.txt 2>&1
# streaming (chunked decoder) latency; and the other model families' bench lines
python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null
python bench.py --stream --model vits2_vocos_v1 --stream-cpu > gpurun_out/stream_vits2_vocos.json 2>/dev/null
python bench.py --model vocos --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_vocos.json 2>/dev/null
python bench.py --model vits2_vocos_v1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_vits2_vocos.json 2>/dev/null
# BASELINE configs[2] (v3, B=64, bf16) and configs[4] (builder-defined 48 kHz stress shape, f16)
python bench.py --model v3 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline --decoder-dtype bf16 > gpurun_out/bench_cfg3_v3_b64_bf16.json 2>/dev/null
python bench.py --model v3 --batch 64 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_cfg3_v3_b64_f32.json 2>/dev/null
python bench.py --model stress48k --steps 10 --warmup 3 --no-cpu-baseline --decoder-dtype f16 > gpurun_out/bench_cfg5_stress48k_f16.json 2>/dev/null
# control flow of the multi-rank bench (two ranks sharing this one GPU, gloo for the broadcast):
# the driver runs the real N = 2/4/8 RCCL scaling bench at round end
WETTS_BENCH_SINGLE_DEVICE=1 WETTS_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_2rank_dryrun.json 2> gpurun_out/bench_2rank_dryrun.err
tail -c 400 gpurun_out/bench_2rank_dryrun.json
