#!/bin/bash
# One GPU-box pass (round 6): parity tests, rocprofv3 kernel stats + HBM counters for every benched line (f32 headline,
# 16-bit decoder, configs[2], configs[4], Vocos, VITS2 + Vocos, uint8), the bench lines, streaming, MAS.
# Outputs -> gpurun_out/ ; `python tools/summarize_profiles.py r06` copies the judged summaries into profiles/.
# SKIP_TESTS=1 skips the pytest pass (it is also run by tools/gpu_tests.sh).
mkdir -p gpurun_out
R=/root/repo
# the library this pass measures: the digest of its sources (wetts_amd/build.py), stamped into the PMC summary so that
# bench.py can say whether the committed traffic figure belongs to the library that runs (roofline.traffic_current)
cut -c1-16 wetts_amd/lib/build.sha256 > gpurun_out/lib_digest.txt
if [ -z "$SKIP_TESTS" ]; then
  python -m pytest tests -m gpu -q --timeout 900 -rf -s 2>&1 | grep -v "^\[wetts" > gpurun_out/pytest_gpu_full.log
  tail -40 gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu.log
  grep -i "rms\|^OK \|passed\|worst\|ragged row" gpurun_out/pytest_gpu_full.log > gpurun_out/pytest_gpu_margins.txt
  tail -3 gpurun_out/pytest_gpu.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
prof() {  # $1 = tag, rest = bench flags: kernel stats + the two HBM counter passes (separate runs, as the guide prescribes)
  tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_$tag -o r -- python $R/bench.py "$@" --steps 5 --warmup 1 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/prof_stats_$tag.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch_$tag -o r -- python $R/bench.py "$@" --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/pmc_fetch_$tag.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write_$tag -o r -- python $R/bench.py "$@" --steps 1 --warmup 1 --presteps-s 0.3 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/pmc_write_$tag.log 2>&1
  # keep the merged-back scratch small: the per-dispatch traces are not needed once the stats / counter CSVs exist
  find $R/gpurun_out/prof_stats_$tag $R/gpurun_out/pmc_fetch_$tag $R/gpurun_out/pmc_write_$tag -name "*kernel_trace.csv" -delete 2>/dev/null
}
prof baker
# the same line with the stage's chains on ONE stream: per-kernel durations of this trace do not overlap (kernel quality)
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_baker_serial -o r -- python $R/bench.py --decoder-serial --steps 5 --warmup 1 --no-cpu-baseline --live-traffic 0 > $R/gpurun_out/prof_stats_baker_serial.log 2>&1
find $R/gpurun_out/prof_stats_baker_serial -name "*kernel_trace.csv" -delete 2>/dev/null
prof bf16 --decoder-dtype bf16
prof cfg2 --config multilingual
prof stress48k --config stress48k
prof vocos --model vocos
prof vits2vocos --model vits2_vocos_v1
prof uint8 --decoder-dtype uint8
cd $R
python tools/summarize_profiles.py r06 > gpurun_out/traffic_summary.txt 2>&1  # bench.py reads the newest profiles/r*_hbm_traffic.json
python bench.py --steps 20 --warmup 3 > gpurun_out/bench.json 2> gpurun_out/bench.err
cut -c1-420 gpurun_out/bench.json
python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/bench_gpus2_refused.json 2> gpurun_out/bench_gpus2_refused.err; echo "gpus2 exit=$?" | tee -a gpurun_out/bench_gpus2_refused.err
WETTS_BENCH_SINGLE_DEVICE=1 WETTS_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_2rank_dryrun.json 2> gpurun_out/bench_2rank_dryrun.err; echo "dryrun exit=$?"
python bench.py --decoder-serial --steps 20 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_serial.json 2>/dev/null
for dt in bf16 f16 uint8; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --decoder-dtype $dt > gpurun_out/bench_$dt.json 2>/dev/null; done
python bench.py --config multilingual --steps 10 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_cfg2_multilingual_bf16.json 2>/dev/null
python bench.py --config multilingual --decoder-dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_cfg2_multilingual_f32.json 2>/dev/null
python bench.py --config aishell3 --steps 5 --warmup 2 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_cfg3_aishell3.json 2>/dev/null
python bench.py --config stress48k --steps 10 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_cfg4_stress48k_f16.json 2>/dev/null
python bench.py --model vocos --steps 10 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_vocos.json 2>/dev/null
python bench.py --model vits2_vocos_v1 --steps 10 --warmup 3 --no-cpu-baseline --live-traffic 0 > gpurun_out/bench_vits2_vocos.json 2>/dev/null
python bench.py --stream --model v1 > gpurun_out/stream_v1.json 2>/dev/null
python bench.py --stream --model vits2_vocos_v1 --stream-cpu > gpurun_out/stream_vits2_vocos.json 2>/dev/null
python bench.py --mas > gpurun_out/mas.json 2>/dev/null
(export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace -d /tmp/b1 -o b1 --output-format csv -- python tools/trace_b1.py --reps 5 > gpurun_out/b1_run.txt 2>&1; python tools/trace_b1.py --summarize /tmp/b1 > gpurun_out/b1_summary.txt 2>&1)
# SQ counters per kernel symbol: matrix-pipe busy share, wave-cycle split, instruction mix per MFMA (three passes each)
bash tools/gpu_sq_counters.sh mrf > gpurun_out/sq_f32.log 2>&1
{ echo "f32 headline, one-stream schedule (bench.py --decoder-serial), per kernel symbol; VALU counts include the MFMAs themselves"; python tools/sq_table.py mrf "conv_mfma|resblock_chain32"; } > gpurun_out/sq_table_mrf.txt 2>&1
bash tools/gpu_sq_counters.sh mrf16_cfg2 "re:." --config multilingual > gpurun_out/sq_cfg2.log 2>&1
{ echo "configs[2] (bench.py --config multilingual), per kernel symbol; VALU counts include the MFMAs themselves"; python tools/sq_table.py mrf16_cfg2 | head -24; } > gpurun_out/sq_table_mrf16_cfg2.txt 2>&1
bash tools/gpu_sq_counters.sh mrf16_stress48k "re:." --config stress48k > gpurun_out/sq_s48.log 2>&1
{ echo "configs[4] (bench.py --config stress48k), per kernel symbol; VALU counts include the MFMAs themselves"; python tools/sq_table.py mrf16_stress48k | head -20; } > gpurun_out/sq_table_mrf16_stress48k.txt 2>&1
find gpurun_out -name "*counter_collection.csv" -path "*pmc_sq_*" -delete 2>/dev/null
python tools/summarize_profiles.py r06 > gpurun_out/traffic_summary2.txt 2>&1
WETTS_BENCH_B=16 WETTS_XSHAPES=512:1536:1:760,1536:512:1:760,192:384:1:760,192:384:5:760,512:1026:1:760 python tools/bench_conv.py 6,0 > gpurun_out/pw_gemm_microbench.txt 2>&1
WETTS_BENCH_ITERS=10 python tools/bench_conv.py 0 2>&1 | grep -v amdgpu.ids > gpurun_out/conv_microbench.txt
for f in gpurun_out/bench_*.json gpurun_out/bench.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); r=d.get('roofline',{}); print('  ', round(d['value']/1e6,1),'M samples/s', round(d['ms_per_step'],2),'ms', d['dtype'][:30], 'frac', round(r.get('frac') or 0,3), 'mrf_share', round(r.get('mrf_share_of_step',0),3), 'traffic', r.get('traffic'))
" 2>/dev/null; done
du -sh gpurun_out
