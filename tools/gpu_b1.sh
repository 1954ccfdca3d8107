#!/bin/bash
# B = 1 latency anatomy on the GPU box: kernel trace of encoder / first decoder window calls
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
if [ -n "$B1_TESTS" ]; then timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 2>&1 | tail -15 > $O/pytest_gpu_b1.log; tail -4 $O/pytest_gpu_b1.log; fi
timeout 300 rocprofv3 --kernel-trace -d /tmp/b1 -o b1 --output-format csv -- python tools/trace_b1.py --reps 5 > $O/b1_run.txt 2>&1
python tools/trace_b1.py --summarize /tmp/b1 > $O/b1_summary.txt 2>&1
grep -c enc_ms $O/b1_run.txt; grep "enc_ms\|dec_ms" $O/b1_run.txt | tail -10 | tr '\n' ' '; echo; cat $O/b1_summary.txt
python bench.py --stream --model v1 > $O/stream_v1_b1.json 2>/dev/null; cut -c1-700 $O/stream_v1_b1.json
WETTS_TUNE=small_max_tiles=0 python bench.py --stream --model v1 > $O/stream_v1_b1_nosplit.json 2>/dev/null; cut -c1-700 $O/stream_v1_b1_nosplit.json
