#!/bin/bash
# Same-box A/B of two builds of the library on one bench line: tmp_ab/libwetts_hip_prev.so (A, a copy of an earlier
# build; untracked, travels with the gpurun snapshot) against wetts_amd/lib/libwetts_hip.so (B), alternating A B A B.
# usage: gpu_ab.sh <tag> <bench flags...>   -> gpurun_out/ab_<tag>.txt
mkdir -p gpurun_out
TAG=$1; shift
L=wetts_amd/lib/libwetts_hip.so
cp $L /tmp/ab_new.so
OUT=gpurun_out/ab_$TAG.txt
echo "A = tmp_ab/libwetts_hip_prev.so, B = the tree's library; bench.py $@ --steps 20 --warmup 3" > $OUT
for rep in 1 2; do
  for which in A B; do
    if [ $which = A ]; then cp tmp_ab/libwetts_hip_prev.so $L; else cp /tmp/ab_new.so $L; fi
    python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('$which', round(d['ms_per_step'],3),'ms/step', round(d['value']/1e6,2),'M samples/s', 'class frac', round(r.get('frac') or 0,4), 'isolated', (r.get('isolated') or {}).get('frac'), 'class ms', r.get('class_ms_per_step'))
" >> $OUT
  done
done
cp /tmp/ab_new.so $L
cat $OUT
