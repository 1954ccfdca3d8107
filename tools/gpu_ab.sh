#!/bin/bash
# Same-box A/B of builds of the library on one bench line: the copies tmp_ab/libwetts_hip_<name>.so named in $LIBS
# (untracked; they travel with the gpurun snapshot) and "tree" = wetts_amd/lib/libwetts_hip.so, run alternately, twice.
# usage: LIBS="prev tree" gpu_ab.sh <tag> <bench flags...>   -> gpurun_out/ab_<tag>.txt
mkdir -p gpurun_out
TAG=$1; shift
LIBS=${LIBS:-"prev tree"}
L=wetts_amd/lib/libwetts_hip.so
cp $L /tmp/ab_tree.so
OUT=gpurun_out/ab_$TAG.txt
echo "libraries: $LIBS (tmp_ab/libwetts_hip_<name>.so; tree = the tree's own); bench.py $@ --steps 20 --warmup 3" > $OUT
for rep in 1 2; do
  for which in $LIBS; do
    if [ $which = tree ]; then cp /tmp/ab_tree.so $L; else cp tmp_ab/libwetts_hip_$which.so $L; fi
    python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --live-traffic 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('$which'.ljust(8), round(d['ms_per_step'],3),'ms/step', round(d['value']/1e6,2),'M samples/s', 'class frac', round(r.get('frac') or 0,4), 'isolated', round((r.get('isolated') or {}).get('frac') or 0, 4))
" >> $OUT
  done
done
cp /tmp/ab_tree.so $L
cat $OUT
