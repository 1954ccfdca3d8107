#!/bin/bash
# round 3, pass N: tile shapes for the rectangular f32 convs (text encoder / flow) at the benched batch sizes
mkdir -p gpurun_out
( WETTS_BENCH_B=64 WETTS_XSHAPES=192:768:3:128,768:192:3:128,192:576:1:128,192:192:1:128 python tools/bench_conv.py 0,6,2,3,4,5
  WETTS_BENCH_B=16 WETTS_XSHAPES=192:768:3:128,768:192:3:128,192:576:1:128,192:192:1:128 python tools/bench_conv.py 0,6,2,3,4,5
  WETTS_BENCH_B=16 WETTS_XSHAPES=192:384:5:760,192:384:1:760,96:192:1:760,192:96:1:760,192:512:7:760 python tools/bench_conv.py 0,6,2,3,4,5
  WETTS_BENCH_B=64 WETTS_XSHAPES=192:384:5:780,192:384:1:780,96:192:1:780,192:96:1:780,192:512:7:780 python tools/bench_conv.py 0,6,2,3,4,5 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/conv_rect_tiles.txt
