#!/usr/bin/env python3
"""The f32 conv inner loop taken apart (GPU box only): what each ingredient costs the MFMA stream.
mode bits: 1 B from LDS, 2 A from global/L2, 4 barrier every 6 groups, 8 B reads pipelined one k-step
ahead (sched_barriers), 16 = 64 VALU per group of 16 MFMAs."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
names = {0: "bare (constants)", 1: "B from LDS, compiler-placed", 9: "B from LDS, pipelined", 2: "A from L2",
         3: "A + B (compiler)", 11: "A + B pipelined", 15: "A + B pipelined + barriers", 7: "A + B (compiler) + barriers",
         16: "bare + 64 VALU/group", 27: "A + B pipelined + 64 VALU/group", 4: "bare + barriers", 13: "B pipelined + barriers"}
print(f"{'mode':34s}" + "".join(f"  {o} blk/CU" for o in (1, 2, 4)))
for mode in (0, 1, 9, 2, 3, 11, 4, 13, 7, 15, 16, 27):
    row = f"{mode:2d} {names[mode]:31s}"
    for kb in (120, 70, 36):
        tf, ms = C.c_double(), C.c_double()
        rc = lib.wetts_bench_mfma_loop(mode, kb, 12, 400, C.byref(tf), C.byref(ms))
        row += f"  {tf.value:8.1f}" if rc == 0 else "  ERR"
    print(row, flush=True)
