#!/usr/bin/env python3
"""Does the f32 vector pipe run beside the f32 matrix pipe on gfx950?  (GPU box only)
mode 0: nv v_fma_f32 issued after every v_mfma_f32_32x32x2_f32 of the same wave (4 waves/block, 2 blocks/CU);
mode 1: 4 MFMA waves + 4 VALU-only waves per block (one of each per SIMD)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
def run(mode, nv, iters=4000):
    tm, tv, ms = C.c_double(), C.c_double(), C.c_double()
    rc = lib.wetts_bench_mfma_valu(mode, nv, iters, C.byref(tm), C.byref(tv), C.byref(ms))
    print(f"mode={mode} nv={nv:2d}: mfma {tm.value:7.1f} TF/s + valu {tv.value:6.1f} TF/s = {tm.value + tv.value:7.1f}  ({ms.value:.3f} ms) rc={rc}", flush=True)
for nv in (0, 2, 4, 8, 12, 16):
    run(0, nv)
for nv in (0, 2, 4, 8, 16, 32):
    run(1, nv)
