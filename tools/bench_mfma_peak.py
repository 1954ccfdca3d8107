#!/usr/bin/env python3
"""Sustained f32-MFMA rate calibration (GPU box only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wetts_amd import _lib
lib = _lib.load()
for iters in (2000, 20000):
    for bpc in (1, 2, 3, 4):
        for nacc in (1, 2, 4):
            tf, ms = C.c_double(), C.c_double()
            lib.wetts_bench_mfma_peak(bpc, nacc, iters, C.byref(tf), C.byref(ms))
            print(f"iters={iters:6d} blocks/CU={bpc} (waves/SIMD={bpc}) nacc={nacc}: {tf.value:7.1f} TF/s  {ms.value:8.3f} ms", flush=True)
