#!/usr/bin/env python3
"""Sustained f32-MFMA rate calibration (GPU box only)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
def run(bpc, nacc, iters, note=""):
    tf, ms = C.c_double(), C.c_double()
    lib.wetts_bench_mfma_peak(bpc, nacc, iters, C.byref(tf), C.byref(ms))
    print(f"grid={'%d blocks' % bpc if bpc >= 1000 else '%d/CU' % bpc:12s} nacc={nacc:2d} iters={iters:6d}: {tf.value:7.1f} TF/s {ms.value:8.3f} ms {note}", flush=True)
run(3, 4, 20000, "persistent, constant operands")
run(3, -4, 20000, "persistent, random operands")
# block turnover: same total work as the C=128 k=11 conv (6912 blocks x 2816 MFMA/wave)
run(6912, -4, 88, "6912 short blocks (2816 MFMA/wave each)")
run(6912, -4, 88, "repeat")
run(1728, -4, 48, "1728 short blocks (1536 MFMA/wave, C=256 k=3 like)")
run(13824, -4, 3, "13824 tiny blocks (96 MFMA/wave, C=32 k=3 like)")
