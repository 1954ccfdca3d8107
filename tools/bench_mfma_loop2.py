#!/usr/bin/env python3
"""Wave-tile shape and operand paths of the f32 conv inner loop (GPU box only).
cfg = MB NB BM AM: MB x NB accumulators per wave; BM 0 const / 1 ds_read_b32 rows / 2 ds_read_b128 [col][ch];
AM 0 const / 1 global dwordx4 / 2 LDS b128."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
print(f"{'MB NB BM AM':14s}" + "".join(f"  {o} blk/CU" for o in (1, 2, 3, 4)))
for cfg in (1400, 1410, 1420, 1411, 1421, 1412, 1422, 2400, 2410, 2420, 2411, 2421, 2422, 2221, 2211, 1800, 1811, 1821):
    row = f"{cfg // 1000}  {(cfg // 100) % 10}  {(cfg // 10) % 10}  {cfg % 10}    "
    for kb in (120, 70, 50, 36):
        tf, ms = C.c_double(), C.c_double()
        rc = lib.wetts_bench_mfma_loop2(cfg, kb, 12, 400, C.byref(tf), C.byref(ms))
        row += f"  {tf.value:8.1f}" if rc == 0 else f"  ERR{rc}"
    print(row, flush=True)
