#!/usr/bin/env python3
"""The bf16 conv inner loop taken apart (GPU box only): v_mfma_f32_32x32x16_bf16 with MB x NB accumulators per wave,
B fragments from LDS (ds_read_b128, the pair kernel's addresses) and A fragments from an L2-resident stream, at 1..4
resident blocks per CU, with the clock the chip sustained.  cfg = MB*10000 + NB*1000 + BM*100 + AM*10 + SYNC."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
rows = [(14000, "1x4 bare (registers)"), (14100, "1x4 B from LDS"), (14010, "1x4 A from L2"), (14110, "1x4 A + B"),
        (14111, "1x4 A + B + barriers"), (22000, "2x2 bare"), (22100, "2x2 B from LDS"), (22110, "2x2 A + B"),
        (22111, "2x2 A + B + barriers"), (24000, "2x4 bare"), (24100, "2x4 B from LDS"), (24110, "2x4 A + B"),
        (24111, "2x4 A + B + barriers"), (12110, "1x2 A + B"), (23110, "2x3 A + B"), (41110, "4x1 A + B")]
rs = int(sys.argv[1]) if len(sys.argv) > 1 else 272
print(f"rs={rs}  TF/s (sustained MHz) at 1 / 2 / 3 / 4 resident 4-wave blocks per CU; bf16 dense peak 2500 TF/s at 2400 MHz")
for cfg, name in rows:
    row = f"{cfg:5d} {name:24s}"
    for per in (1, 2, 3, 4):
        tf, ms, mhz = C.c_double(), C.c_double(), C.c_double()
        rc = lib.wetts_bench_mfma16_loop(cfg, per, rs, 44, 150, C.byref(tf), C.byref(ms), C.byref(mhz))
        row += f"  {tf.value:7.0f} ({mhz.value:4.0f})" if rc == 0 else "      -       "
    print(row, flush=True)
