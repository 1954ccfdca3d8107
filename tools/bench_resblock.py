#!/usr/bin/env python3
"""ResBlock1 fusion depth at f32 (GPU box only): conv by conv / old fused pair / chain kernel per pair /
chain kernel over the whole ResBlock, on the HiFi-GAN v1 MRF shapes.  "=" : same full-tensor hash as
conv by conv (all forms must be bit-identical).
    python tools/bench_resblock.py [C:k,...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib
lib = benchlib.load()
B, Ty = 16, 864
Ls = {256: Ty * 8, 128: Ty * 64, 64: Ty * 128, 32: Ty * 256}
shapes = [(c, k) for c in (128, 64, 32) for k in (3, 7, 11)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in it.split(":")) for it in sys.argv[1].split(",")]
iters = int(os.environ.get("WETTS_ITERS", "10"))
print(f"{'shape':22s}" + "".join(f"  {m:>22s}" for m in ("conv by conv", "pair32 x npairs", "chain x npairs", "chain whole")))
flags = int(os.environ.get("WETTS_FLAGS", "4"))
for (ch, k) in shapes:
    for npairs, d0 in ((1, 1), (1, 5), (3, 1)):
        row = f"C={ch:3d} k={k:2d} P={npairs} d0={d0}  "
        ref = None
        for mode in (0, 1, 2, 3):
            if mode == 3 and npairs == 1:
                continue
            ms, cs = C.c_double(), C.c_double()
            rc = lib.wetts_bench_resblock(ch, k, npairs, d0, B, Ls[ch], flags, mode, iters, C.byref(ms), C.byref(cs))
            if rc != 0:
                row += f"  {'ERR ' + benchlib.last_error()[:16]:>22s}"
                continue
            tf = npairs * 2 * 2.0 * ch * ch * k * Ls[ch] * B / (ms.value * 1e-3) / 1e12
            same = "" if ref is None else ("=" if cs.value == ref else "!=")
            ref = cs.value if ref is None else ref
            row += f"  {ms.value:8.3f} ms {tf:6.1f} TF {same:2s}"
        print(row, flush=True)
