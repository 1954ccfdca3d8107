#!/usr/bin/env python3
"""Conv micro-benchmark sweep over the HiFi-GAN v1 MRF shapes (GPU box only)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wetts_amd import _lib  # noqa: E402

lib = _lib.load()
B, Ty = 16, 864
variants = [int(v, 0) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
shapes = []
L = Ty
for C_, u in [(256, 8), (128, 8), (64, 2), (32, 2)]:
    L *= u
    for k in (3, 7, 11):
        shapes.append((C_, k, 1, L, 1 | 2))   # c2-style: lrelu + residual
        shapes.append((C_, k, 5, L, 1))       # c1-style: lrelu, dilation 5
only = os.environ.get("WETTS_SHAPES")
if only:
    keep = {tuple(int(v) for v in it.split(":")) for it in only.split(",")}
    shapes = [sh for sh in shapes if (sh[0], sh[1]) in keep]
print(f"{'shape':34s}" + "".join(f"  v{v:#x}: ms / TF/s   " for v in variants))
for (ch, k, d, L, fl) in shapes:
    row = f"C={ch:3d} k={k:2d} d={d} L={L:6d} fl={fl}      "
    ref = None
    for v in variants:
        ms, cs = C.c_double(), C.c_double()
        rc = lib.wetts_bench_conv(ch, ch, k, d, B, L, fl, v, 5, C.byref(ms), C.byref(cs))
        if rc != 0:
            row += f"  ERR {_lib.last_error()}"
            continue
        tf = 2.0 * ch * ch * k * L * B / (ms.value * 1e-3) / 1e12
        same = "" if ref is None else (" =" if abs(cs.value - ref) <= 1e-6 * max(1, abs(ref)) else " !=")
        ref = cs.value if ref is None else ref
        row += f"  {ms.value:8.3f} {tf:6.1f}{same:3s}"
    print(row, flush=True)
