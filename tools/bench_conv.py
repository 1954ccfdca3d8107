#!/usr/bin/env python3
"""Conv micro-benchmark sweep over the HiFi-GAN v1 MRF shapes (GPU box only), through the C ABI's
measurement entry point wetts_bench_conv (wetts_amd/csrc/bench_conv.hip).

    python tools/bench_conv.py [variants]        variants: comma-separated ints, default "0,1"

Variant 16 = fused pair / chain kernel, 32 = the same pair as two launches (pair mode); other values
select a conv tile shape (WETTS_CONV_VARIANT).  Environment:
    WETTS_CONV_FLAGS=16|32   16-bit decoder kernels (bf16 | f16) instead of f32
    WETTS_PAIR=1             ResBlock1 pairs (c1 at dilation d, c2 at 1): compare variants 32 and 16
    WETTS_RB2=1              with WETTS_PAIR: ResBlock2 chains (both convs residual, c2 at 2d)
    WETTS_SHAPES=C:k,...     restrict the sweep
A trailing "=" / "!=" compares a full-tensor hash with the first variant (fused vs two launches
must be "=")."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import benchlib  # noqa: E402
from wetts_amd import _lib  # noqa: E402

lib = benchlib.load()
B, Ty = int(os.environ.get("WETTS_BENCH_B", "16")), 864
variants = [int(v, 0) for v in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
shapes = []
L = Ty
for C_, u in [(256, 8), (128, 8), (64, 2), (32, 2)]:
    L *= u
    for k in (3, 7, 11):
        shapes.append((C_, k, 1, L, 1 | 2))   # c2-style: lrelu + residual
        shapes.append((C_, k, 5, L, 1))       # c1-style: lrelu, dilation 5
extra = int(os.environ.get("WETTS_CONV_FLAGS", "0"), 0)  # 16: bf16 decoder kernel, 32: f16
esz = 2 if extra & 48 else 4
rb2 = os.environ.get("WETTS_RB2") == "1"  # with WETTS_PAIR: ResBlock2 chains (c2 residual, dilation 2d)
pair = os.environ.get("WETTS_PAIR") == "1"  # ResBlock1 (c1 dil d, c2 dil 1) pairs: variants 32 / 16
if pair:
    shapes = [(c, k, d, L_, 1 | 2) for (c, k, _, L_, fl) in shapes if fl == 3 for d in (1, 3, 5)]
only = os.environ.get("WETTS_SHAPES")
if only:
    keep = {tuple(int(v) for v in it.split(":")) for it in only.split(",")}
    shapes = [sh for sh in shapes if (sh[0], sh[1]) in keep]
xs = os.environ.get("WETTS_XSHAPES")  # cin:cout:k:L,... rectangular convs (flow / encoder shapes)
if xs:
    print(f"{'shape':34s}" + "".join(f"  v{v:#06x}: ms TF/s      " for v in variants))
    for it in xs.split(","):
        ci, co, k, L = (int(v) for v in it.split(":"))
        row = f"{ci:4d}->{co:4d} k={k:2d} L={L:6d}          "
        first = None
        for v in variants:
            ms, cs = C.c_double(), C.c_double()
            rc = lib.wetts_bench_conv(ci, co, k, 1, B, L, extra, v, 20, C.byref(ms), C.byref(cs))
            if rc != 0:
                row += f"  ERR {_lib.last_error()}"
                continue
            first = cs.value if first is None else first
            row += f"  {ms.value:7.4f} {2.0 * ci * co * k * L * B / (ms.value * 1e-3) / 1e12:6.1f} {'=' if cs.value == first else '!='}"
        print(row, flush=True)
    sys.exit(0)
print(f"{'shape':34s}" + "".join(f"  v{v:#06x}: ms TF/s GB/s  " for v in variants))
for (ch, k, d, L, fl) in shapes:
    row = f"C={ch:3d} k={k:2d} d={d} L={L:6d} fl={fl}      "
    ref = None
    for v in variants:
        ms, cs = C.c_double(), C.c_double()
        rc = lib.wetts_bench_conv(ch, ch, k, d, B, L, fl | extra | (64 if rb2 else 0), v,
                                  int(os.environ.get("WETTS_BENCH_ITERS", "5")),
                                  C.byref(ms), C.byref(cs))
        if rc != 0:
            row += f"  ERR {_lib.last_error()}"
            continue
        tf = (2 if pair else 1) * 2.0 * ch * ch * k * L * B / (ms.value * 1e-3) / 1e12
        same = "" if ref is None else (" =" if abs(cs.value - ref) <= 1e-6 * max(1, abs(ref)) else " !=")
        ref = cs.value if ref is None else ref
        ntens = 5 if pair else (2 + (1 if fl & 2 else 0))  # per-conv algorithmic accounting
        gbs = ntens * ch * L * B * esz / (ms.value * 1e-3) / 1e9
        row += f"  {ms.value:7.3f} {tf:6.1f} {gbs:5.0f}{same:3s}"
    print(row, flush=True)
