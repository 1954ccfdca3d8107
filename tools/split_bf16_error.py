#!/usr/bin/env python3
"""CPU estimate of the error of split-bf16 emulation of an f32 GEMM/conv: an f32
operand is hi + mid + lo with three bfloat16 terms; the matrix cores multiply bf16 pairs exactly and
accumulate in f32.  Compares, against a float64 reference on a K = 256 x 7 contraction (C = 256,
k = 7 conv), the f32 MFMA chain (2 k per step) with 3 / 6 / 9 cross products accumulated in f32 per
16-k MFMA.  No GPU needed:  python tools/split_bf16_error.py"""
import numpy as np


def bf(x):  # round to nearest even to bfloat16, returned as f32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    h = bf(x)
    r = (x - h).astype(np.float32)
    m = bf(r)
    return h, m, bf((r - m).astype(np.float32))


def main():
    rng = np.random.default_rng(0)
    K = 256 * 7
    x = rng.standard_normal((K, 512)).astype(np.float32)
    x = np.where(x > 0, x, 0.1 * x).astype(np.float32)  # leaky-relu'd activations
    w = (rng.standard_normal((64, K)) / np.sqrt(K)).astype(np.float32)
    ref = w.astype(np.float64) @ x.astype(np.float64)
    rms = np.sqrt((ref ** 2).mean())
    D = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)  # noqa: E731  exact products
    (wh, wm, wl), (xh, xm, xl) = split3(w), split3(x)
    terms = [(wh, xh), (wh, xm), (wm, xh), (wm, xm), (wh, xl), (wl, xh), (wm, xl), (wl, xm), (wl, xl)]

    def report(name, y):
        e = y - ref
        print(f"{name:34s} rel rms err {np.sqrt((e ** 2).mean()) / rms:.3e}   max abs {np.abs(e).max():.3e}")

    acc = np.zeros_like(ref, dtype=np.float32)
    for k0 in range(0, K, 2):  # v_mfma_f32_32x32x2_f32 chain: one f32 rounding per 2 k
        acc = (acc + D(w[:, k0:k0 + 2], x[k0:k0 + 2])).astype(np.float32)
    report("f32 MFMA chain (today)", acc)
    for n in (3, 6, 9):
        acc = np.zeros_like(ref, dtype=np.float32)
        for k0 in range(0, K, 16):  # v_mfma_f32_32x32x16_bf16: one f32 rounding per 16 k and term set
            s = slice(k0, k0 + 16)
            acc = (acc + sum(D(a[:, s], b[s]) for a, b in terms[:n])).astype(np.float32)
        report(f"split bf16, {n} products, f32 acc", acc)
        report(f"   truncation alone ({n} products)", sum(D(a, b) for a, b in terms[:n]))


if __name__ == "__main__":
    main()
