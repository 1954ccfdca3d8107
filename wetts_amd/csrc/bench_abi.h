// C entry points of libwetts_bench.so: kernel micro-benchmarks (tools/bench_*.py).  Measurement
// tooling only -- NOT part of the product ABI (include/wetts_hip.h) and not linked into
// libwetts_hip.so; they allocate their own buffers and drive the product's launchers directly.
#pragma once
#include <stdint.h>

extern "C" {
// One Conv1d(Cin->Cout, k, dilation) on [B,Cin,T] with pseudo-random data.
// flags: 1 = leaky-relu prologue, 2 = residual add, 4 = accumulate into the output, 8 = MRF mean
// division, 16 / 32 = bf16 / f16 decoder kernels, 64 = ResBlock2 chain (pair modes).
// variant: low byte = tile override (WETTS_CONV_VARIANT values) or 16 = fused pair, 32 = the same
// pair as two launches.
int32_t wetts_bench_conv(int32_t Cin, int32_t Cout, int32_t k, int32_t dil, int32_t B, int32_t T,
                         int32_t flags, int32_t variant, int32_t iters, double* ms_out,
                         double* checksum_out);
int32_t wetts_set_conv_variant(int32_t variant);
// ResBlock1 of `npairs` (c1 at dilation first_dil / 3 / 5, c2 at 1) pairs at f32.  mode 0: conv by
// conv; 1: resblock_pair32_kernel per pair; 2: resblock_chain32_kernel per pair; 3: the chain kernel
// once for all pairs.  flags: 4 accumulate into out, 8 divide by 3.  checksum = full-tensor hash.
int32_t wetts_bench_resblock(int32_t C, int32_t k, int32_t npairs, int32_t first_dil, int32_t B,
                             int32_t T, int32_t flags, int32_t mode, int32_t iters, double* ms_out,
                             double* checksum_out);
// Sustained v_mfma_f32_32x32x2_f32 rate with `blocks_per_cu` 4-wave blocks per CU (>= 1000: that
// many blocks) and |nacc| (1,2,4) independent accumulators per wave (< 0: random operands).
int32_t wetts_bench_mfma_peak(int32_t blocks_per_cu, int32_t nacc, int32_t iters, double* tflops,
                              double* ms);
// Does the f32 vector pipe run beside the f32 matrix pipe?  mode 0: `nv` v_fma_f32 interleaved after
// every MFMA of the same wave; mode 1: wave-specialised (4 MFMA waves + 4 VALU-only waves per block,
// one of each per SIMD; the VALU waves run nv FMAs per MFMA of their sibling).  Returns both rates.
// The conv inner loop taken apart (see mfma_loop_kernel in bench_conv.hip for the mode bits);
// lds_kb sets the dynamic LDS per block and with it the blocks resident per CU (160 KB / lds_kb, <= 4).
int32_t wetts_bench_mfma_loop(int32_t mode, int32_t lds_kb, int32_t groups, int32_t iters,
                              double* tflops, double* ms);
// cfg = MB*1000 + NB*100 + BM*10 + AM (see mfma_loop2_kernel): wave-tile shape and operand paths.
int32_t wetts_bench_mfma_loop2(int32_t cfg, int32_t lds_kb, int32_t groups, int32_t iters,
                               double* tflops, double* ms);
int32_t wetts_bench_mfma_valu(int32_t mode, int32_t nv, int32_t iters, double* tflops_mfma,
                              double* tflops_valu, double* ms);
// The bf16 loop taken apart: cfg = MB*10000 + NB*1000 + BM*100 + AM*10 + SYNC (mfma16_loop_kernel); also returns the
// clock the chip sustained (s_memtime cycles per 100 MHz tick).
int32_t wetts_bench_mfma16_loop(int32_t cfg, int32_t blocks_per_cu, int32_t rs, int32_t groups, int32_t iters,
                                double* tflops, double* ms, double* mhz);
}
