// Measurement-only entry points of the C ABI (wetts_bench_conv, wetts_bench_mfma_peak,
// wetts_set_conv_variant): conv micro-benchmarks with ablation / variant switches.  They allocate
// their own buffers and are not on the product path; kept out of model.hip so that file holds only
// the model (layout, weights, stage functions).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/wetts_hip.h"
#include "bench_abi.h"
#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"
#include "resblock32.h"
#include "kernels.h"

using namespace wetts;

extern "C" {

// ---------------------------------------------------------------------------------------------
// conv micro-benchmark (measurement helper, allocates its own buffers; not on the product path)
// ---------------------------------------------------------------------------------------------
namespace wetts {
__global__ void fill_pseudo_kernel(float* p, int64_t n, unsigned seed, float scale) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + seed;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
  p[i] = ((float)(h & 0xffffff) / 8388608.f - 1.f) * scale;
}
static int32_t fill_pseudo(float* p, int64_t n, unsigned seed, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fill_pseudo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n,
                     seed, scale);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}
}  // namespace wetts

// 16-bit decoder conv microbench (flags & 16: bf16, & 32: f16); variant 16 = fused pair kernel,
// 32 = the same pair as two launches.  (The round-1 ablation instantiations are gone; their results
// are kept in profiles/r01_conv16_ablation.txt.)
static int32_t bench_conv_16bit(int32_t Cin, int32_t Cout, int32_t k, int32_t dil, int32_t B,
                                int32_t T, int32_t flags, int32_t variant, int32_t iters,
                                double* ms_out, double* checksum_out) {
  hipStream_t s = nullptr;
  const int f16 = (flags & 32) ? 1 : 0;
  const int64_t nx = (int64_t)B * Cin * T, no = (int64_t)B * Cout * T, nw = (int64_t)Cout * Cin * k;
  float *xf = nullptr, *rf = nullptr, *w = nullptr, *bias = nullptr;
  unsigned short *x = nullptr, *o = nullptr, *r = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&xf, nx * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&rf, no * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&x, nx * 2));
  WETTS_HIP_CHECK(hipMalloc((void**)&o, no * 2));
  WETTS_HIP_CHECK(hipMalloc((void**)&r, no * 2));
  WETTS_HIP_CHECK(hipMalloc((void**)&w, nw * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&bias, (int64_t)Cout * 4));
  WETTS_TRY(fill_pseudo(xf, nx, 1, 1.f, s));
  WETTS_TRY(fill_pseudo(rf, no, 2, 1.f, s));
  WETTS_TRY(fill_pseudo(w, nw, 3, 1.f / sqrtf((float)Cin * k), s));
  WETTS_TRY(fill_pseudo(bias, Cout, 4, 0.1f, s));
  WETTS_TRY(k_cf32_to_cl16(xf, x, B, Cin, T, f16, s));
  WETTS_TRY(k_cf32_to_cl16(rf, r, B, Cout, T, f16, s));
  WETTS_HIP_CHECK(hipMemsetAsync(o, 0, no * 2, s));
  PackedConvB pc;
  WETTS_TRY(pack_conv_weight_bf16(w, bias, Cout, Cin, k, dil, (k * dil - dil) / 2, 0, 0, f16, s, &pc));
  if (variant & 48) {  // ResBlock1 pair: 16 = fused kernel, 32 = two conv launches
    WETTS_REQUIRE(Cin == Cout, "pair bench needs Cin == Cout");
    PackedConvB pc2;
    float* w2 = nullptr;
    unsigned short* ft = nullptr;
    WETTS_HIP_CHECK(hipMalloc((void**)&w2, nw * 4));
    WETTS_HIP_CHECK(hipMalloc((void**)&ft, no * 2));
    WETTS_TRY(fill_pseudo(w2, nw, 5, 1.f / sqrtf((float)Cin * k), s));
    WETTS_TRY(pack_conv_weight_bf16(w2, bias, Cout, Cin, k, 1, (k - 1) / 2, 0, 0, f16, s, &pc2));
    auto run = [&]() -> int32_t {
      if (variant & 16) {
        ResPairParams pp;
        memset(&pp, 0, sizeof(pp));
        pp.x = x; pp.out = o; pp.T = T; pp.B = B; pp.accum = (flags & 4) ? 1 : 0;
        pp.out_div = (flags & 8) ? 3.f : 1.f; pp.slope = 0.1f;
        return launch_resblock_pair16(pc, pc2, pp, s);
      }
      ConvBParams p1;
      memset(&p1, 0, sizeof(p1));
      p1.x = x; p1.x_bs = (int64_t)Cin * T; p1.Cin = Cin; p1.Tin = T; p1.in_act = IN_LRELU;
      p1.in_slope = 0.1f; p1.out = ft; p1.o_bs = p1.x_bs; p1.cout = Cout; p1.Tout = T;
      p1.out_div = 1.f; p1.B = B;
      int32_t rc1 = launch_conv_bf16(pc, p1, s);
      if (rc1 != WETTS_OK) return rc1;
      ConvBParams p2 = p1;
      p2.x = ft; p2.out = o; p2.res = x; p2.r_bs = p1.x_bs; p2.accum = (flags & 4) ? 1 : 0;
      p2.out_div = (flags & 8) ? 3.f : 1.f;
      return launch_conv_bf16(pc2, p2, s);
    };
    int32_t rc = WETTS_OK;
    for (int i = 0; i < 2 && rc == WETTS_OK; ++i) rc = run();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == WETTS_OK; ++i) rc = run();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    if (checksum_out && rc == WETTS_OK) {
      // full-tensor fingerprint of ONE application on a known previous `out` (the residual copy)
      (void)hipMemcpyAsync(o, r, no * 2, hipMemcpyDeviceToDevice, s);
      rc = run();
      std::vector<unsigned short> host((size_t)no);
      (void)hipMemcpy(host.data(), o, host.size() * 2, hipMemcpyDeviceToHost);
      uint64_t hsh = 1469598103934665603ull;
      for (size_t i = 0; i < host.size(); ++i) hsh = (hsh ^ host[i]) * 1099511628211ull;
      *checksum_out = (double)(hsh >> 12);  // 52 bits survive the double
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    free_packed_bf16(&pc);
    free_packed_bf16(&pc2);
    (void)hipFree(xf); (void)hipFree(rf); (void)hipFree(x); (void)hipFree(o); (void)hipFree(r);
    (void)hipFree(w); (void)hipFree(w2); (void)hipFree(ft); (void)hipFree(bias);
    return rc;
  }
  ConvBParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.x_bs = (int64_t)Cin * T; p.Cin = Cin; p.Tin = T;
  p.out = o; p.o_bs = (int64_t)Cout * T; p.cout = Cout; p.Tout = T; p.out_div = 1.f; p.B = B;
  if (flags & 1) { p.in_act = IN_LRELU; p.in_slope = 0.1f; }
  if (flags & 2) { p.res = r; p.r_bs = (int64_t)Cout * T; }
  if (flags & 4) p.accum = 1;
  int32_t rc = WETTS_OK;
  for (int i = 0; i < 2 && rc == WETTS_OK; ++i) rc = launch_conv_bf16(pc, p, s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && rc == WETTS_OK; ++i) rc = launch_conv_bf16(pc, p, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  if (checksum_out) {
    if (p.accum) {  // fingerprint of a single application
      (void)hipMemsetAsync(o, 0, no * 2, s);
      rc = launch_conv_bf16(pc, p, s);
    }
    std::vector<unsigned short> host((size_t)(no < 65536 ? no : 65536));
    (void)hipMemcpy(host.data(), o, host.size() * 2, hipMemcpyDeviceToHost);
    double cs = 0;
    for (size_t i = 0; i < host.size(); ++i) {
      float v;
      if (f16) {
        _Float16 h;
        memcpy(&h, &host[i], 2);
        v = (float)h;
      } else {
        unsigned u = (unsigned)host[i] << 16;
        memcpy(&v, &u, 4);
      }
      cs += (double)v * (double)((i % 7) + 1);
    }
    *checksum_out = cs;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  free_packed_bf16(&pc);
  (void)hipFree(xf); (void)hipFree(rf); (void)hipFree(x); (void)hipFree(o); (void)hipFree(r);
  (void)hipFree(w); (void)hipFree(bias);
  return rc;
}

int32_t wetts_bench_conv(int32_t Cin, int32_t Cout, int32_t k, int32_t dil, int32_t B, int32_t T,
                         int32_t flags, int32_t variant, int32_t iters, double* ms_out,
                         double* checksum_out) {
  WETTS_REQUIRE(ms_out && Cin > 0 && Cout > 0 && k > 0 && B > 0 && T > 0 && iters > 0,
                "bad argument");
  hipStream_t s = nullptr;
  const int64_t nx = (int64_t)B * Cin * T, no = (int64_t)B * Cout * T, nw = (int64_t)Cout * Cin * k;
  if (flags & 48) return bench_conv_16bit(Cin, Cout, k, dil, B, T, flags, variant, iters, ms_out, checksum_out);
  float *x = nullptr, *o = nullptr, *r = nullptr, *w = nullptr, *bias = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&x, nx * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&o, no * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&r, no * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&w, nw * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&bias, (int64_t)Cout * 4));
  WETTS_TRY(fill_pseudo(x, nx, 1, 1.f, s));
  WETTS_TRY(fill_pseudo(r, no, 2, 1.f, s));
  WETTS_TRY(fill_pseudo(w, nw, 3, 1.f / sqrtf((float)Cin * k), s));
  WETTS_TRY(fill_pseudo(bias, Cout, 4, 0.1f, s));
  WETTS_HIP_CHECK(hipMemsetAsync(o, 0, no * 4, s));
  PackedConv pc;
  WETTS_TRY(pack_conv_weight(w, bias, Cout, Cin, k, dil, (k * dil - dil) / 2, 0, 0, s, &pc));
  if (variant & 48) {  // ResBlock1 pair: 16 = fused kernel, 32 = two conv launches
    WETTS_REQUIRE(Cin == Cout, "pair bench needs Cin == Cout");
    PackedConv pc2;
    float *w2 = nullptr, *ft = nullptr;
    WETTS_HIP_CHECK(hipMalloc((void**)&w2, nw * 4));
    WETTS_HIP_CHECK(hipMalloc((void**)&ft, no * 4));
    WETTS_TRY(fill_pseudo(w2, nw, 5, 1.f / sqrtf((float)Cin * k), s));
    const bool rb2 = (flags & 64) != 0;  // ResBlock2 chain: both convs residual, c2 at dilation 2*dil
    const int dil2 = rb2 ? 2 * dil : 1;
    WETTS_TRY(pack_conv_weight(w2, bias, Cout, Cin, k, dil2, (k - 1) / 2 * dil2, 0, 0, s, &pc2));
    auto run = [&]() -> int32_t {
      if ((variant & 16) && rb2) {
        ResPair32Params pp;
        memset(&pp, 0, sizeof(pp));
        pp.x = x; pp.out = o; pp.T = T; pp.B = B; pp.accum = (flags & 4) ? 1 : 0;
        pp.out_div = (flags & 8) ? 3.f : 1.f; pp.slope = 0.1f;
        return launch_resblock2_chain32(pc, pc2, pp, s);
      }
      if (variant & 16) {
        ResPair32Params pp;
        memset(&pp, 0, sizeof(pp));
        pp.x = x; pp.out = o; pp.T = T; pp.B = B; pp.accum = (flags & 4) ? 1 : 0;
        pp.out_div = (flags & 8) ? 3.f : 1.f; pp.slope = 0.1f;
        return launch_resblock_pair32(pc, pc2, pp, s);
      }
      ConvParams p1 = conv_io(x, Cin, T, ft, Cout, B);
      p1.in_act = IN_LRELU; p1.in_slope = 0.1f; p1.tag = 1;
      if (rb2) { p1.res = x; p1.r_bs = (int64_t)Cout * T; p1.r_cs = T; }
      int32_t rc1 = launch_conv(pc, p1, s);
      if (rc1 != WETTS_OK) return rc1;
      ConvParams p2 = conv_io(ft, Cin, T, o, Cout, B);
      p2.in_act = IN_LRELU; p2.in_slope = 0.1f; p2.tag = 1;
      p2.res = rb2 ? ft : x; p2.r_bs = (int64_t)Cout * T; p2.r_cs = T;
      p2.accum = (flags & 4) ? 1 : 0;
      p2.out_div = (flags & 8) ? 3.f : 1.f;
      return launch_conv(pc2, p2, s);
    };
    int32_t rc = WETTS_OK;
    const int saved_variant = conv_variant();
    set_conv_variant(0);
    for (int i = 0; i < 2 && rc == WETTS_OK; ++i) rc = run();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, s);
    for (int i = 0; i < iters && rc == WETTS_OK; ++i) rc = run();
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    *ms_out = ms / iters;
    if (checksum_out && rc == WETTS_OK) {  // full-tensor hash of ONE application on a known `out`
      (void)hipMemcpyAsync(o, r, no * 4, hipMemcpyDeviceToDevice, s);
      rc = run();
      std::vector<uint32_t> host((size_t)no);
      (void)hipMemcpy(host.data(), o, host.size() * 4, hipMemcpyDeviceToHost);
      uint64_t hsh = 1469598103934665603ull;
      for (size_t i = 0; i < host.size(); ++i) hsh = (hsh ^ host[i]) * 1099511628211ull;
      *checksum_out = (double)(hsh >> 12);
    }
    set_conv_variant(saved_variant);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    free_packed(&pc);
    free_packed(&pc2);
    (void)hipFree(x); (void)hipFree(o); (void)hipFree(r); (void)hipFree(w); (void)hipFree(w2);
    (void)hipFree(ft); (void)hipFree(bias);
    return rc;
  }
  ConvParams p = conv_io(x, Cin, T, o, Cout, B);
  if (flags & 1) { p.in_act = IN_LRELU; p.in_slope = 0.1f; }
  if (flags & 2) { p.res = r; p.r_bs = (int64_t)Cout * T; p.r_cs = T; }
  if (flags & 4) { p.accum = 1; }
  p.tag = 1;  // the MRF launch class (own kernel symbol)
  const int saved = conv_variant();
  set_conv_variant(variant & 0xff);
  int32_t rc = WETTS_OK;
  for (int i = 0; i < 2 && rc == WETTS_OK; ++i) rc = launch_conv(pc, p, s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && rc == WETTS_OK; ++i) rc = launch_conv(pc, p, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  set_conv_variant(saved);
  if (checksum_out) {
    // cheap order-insensitive fingerprint so variants can be compared for equality
    std::vector<float> host((size_t)(no < 65536 ? no : 65536));
    (void)hipMemcpy(host.data(), o, host.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (size_t i = 0; i < host.size(); ++i) cs += (double)host[i] * (double)((i % 7) + 1);
    *checksum_out = cs;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  free_packed(&pc);
  (void)hipFree(x); (void)hipFree(o); (void)hipFree(r); (void)hipFree(w); (void)hipFree(bias);
  return rc;
}

}  // extern "C"

namespace wetts {
typedef float f32x16 __attribute__((ext_vector_type(16)));
// ------------------------------------------------------------------------------------------
// calibration: sustained rate of v_mfma_f32_32x32x2_f32 on this chip (no memory traffic)
// ------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + (threadIdx.x & 7) * 1e-3f, b = b0 + (threadIdx.x & 3) * 1e-3f;
  if (a0 < 0.f) {
    // random-operand mode: 8 distinct pseudo-random A and B registers per lane (data toggling
    // like a real conv), products have random sign so the accumulators random-walk
    float ar[8], br[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned h = (threadIdx.x * 8 + u + blockIdx.x * 2048) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      ar[u] = ((float)(h & 0xffff) / 32768.f - 1.f);
      br[u] = ((float)((h >> 16) & 0xffff) / 32768.f - 1.f);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[u], br[(u + i) & 7], acc[i], 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

int32_t bench_mfma_peak(int blocks_per_cu, int nacc, int iters, double* tflops, double* ms_out) {
  const float a0 = nacc < 0 ? -1.f : 1.f;
  nacc = nacc < 0 ? -nacc : nacc;
  float* out = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&out, 4096));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = blocks_per_cu >= 1000 ? blocks_per_cu : 256 * blocks_per_cu;
  auto launch = [&]() {
    if (nacc == 1) hipLaunchKernelGGL(mfma_peak_kernel<1>, dim3(grid), dim3(256), 0, 0, out, iters, a0, 1.f);
    else if (nacc == 2) hipLaunchKernelGGL(mfma_peak_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters, a0, 1.f);
    else hipLaunchKernelGGL(mfma_peak_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, a0, 1.f);
  };
  launch();
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const int na = nacc >= 4 ? 4 : nacc;
  double flops = (double)grid * 4 /*waves*/ * (double)iters * 8 * na * 4096.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  *ms_out = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return WETTS_OK;
}


// ------------------------------------------------------------------------------------------
// f32 matrix pipe + f32 vector pipe side by side
// ------------------------------------------------------------------------------------------
#define WETTS_VFMA(c, a, b) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b))

template <int NV>
__global__ __launch_bounds__(256) void mfma_valu_interleaved_kernel(float* out, int iters, float a0) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float va[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) va[u] = 0.f;
  const float a = a0 + (threadIdx.x & 7) * 1e-3f, b = 1.f + (threadIdx.x & 3) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NV; ++v) WETTS_VFMA(va[(v + i) & 7], a, b);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
  for (int u = 0; u < 8; ++u) s += va[u];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// waves 0..3: MFMA only; waves 4..7: VALU only (nv FMAs per MFMA of the sibling wave)
__global__ __launch_bounds__(512) void mfma_valu_split_kernel(float* out, int iters, int nv, float a0) {
  const int wave = threadIdx.x >> 6;
  const float a = a0 + (threadIdx.x & 7) * 1e-3f, b = 1.f + (threadIdx.x & 3) * 1e-3f;
  float s = 0.f;
  if (wave < 4) {
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    float va[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) va[u] = 0.f;
    const int n = iters * 2 * nv;  // 32 MFMAs per iteration on the sibling; 16 FMAs per inner pass
    for (int it = 0; it < n; ++it) {
#pragma unroll
      for (int u = 0; u < 16; ++u) WETTS_VFMA(va[u], a, b);
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) s += va[u];
  }
  if (s == 12345.678f) out[threadIdx.x] = s;
}

// ------------------------------------------------------------------------------------------
// what does each ingredient of the conv inner loop cost the f32 MFMA stream?
// MODE bits: 1 = B fragments from LDS (else constants in registers), 2 = A fragments streamed from
// global / L2 (1 KB per wave per 16 MFMAs, else constants), 4 = __syncthreads() every 6 groups,
// 8 = B reads software-pipelined one k-step ahead with sched_barriers (else: as the compiler places them),
// 16 = 64 VALU instructions (v_fma) per group of 16 MFMAs (a stand-in for staging / epilogue ALU work)
// ------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void mfma_loop_kernel(const float4* __restrict__ A, float* out,
                                                        int groups, int iters, int Wp) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * Wp; i += 256) sm[i] = (float)(i & 255) * 1e-3f;
  __syncthreads();
  const float* bcol = sm + (size_t)(lane >> 5) * Wp + wave * 128 + (lane & 31);
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const float4* ab = A + lane;
  float va[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) va[u] = 0.f;
  const float c0 = 1.f + (tid & 7) * 1e-3f;
  for (int it = 0; it < iters; ++it) {
    float4 aa[2];
    aa[0] = (MODE & 2) ? ab[0] : make_float4(c0, c0 + 1e-3f, c0 + 2e-3f, c0 + 3e-3f);
    aa[1] = (MODE & 2) ? ab[64] : aa[0];
    float bv[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bv[0][j] = (MODE & 1) ? bcol[32 * j] : 0.5f + 1e-3f * j;
    auto group = [&](float4& areg, const float4* anext, const float* cur, const float* nxt) {
      const float4 av = areg;
      if (MODE & 2) areg = *anext;
      if (MODE & 8) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float a = s == 0 ? av.x : s == 1 ? av.y : s == 2 ? av.z : av.w;
        if (MODE & 8) {
          const float* src = s < 3 ? cur + (size_t)(2 * (s + 1)) * Wp : nxt;
          if (MODE & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[(s + 1) & 1][j] = src[32 * j];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[(MODE & 1) ? (s & 1) : 0][j], acc[j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          const float* src = cur + (size_t)(2 * s) * Wp;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float b = (MODE & 1) ? src[32 * j] : bv[0][j];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
          }
        }
        if (MODE & 16) {
#pragma unroll
          for (int u = 0; u < 16; ++u) WETTS_VFMA(va[u & 7], a, c0);
        }
      }
    };
    for (int g = 0; g < groups; g += 2) {
      const int tap = (g >> 1) % 3, chunk = (g >> 1) / 3;
      const float* r0 = bcol + (size_t)((chunk & 1) * 16) * Wp + tap;
      const float* r1 = r0 + (size_t)8 * Wp;
      group(aa[0], ab + (int64_t)(g + 2) * 64, r0, r1);
      group(aa[1], ab + (int64_t)(g + 3) * 64, r1, r0 + 1);
      if ((MODE & 4) && (g % 6) == 4) __syncthreads();
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[j][r];
#pragma unroll
  for (int u = 0; u < 8; ++u) s += va[u];
  if (s == 12345.678f) out[tid] = s;
}

// Second decomposition: wave tile shape and operand paths.
//   MB x NB 32x32 accumulators per wave; BM: 0 constants, 1 ds_read_b32 from [ch][col] rows (today's
//   layout), 2 ds_read_b128 from a [col][16 ch + 4 pad] image (4 k-steps per read); AM: 0 constants,
//   1 global_load_dwordx4 per m-block per group (today), 2 ds_read_b128 from LDS (A staged by someone
//   else).  B reads are pipelined one k-step (BM 1) / one group (BM 2) ahead.
template <int MB, int NB, int BM, int AM>
__global__ __launch_bounds__(256) void mfma_loop2_kernel(const float4* __restrict__ A, float* out,
                                                         int groups, int iters) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int Wp = 516;            // BM 1 row stride (floats)
  constexpr int CS = 20;             // BM 2 column stride (floats): 16 channels + 4 pad
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 32 * Wp; i += 256) sm[i] = (float)(i & 255) * 1e-3f;
  __syncthreads();
  const int half = lane >> 5, l31 = lane & 31;
  const float* b1 = sm + (size_t)half * Wp + (wave & 1) * (32 * NB) + l31;
  const float* b2 = sm + (size_t)((wave & 1) * (32 * NB) + l31) * CS + half * 4;
  const float4* alds = reinterpret_cast<const float4*>(sm) + lane;
  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const float4* ab = A + lane;
  const float c0 = 1.f + (tid & 7) * 1e-3f;
  auto loadA = [&](int g, int i) -> float4 {
    if (AM == 1) return ab[(int64_t)(g * MB + i) * 64];
    if (AM == 2) return alds[((g * MB + i) & 31) * 64];
    return make_float4(c0, c0 + 1e-3f, c0 + 2e-3f, c0 + 3e-3f);
  };
  for (int it = 0; it < iters; ++it) {
    float4 aa[2][MB];
#pragma unroll
    for (int i = 0; i < MB; ++i) { aa[0][i] = loadA(0, i); aa[1][i] = loadA(1, i); }
    if (BM == 2) {
      float4 bq[2][NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) bq[0][j] = *reinterpret_cast<const float4*>(b2 + (size_t)(32 * j) * CS);
      auto group = [&](int par, int g) {
        float4 av[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) { av[i] = aa[par][i]; aa[par][i] = loadA(g + 2, i); }
        const float* nx = b2 + ((g + 1) % 3) * CS + (((g + 1) / 3) & 1) * 8;
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[par ^ 1][j] = *reinterpret_cast<const float4*>(nx + (size_t)(32 * j) * CS);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            const float a = s == 0 ? av[i].x : s == 1 ? av[i].y : s == 2 ? av[i].z : av[i].w;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
              const float4 q = bq[par][j];
              const float b = s == 0 ? q.x : s == 1 ? q.y : s == 2 ? q.z : q.w;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i][j], 0, 0, 0);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      for (int g = 0; g < groups; g += 2) { group(0, g); group(1, g + 1); }
    } else {
      float bv[2][NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) bv[0][j] = BM ? b1[32 * j] : 0.5f + 1e-3f * j;
      auto group = [&](int par, int g) {
        float4 av[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) { av[i] = aa[par][i]; aa[par][i] = loadA(g + 2, i); }
        const float* cur = b1 + (size_t)((g & 1) * 8 + ((g >> 1) & 1) * 16) * Wp + (g % 3);
        const float* nxt = b1 + (size_t)(((g + 1) & 1) * 8) * Wp + ((g + 1) % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (BM) {
            const float* src = s < 3 ? cur + (size_t)(2 * (s + 1)) * Wp : nxt;
#pragma unroll
            for (int j = 0; j < NB; ++j) bv[(s + 1) & 1][j] = src[32 * j];
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            const float a = s == 0 ? av[i].x : s == 1 ? av[i].y : s == 2 ? av[i].z : av[i].w;
#pragma unroll
            for (int j = 0; j < NB; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[BM ? (s & 1) : 0][j], acc[i][j], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      for (int g = 0; g < groups; g += 2) { group(0, g); group(1, g + 1); }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (s == 12345.678f) out[tid] = s;
}

// The 16-bit conv inner loop taken apart (round 3): v_mfma_f32_32x32x16_bf16 with MB x NB accumulators per wave,
// KS = 4 k-steps per (chunk, tap) group.  BM: 0 B fragments in registers, 1 ds_read_b128 per fragment from a
// [column][rs bytes] image (the pair kernel's addresses: tap shift in rows, chunk / k-step in bytes), pipelined one
// k-step ahead.  AM: 0 A in registers, 1 one global_load_dwordx4 per (m-block, k-step) from an L2-resident packed
// stream, ring of two groups (the pair kernel's NR = 2).  SYNC: a block barrier every 22 groups (one conv of a
// C = 128, k = 11 pair).  Operands are pseudo-random bf16 in +-[1, 2) (toggle rate of real data; the accumulators
// random-walk).  clk[2 bid], clk[2 bid + 1]: shader cycles (s_memtime) and 100 MHz ticks of the block's loop.
template <int MB, int NB, int BM, int AM, int SYNC>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(MB * NB >= 8 ? 2 : (MB * NB >= 4 ? 3 : 4), MB * NB >= 8 ? 2 : (MB * NB >= 4 ? 3 : 4))))
void mfma16_loop_kernel(const uint4* __restrict__ A, float* out,
                                                          unsigned long long* clk, int groups, int iters, int rs,
                                                          int lds_words) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
  constexpr int KS = 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned* smw = reinterpret_cast<unsigned*>(smb);
  for (int i = tid; i < lds_words; i += 256) {
    unsigned h = (unsigned)i * 2654435761u + 77u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    smw[i] = 0x3f803f80u | (h & 0x807f807fu);
  }
  __syncthreads();
  const int half = lane >> 5, l31 = lane & 31;
  const unsigned char* bcol = smb + (size_t)((wave & 1) * (32 * NB) + l31) * rs + half * 16;
  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const uint4* ab = A + lane;
  auto rnd4 = [&](unsigned k) {
    unsigned h = (unsigned)(tid * 131 + k) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const unsigned w = 0x3f803f80u | (h & 0x807f807fu);
    return make_uint4(w, w ^ 0x00110022u, w ^ 0x80000033u, w ^ 0x00448000u);
  };
  auto loadA = [&](int g, int i, int s) -> uint4 {
    if (AM == 1) return ab[((int64_t)(g * KS + s) * MB + i) * 64];
    return rnd4(i * KS + s);
  };
  unsigned long long t0 = 0, w0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); w0 = wall_clock64(); }
  for (int it = 0; it < iters; ++it) {
    uint4 aa[2][MB][KS];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) { aa[0][i][s2] = loadA(0, i, s2); aa[1][i][s2] = aa[0][i][s2]; }
    uint4 bq[2][NB];
    auto bpos = [&](int g) { return bcol + (size_t)(g % 3) * rs + ((g / 3) & 1) * 128; };
    auto b_load = [&](uint4 (&dst)[NB], const unsigned char* bb, int s2) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (BM) dst[j] = *reinterpret_cast<const uint4*>(bb + (size_t)(32 * j) * rs + s2 * 32);
        else dst[j] = rnd4(100 + j);
      }
    };
    b_load(bq[0], bpos(0), 0);
    auto group = [&](int par, int g) {
      const int gn = g + 1 < groups ? g + 1 : groups - 1;  // clamped: unconditional prefetch
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) aa[par ^ 1][i][s2] = loadA(gn, i, s2);
      __builtin_amdgcn_sched_barrier(0);
      const unsigned char* cur = bpos(g);
      const unsigned char* nxt = bpos(gn);
#pragma unroll
      for (int s2 = 0; s2 < KS; ++s2) {
        if (BM) {
          if (s2 + 1 < KS) b_load(bq[(s2 + 1) & 1], cur, s2 + 1);
          else b_load(bq[(s2 + 1) & 1], nxt, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int i = 0; i < MB; ++i)
            acc[i][j] = mfma16<false>(aa[par][i][s2], bq[BM ? (s2 & 1) : 0][j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int g = 0; g < groups; g += 2) {
      group(0, g);
      group(1, g + 1);
      if (SYNC && (g % 22) == 20) __syncthreads();
    }
  }
  if (tid == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
  if (sum == 12345.678f) out[tid] = sum;
}
__global__ void fill_bf16_words_kernel(unsigned* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned h = (unsigned)i * 2654435761u + 5u;
  h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  p[i] = 0x3f803f80u | (h & 0x807f807fu);
}
}  // namespace wetts

extern "C" {

// cfg = MB*10000 + NB*1000 + BM*100 + AM*10 + SYNC (see mfma16_loop_kernel).  blocks_per_cu resident 4-wave blocks
// (LDS-limited); rs = LDS row stride in bytes (272: C = 128, 144: C = 64).
int32_t wetts_bench_mfma16_loop(int32_t cfg, int32_t blocks_per_cu, int32_t rs, int32_t groups, int32_t iters,
                                double* tflops, double* ms_out, double* mhz) {
  WETTS_REQUIRE(tflops && ms_out && mhz && groups > 0 && (groups & 1) == 0 && iters > 0, "bad argument");
  WETTS_REQUIRE(blocks_per_cu >= 1 && blocks_per_cu <= 8 && (rs == 272 || rs == 144), "bad argument");
  const int MBv = cfg / 10000, NBv = (cfg / 1000) % 10;
  uint4* A = nullptr;
  float* out = nullptr;
  unsigned long long* clk = nullptr;
  const size_t abytes = (size_t)(groups + 2) * 4 * MBv * 64 * sizeof(uint4);
  WETTS_HIP_CHECK(hipMalloc((void**)&A, abytes));
  hipLaunchKernelGGL(fill_bf16_words_kernel, dim3((unsigned)((abytes / 4 + 255) / 256)), dim3(256), 0, 0,
                     reinterpret_cast<unsigned*>(A), (int64_t)(abytes / 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&out, 4096));
  const int grid = 256 * blocks_per_cu;
  WETTS_HIP_CHECK(hipMalloc((void**)&clk, (size_t)grid * 2 * sizeof(unsigned long long)));
  const size_t need = (size_t)(2 * 32 * NBv + 4) * rs + 256;
  size_t lds = (size_t)(160 * 1024 / blocks_per_cu) & ~(size_t)1023;   // exactly blocks_per_cu fit
  if (blocks_per_cu < 8) {
    const size_t lo = (size_t)(160 * 1024 / (blocks_per_cu + 1)) + 1024; // ... and not one more
    if (lds < lo) lds = lo;
  }
  WETTS_REQUIRE(lds >= need, "tile of %zu bytes does not fit %d blocks per CU", need, blocks_per_cu);
  bool ok = true;
#define WETTS_L16(MB_, NB_, BM_, AM_, SY_)                                                                  \
  if (cfg == MB_ * 10000 + NB_ * 1000 + BM_ * 100 + AM_ * 10 + SY_) {                                       \
    (void)hipFuncSetAttribute((const void*)mfma16_loop_kernel<MB_, NB_, BM_, AM_, SY_>,                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                       \
    hipLaunchKernelGGL((mfma16_loop_kernel<MB_, NB_, BM_, AM_, SY_>), dim3(grid), dim3(256), lds, 0, A, out, \
                       clk, groups, iters, rs, (int)(need / 4));                                             \
    return;                                                                                                  \
  }
  auto launch = [&]() {
    WETTS_L16(1, 4, 0, 0, 0) WETTS_L16(1, 4, 1, 0, 0) WETTS_L16(1, 4, 0, 1, 0) WETTS_L16(1, 4, 1, 1, 0)
    WETTS_L16(1, 4, 1, 1, 1) WETTS_L16(2, 4, 0, 0, 0) WETTS_L16(2, 4, 1, 0, 0) WETTS_L16(2, 4, 1, 1, 0)
    WETTS_L16(2, 4, 1, 1, 1) WETTS_L16(2, 2, 0, 0, 0) WETTS_L16(2, 2, 1, 0, 0) WETTS_L16(2, 2, 1, 1, 0)
    WETTS_L16(2, 2, 1, 1, 1) WETTS_L16(1, 2, 1, 1, 0) WETTS_L16(1, 8, 1, 1, 0) WETTS_L16(4, 1, 1, 1, 0)
    WETTS_L16(2, 3, 1, 1, 0) WETTS_L16(4, 2, 1, 1, 0)
    ok = false;
  };
  launch();
  WETTS_REQUIRE(ok, "unknown mfma16 loop configuration %d", cfg);
  WETTS_LAUNCH_CHECK();
  WETTS_HIP_CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)grid * 2);
  WETTS_HIP_CHECK(hipMemcpy(h.data(), clk, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (int b = 0; b < grid; ++b) { cyc += (double)h[2 * b]; wall += (double)h[2 * b + 1]; }
  *mhz = wall > 0 ? cyc / wall * 100.0 : 0.0;
  *tflops = (double)grid * 4 * (double)iters * groups * 4 * MBv * NBv * 32768.0 / (ms * 1e-3) / 1e12;
  *ms_out = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A);
  (void)hipFree(out);
  (void)hipFree(clk);
  return WETTS_OK;
}

// cfg = MB*1000 + NB*100 + BM*10 + AM
int32_t wetts_bench_mfma_loop2(int32_t cfg, int32_t lds_kb, int32_t groups, int32_t iters,
                               double* tflops, double* ms_out) {
  WETTS_REQUIRE(tflops && ms_out && groups > 0 && (groups & 1) == 0 && iters > 0, "bad argument");
  float4* A = nullptr;
  float* out = nullptr;
  const size_t abytes = (size_t)(groups + 4) * 2 * 64 * sizeof(float4);
  WETTS_HIP_CHECK(hipMalloc((void**)&A, abytes));
  WETTS_HIP_CHECK(hipMemset(A, 0, abytes));
  WETTS_HIP_CHECK(hipMalloc((void**)&out, 4096));
  const size_t lds = (size_t)lds_kb * 1024 > (size_t)32 * 516 * 4 ? (size_t)lds_kb * 1024 : (size_t)32 * 516 * 4;
  int per_cu = (int)(160 * 1024 / lds);
  per_cu = per_cu < 1 ? 1 : (per_cu > 4 ? 4 : per_cu);
  const int grid = 256 * per_cu;
  const int MBv = cfg / 1000, NBv = (cfg / 100) % 10;
  bool ok = true;
#define WETTS_L2(MB_, NB_, BM_, AM_)                                                                  \
  if (cfg == MB_ * 1000 + NB_ * 100 + BM_ * 10 + AM_) {                                               \
    (void)hipFuncSetAttribute((const void*)mfma_loop2_kernel<MB_, NB_, BM_, AM_>,                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                 \
    hipLaunchKernelGGL((mfma_loop2_kernel<MB_, NB_, BM_, AM_>), dim3(grid), dim3(256), lds, 0, A, out, \
                       groups, iters);                                                                 \
    return;                                                                                            \
  }
  auto launch = [&]() {
    WETTS_L2(1, 4, 0, 0) WETTS_L2(1, 4, 1, 0) WETTS_L2(1, 4, 2, 0) WETTS_L2(1, 4, 1, 1) WETTS_L2(1, 4, 2, 1)
    WETTS_L2(1, 4, 1, 2) WETTS_L2(1, 4, 2, 2) WETTS_L2(2, 4, 0, 0) WETTS_L2(2, 4, 1, 0) WETTS_L2(2, 4, 2, 0)
    WETTS_L2(2, 4, 1, 1) WETTS_L2(2, 4, 2, 1) WETTS_L2(2, 4, 2, 2) WETTS_L2(2, 2, 2, 1) WETTS_L2(2, 2, 1, 1)
    WETTS_L2(1, 8, 2, 1) WETTS_L2(1, 8, 1, 1) WETTS_L2(1, 8, 0, 0)
    ok = false;
  };
  launch();
  WETTS_REQUIRE(ok, "unknown loop2 configuration %d", cfg);
  WETTS_LAUNCH_CHECK();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *tflops = (double)grid * 4 * (double)iters * groups * 16 * MBv * (NBv / 4.0) * 4096.0 / (ms * 1e-3) / 1e12;
  *ms_out = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A);
  (void)hipFree(out);
  return WETTS_OK;
}

int32_t wetts_bench_mfma_loop(int32_t mode, int32_t lds_kb, int32_t groups, int32_t iters,
                              double* tflops, double* ms_out) {
  WETTS_REQUIRE(tflops && ms_out && groups > 0 && (groups & 1) == 0 && iters > 0, "bad argument");
  float4* A = nullptr;
  float* out = nullptr;
  const size_t abytes = (size_t)(groups + 4) * 64 * sizeof(float4);
  WETTS_HIP_CHECK(hipMalloc((void**)&A, abytes));
  WETTS_HIP_CHECK(hipMemset(A, 0, abytes));
  WETTS_HIP_CHECK(hipMalloc((void**)&out, 4096));
  const int Wp = 516;
  const size_t lds = (size_t)lds_kb * 1024 > (size_t)32 * Wp * 4 ? (size_t)lds_kb * 1024 : (size_t)32 * Wp * 4;
  const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
  const int grid = 256 * (per_cu > 4 ? 4 : per_cu);  // one resident wave of blocks
#define WETTS_LOOP_CASE(M)                                                                          \
  case M:                                                                                           \
    (void)hipFuncSetAttribute((const void*)mfma_loop_kernel<M>,                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);              \
    hipLaunchKernelGGL(mfma_loop_kernel<M>, dim3(grid), dim3(256), lds, 0, A, out, groups, iters, Wp); \
    break;
  auto launch = [&]() {
    switch (mode) {
      WETTS_LOOP_CASE(0) WETTS_LOOP_CASE(1) WETTS_LOOP_CASE(9) WETTS_LOOP_CASE(2) WETTS_LOOP_CASE(3)
      WETTS_LOOP_CASE(11) WETTS_LOOP_CASE(15) WETTS_LOOP_CASE(7) WETTS_LOOP_CASE(16) WETTS_LOOP_CASE(27)
      WETTS_LOOP_CASE(4) WETTS_LOOP_CASE(13)
      default: break;
    }
  };
  launch();
  WETTS_LAUNCH_CHECK();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *tflops = (double)grid * 4 * (double)iters * groups * 16 * 4096.0 / (ms * 1e-3) / 1e12;
  *ms_out = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A);
  (void)hipFree(out);
  return WETTS_OK;
}

int32_t wetts_bench_mfma_valu(int32_t mode, int32_t nv, int32_t iters, double* tflops_mfma,
                              double* tflops_valu, double* ms_out) {
  WETTS_REQUIRE(tflops_mfma && tflops_valu && ms_out && iters > 0 && nv >= 0, "bad argument");
  float* out = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&out, 4096));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = 256 * 2;
  auto launch = [&]() {
    if (mode == 1) {
      hipLaunchKernelGGL(mfma_valu_split_kernel, dim3(grid), dim3(512), 0, 0, out, iters, nv, 1.f);
      return;
    }
    switch (nv) {
      case 0: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<0>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
      case 2: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
      case 4: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
      case 8: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<8>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
      case 12: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<12>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
      default: hipLaunchKernelGGL(mfma_valu_interleaved_kernel<16>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f); break;
    }
  };
  launch();
  (void)hipEventRecord(e0, 0);
  launch();
  (void)hipEventRecord(e1, 0);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_mfma = (double)grid * 4 * (double)iters * 32;  // per MFMA wave: 32 per iteration
  const int nvi = (mode == 1 || nv == 0 || nv == 2 || nv == 4 || nv == 8 || nv == 12) ? nv : 16;
  *tflops_mfma = n_mfma * 4096.0 / (ms * 1e-3) / 1e12;
  *tflops_valu = n_mfma * nvi * 128.0 / (ms * 1e-3) / 1e12;
  *ms_out = ms;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(out);
  return WETTS_OK;
}

// ResBlock1 (k, dilations 1/3/5 in front of c1, c2 at 1) on [B,C,T], npairs = 1..3 pairs, at f32.
// mode 0: conv by conv (2 launches per pair); 1: resblock_pair32_kernel per pair; 2: the chain kernel
// per pair; 3: the chain kernel once for all pairs.  flags: 4 accumulate into out, 8 divide by 3.
int32_t wetts_bench_resblock(int32_t C, int32_t k, int32_t npairs, int32_t first_dil, int32_t B, int32_t T,
                             int32_t flags, int32_t mode, int32_t iters, double* ms_out,
                             double* checksum_out) {
  WETTS_REQUIRE(ms_out && C > 0 && k > 0 && npairs >= 1 && npairs <= 3 && B > 0 && T > 0 && iters > 0,
                "bad argument");
  hipStream_t s = nullptr;
  const int64_t n = (int64_t)B * C * T, nw = (int64_t)C * C * k;
  float *x = nullptr, *o = nullptr, *r = nullptr, *ft = nullptr, *pp[2] = {nullptr, nullptr};
  float *w[6] = {}, *bias[6] = {};
  WETTS_HIP_CHECK(hipMalloc((void**)&x, n * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&o, n * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&r, n * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&ft, n * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&pp[0], n * 4));
  WETTS_HIP_CHECK(hipMalloc((void**)&pp[1], n * 4));
  WETTS_TRY(fill_pseudo(x, n, 1, 1.f, s));
  WETTS_TRY(fill_pseudo(r, n, 2, 1.f, s));
  WETTS_HIP_CHECK(hipMemsetAsync(o, 0, n * 4, s));
  PackedConv c1[3], c2[3];
  const int dils[3] = {first_dil, first_dil == 1 ? 3 : first_dil, first_dil == 1 ? 5 : first_dil};
  for (int i = 0; i < 2 * npairs; ++i) {
    WETTS_HIP_CHECK(hipMalloc((void**)&w[i], nw * 4));
    WETTS_HIP_CHECK(hipMalloc((void**)&bias[i], (int64_t)C * 4));
    WETTS_TRY(fill_pseudo(w[i], nw, 10 + i, 0.5f / sqrtf((float)C * k), s));
    WETTS_TRY(fill_pseudo(bias[i], C, 20 + i, 0.1f, s));
    const int d = (i & 1) ? 1 : dils[i >> 1];
    WETTS_TRY(pack_conv_weight(w[i], bias[i], C, C, k, d, (k - 1) / 2 * d, 0, 0, s,
                               (i & 1) ? &c2[i >> 1] : &c1[i >> 1]));
  }
  const int accum = (flags & 4) ? 1 : 0;
  const float odiv = (flags & 8) ? 3.f : 1.f;
  auto run = [&]() -> int32_t {
    if (mode == 3) {
      ResChain32Params cp;
      memset(&cp, 0, sizeof(cp));
      cp.x = x; cp.out = o; cp.T = T; cp.B = B; cp.accum = accum; cp.out_div = odiv; cp.slope = 0.1f;
      return launch_resblock_chain32(c1, c2, npairs, cp, s);
    }
    const float* cur = x;
    for (int pr = 0; pr < npairs; ++pr) {
      const bool last = pr == npairs - 1;
      float* dst = last ? o : pp[pr & 1];
      if (mode == 2) {
        ResChain32Params cp;
        memset(&cp, 0, sizeof(cp));
        cp.x = cur; cp.out = dst; cp.T = T; cp.B = B; cp.accum = last ? accum : 0;
        cp.out_div = last ? odiv : 1.f; cp.slope = 0.1f;
        WETTS_TRY(launch_resblock_chain32(&c1[pr], &c2[pr], 1, cp, s));
      } else if (mode == 1) {
        ResPair32Params q;
        memset(&q, 0, sizeof(q));
        q.x = cur; q.out = dst; q.T = T; q.B = B; q.accum = last ? accum : 0;
        q.out_div = last ? odiv : 1.f; q.slope = 0.1f;
        WETTS_TRY(launch_resblock_pair32(c1[pr], c2[pr], q, s));
      } else {
        ConvParams p1 = conv_io(cur, C, T, ft, C, B);
        p1.in_act = IN_LRELU; p1.in_slope = 0.1f; p1.tag = 1;
        WETTS_TRY(launch_conv(c1[pr], p1, s));
        ConvParams p2 = conv_io(ft, C, T, dst, C, B);
        p2.in_act = IN_LRELU; p2.in_slope = 0.1f; p2.tag = 1;
        p2.res = cur; p2.r_bs = (int64_t)C * T; p2.r_cs = T;
        p2.accum = last ? accum : 0;
        p2.out_div = last ? odiv : 1.f;
        WETTS_TRY(launch_conv(c2[pr], p2, s));
      }
      cur = dst;
    }
    return WETTS_OK;
  };
  int32_t rc = WETTS_OK;
  for (int i = 0; i < 3 && rc == WETTS_OK; ++i) rc = run();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters && rc == WETTS_OK; ++i) rc = run();
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *ms_out = ms / iters;
  if (checksum_out && rc == WETTS_OK) {  // full-tensor hash of ONE application on a known `out`
    (void)hipMemcpyAsync(o, r, n * 4, hipMemcpyDeviceToDevice, s);
    rc = run();
    std::vector<uint32_t> host((size_t)n);
    (void)hipMemcpy(host.data(), o, host.size() * 4, hipMemcpyDeviceToHost);
    uint64_t hsh = 1469598103934665603ull;
    for (size_t i = 0; i < host.size(); ++i) hsh = (hsh ^ host[i]) * 1099511628211ull;
    *checksum_out = (double)(hsh >> 12);
    // numeric comparison with the previous call's output (diagnostics: where do two forms differ?)
    static std::vector<uint32_t> prev;
    if (prev.size() == host.size() && getenv("WETTS_BENCH_DIFF")) {
      double mx = 0, mxref = 0;
      int64_t nbad = 0, first = -1, last = -1;
      for (size_t i = 0; i < host.size(); ++i) {
        float a, b2;
        memcpy(&a, &host[i], 4);
        memcpy(&b2, &prev[i], 4);
        double d = fabs((double)a - (double)b2);
        if (!(d <= mx)) mx = d;
        if (fabs(b2) > mxref) mxref = fabs(b2);
        if (host[i] != prev[i]) { ++nbad; if (first < 0) first = (int64_t)i; last = (int64_t)i; }
      }
      for (int64_t row : {0, 5, 31}) {
        for (int64_t t : {0, 1, 2, 3, 100, 509, 510, 511, 512, 1000}) {
          float a, b2;
          size_t i = (size_t)(row * T + t);
          memcpy(&a, &host[i], 4);
          memcpy(&b2, &prev[i], 4);
          fprintf(stderr, " r%lld t%lld: %.5f / %.5f;", (long long)row, (long long)t, a, b2);
        }
        fprintf(stderr, "\n");
      }
      fprintf(stderr, "[diff vs previous call] max|d| %.3e (ref max %.3e), %lld of %lld differ, first idx %lld "
              "(t=%lld row=%lld) last %lld (t=%lld)\n", mx, mxref, (long long)nbad, (long long)host.size(),
              (long long)first, (long long)(first % T), (long long)(first / T), (long long)last,
              (long long)(last % T));
    }
    prev = host;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int i = 0; i < 2 * npairs; ++i) {
    free_packed((i & 1) ? &c2[i >> 1] : &c1[i >> 1]);
    (void)hipFree(w[i]);
    (void)hipFree(bias[i]);
  }
  (void)hipFree(x); (void)hipFree(o); (void)hipFree(r); (void)hipFree(ft); (void)hipFree(pp[0]);
  (void)hipFree(pp[1]);
  return rc;
}

int32_t wetts_bench_mfma_peak(int32_t blocks_per_cu, int32_t nacc, int32_t iters, double* tflops,
                              double* ms) {
  WETTS_REQUIRE(tflops && ms && blocks_per_cu > 0 && iters > 0, "bad argument");
  return bench_mfma_peak(blocks_per_cu, nacc, iters, tflops, ms);
}

int32_t wetts_set_conv_variant(int32_t v) {
  set_conv_variant(v);
  return WETTS_OK;
}

}  // extern "C"
