// 16-bit WaveNet layers of the flow's coupling layers (modules.py:60-87, commons.py:98-105) in the
// channel-last 16-bit layout of the 16-bit decoder: the two convs of a layer run on conv_bf16_kernel
// (f32 accumulate), the element-wise steps between them here.  HBM-bound byte movers: one 16-byte
// piece (8 channels of one frame) per thread, arithmetic in f32, one rounding per stored value.
// Opt-in (wetts_set_flow_precision): the f32 path stays the parity-gated default.
#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

template <bool F16>
__global__ void gate_cl16_kernel(const unsigned short* __restrict__ xin, unsigned short* __restrict__ acts,
                                 int64_t total, int H8, int H) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t row = idx / H8;
  const int c = (int)(idx % H8) * 8;
  const uint4 ta = *reinterpret_cast<const uint4*>(xin + row * 2 * H + c);
  const uint4 sa = *reinterpret_cast<const uint4*>(xin + row * 2 * H + H + c);
  const unsigned tw[4] = {ta.x, ta.y, ta.z, ta.w}, sw[4] = {sa.x, sa.y, sa.z, sa.w};
  unsigned o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float a0 = tanhf(lo16<F16>(tw[e])) * (1.f / (1.f + expf(-lo16<F16>(sw[e]))));
    const float a1 = tanhf(hi16<F16>(tw[e])) * (1.f / (1.f + expf(-hi16<F16>(sw[e]))));
    o[e] = pk2<F16>(a0, a1);
  }
  *reinterpret_cast<uint4*>(acts + row * H + c) = make_uint4(o[0], o[1], o[2], o[3]);
}

int32_t k_gate_cl16(const unsigned short* xin, unsigned short* acts, int64_t rows, int H, int f16,
                    hipStream_t s) {
  WETTS_REQUIRE(H % 8 == 0, "gate: channel count must be a multiple of 8");
  const int64_t n = rows * (H / 8);
  if (n == 0) return WETTS_OK;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (f16)
    hipLaunchKernelGGL(gate_cl16_kernel<true>, grid, dim3(256), 0, s, xin, acts, n, H / 8, H);
  else
    hipLaunchKernelGGL(gate_cl16_kernel<false>, grid, dim3(256), 0, s, xin, acts, n, H / 8, H);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

template <bool F16>
__global__ void wn_update_cl16_kernel(const unsigned short* __restrict__ rs, unsigned short* __restrict__ h,
                                      float* __restrict__ skip, const float* __restrict__ mask, int last,
                                      int first, int64_t total, int H8, int H) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t row = idx / H8;
  const int c = (int)(idx % H8) * 8;
  const int RC = last ? H : 2 * H;
  float sk[8];
  if (!last) {
    const uint4 rv = *reinterpret_cast<const uint4*>(rs + row * RC + c);
    const uint4 sv = *reinterpret_cast<const uint4*>(rs + row * RC + H + c);
    const uint4 hv = *reinterpret_cast<const uint4*>(h + row * H + c);
    const float mk = mask[row];
    const unsigned rw[4] = {rv.x, rv.y, rv.z, rv.w}, hw[4] = {hv.x, hv.y, hv.z, hv.w};
    const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w};
    unsigned o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = pk2<F16>((lo16<F16>(hw[e]) + lo16<F16>(rw[e])) * mk, (hi16<F16>(hw[e]) + hi16<F16>(rw[e])) * mk);
      sk[2 * e] = lo16<F16>(sw[e]);
      sk[2 * e + 1] = hi16<F16>(sw[e]);
    }
    *reinterpret_cast<uint4*>(h + row * H + c) = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
    const uint4 sv = *reinterpret_cast<const uint4*>(rs + row * RC + c);
    const unsigned sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      sk[2 * e] = lo16<F16>(sw[e]);
      sk[2 * e + 1] = hi16<F16>(sw[e]);
    }
  }
  float4* sp = reinterpret_cast<float4*>(skip + row * H + c);
  float4 a = make_float4(sk[0], sk[1], sk[2], sk[3]), b = make_float4(sk[4], sk[5], sk[6], sk[7]);
  if (!first) {
    const float4 pa = sp[0], pb = sp[1];
    a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
    b.x += pb.x; b.y += pb.y; b.z += pb.z; b.w += pb.w;
  }
  sp[0] = a;
  sp[1] = b;
}

int32_t k_wn_update_cl16(const unsigned short* rs, unsigned short* h, float* skip, const float* mask,
                         int last, int first, int64_t rows, int H, int f16, hipStream_t s) {
  WETTS_REQUIRE(H % 8 == 0, "wn_update: channel count must be a multiple of 8");
  const int64_t n = rows * (H / 8);
  if (n == 0) return WETTS_OK;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (f16)
    hipLaunchKernelGGL(wn_update_cl16_kernel<true>, grid, dim3(256), 0, s, rs, h, skip, mask, last, first,
                       n, H / 8, H);
  else
    hipLaunchKernelGGL(wn_update_cl16_kernel<false>, grid, dim3(256), 0, s, rs, h, skip, mask, last, first,
                       n, H / 8, H);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// [B][T][C] -> [B][C][T] through a 32 x 33 LDS tile: both sides move 128-byte rows
__global__ __launch_bounds__(256) void cl32_to_cf32_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int C, int T) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* xb = x + (int64_t)b * T * C;
  float* ob = out + (int64_t)b * C * T;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (t < T && c < C) ? xb[(int64_t)t * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, t = t0 + tx;
    if (c < C && t < T) ob[(int64_t)c * T + t] = tile[tx][ty + 8 * i];
  }
}

int32_t k_cl32_to_cf32(const float* x, float* out, int B, int C, int T, hipStream_t s) {
  if ((int64_t)B * C * T == 0) return WETTS_OK;
  hipLaunchKernelGGL(cl32_to_cf32_kernel, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(256), 0, s, x, out, C, T);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
