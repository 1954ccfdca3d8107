// DDSConv.forward (reference duration_predictors.py:45-57) -- three layers of
//     y = dwconv_k3(x * mask, dilation 3^i);  y = gelu(LN(y));  y = conv1x1(y);  y = gelu(LN(y));  x = x + y
// and the trailing x * mask -- in ONE launch.
//
// The stochastic duration predictor runs this block four times per utterance (once on the text encoding, once per
// ConvFlow), conv by conv that is 48 launches of 4-10 us on [B][192][Tx] tensors of a few kilobytes: 0.3 of the 1.7 ms
// of the B = 1 encoder call (profiles/r03_b1_anatomy.txt: 24 LayerNorm, 12 dwconv and 12 1x1 conv launches).  Nothing
// in it crosses a column except the depthwise taps (reach 1 + 3 + 9 = 13), so a block takes 32 output columns plus a
// 16-column halo on each side -- 64 columns, two 32-column MFMA tiles -- keeps x and the working tensor in LDS
// ([C][64] f32 each, 96 KB at C = 192) and runs all three layers on them; the halo columns go stale by the dilation
// of each layer and are discarded.  Per column the arithmetic is that of the separate kernels: the dwconv's tap
// order, LayerNorm's partial sums (16 channel groups, combined in group order -- layernorm_reg_kernel), erf GELU, and
// the 1x1 conv summed over K in ascending order in one accumulator (conv_mfma_kernel's order).
#include "common.h"
#include "kernels.h"

namespace wetts {

typedef float f32x16d __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float gelu_erf_d(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }

constexpr int kDdsHalo = 13, kDdsCG = 16, kDdsPer = 12;  // reach of the three dilated taps; C <= 16 * 12
// NW = columns a block computes (one or two 32-column MFMA tiles); NW - 26 of them are valid outputs.  NW = 32 redoes
// 5/6 of its work -- and puts a short utterance on eleven CUs instead of two, which is what a B = 1 call is short of.

template <int NW>
__global__ __launch_bounds__(NW * 16) void dds_fused_kernel(const DdsFusedParams p) {
  constexpr int kDdsW = NW, kDdsValid = NW - 2 * kDdsHalo, NTHR = NW * kDdsCG;
  extern __shared__ __attribute__((aligned(16))) float sm_d[];
  const int C = p.C, T = p.T;
  float* xs = sm_d;                    // [C][64]  x
  float* ys = xs + C * kDdsW;          // [C][64]  working tensor
  float* red = ys + C * kDdsW;         // [16][65]
  float* mk = red + kDdsCG * (kDdsW + 1);  // [64] mask, 0 outside the sequence
  float* prm = mk + kDdsW;             // [3 layers][9][C]: dw taps 0..2, dw bias, n1 gamma / beta, n2 gamma / beta, 1x1 bias
  const int tid = threadIdx.x;
  const int col = tid % kDdsW, cg = tid / kDdsW;  // LayerNorm view: NW columns x 16 channel groups
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.y, n0 = blockIdx.x * kDdsValid;
  const int t = n0 - kDdsHalo + col;
  const bool inside = t >= 0 && t < T;
  const float* xg = p.x + (int64_t)b * C * T;

  // every per-channel parameter of the three layers goes to LDS once (27 C floats): the loops below would otherwise
  // wait for an L2 round trip per channel.  All loads of a thread are issued before its first LDS store.
  {
    float xv[kDdsPer];
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) {
      const int c = cg + kDdsCG * j;
      xv[j] = (inside && c < C) ? xg[(int64_t)c * T + t] : 0.f;
    }
    // (round 6) one thread per channel fetches that channel's 27 parameters through UNIFORM base pointers, all 27
    // loads issued back to back.  Before, a thread derived (layer, k, c) from a flat index and picked its source
    // pointer per lane out of the argument arrays: a pointer fetch and a dependent value fetch, one after the other,
    // for each of its ~10 parameters -- the runs of serialised round trips tools/isa_scan.py flagged, and most of the
    // 52 us this launch took at B = 1.
    float pv[27];
    if (tid < C) {
#pragma unroll
      for (int layer = 0; layer < 3; ++layer)
#pragma unroll
        for (int k = 0; k < 9; ++k) {
          const float* src = k < 3 ? p.sep_w[layer] : k == 3 ? p.sep_b[layer] : k == 4 ? p.n1g[layer]
                           : k == 5 ? p.n1b[layer] : k == 6 ? p.n2g[layer] : k == 7 ? p.n2b[layer] : p.bias[layer];
          pv[layer * 9 + k] = src ? src[k < 3 ? tid * 3 + k : tid] : 0.f;
        }
    }
    const float mv = inside ? p.mask[(int64_t)b * T + t] : 0.f;
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) {
      const int c = cg + kDdsCG * j;
      if (c < C) xs[c * kDdsW + col] = xv[j];
    }
    if (tid < C) {
#pragma unroll
      for (int q = 0; q < 27; ++q) prm[q * C + tid] = pv[q];
    }
    if (cg == 0) mk[col] = mv;
  }
  __syncthreads();

  // LayerNorm over the channels of ys's column `col` (+ GELU), layernorm_reg_kernel's arithmetic; the result goes to
  // ys (first norm) or, with the residual added, to xs (second norm)
  auto layer_norm = [&](const float* gamma, const float* beta, bool second, float mkv) {
    float v[kDdsPer];
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) {
      const int c = cg + kDdsCG * j;
      v[j] = c < C ? ys[c * kDdsW + col] : 0.f;
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) sum += (cg + kDdsCG * j < C) ? v[j] : 0.f;
    red[cg * (kDdsW + 1) + col] = sum;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < kDdsCG; ++q) tot += red[q * (kDdsW + 1) + col];
    const float mean = tot / (float)C;
    __syncthreads();
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) {
      const float d = v[j] - mean;
      sq += (cg + kDdsCG * j < C) ? d * d : 0.f;
    }
    red[cg * (kDdsW + 1) + col] = sq;
    __syncthreads();
    tot = 0.f;
#pragma unroll
    for (int q = 0; q < kDdsCG; ++q) tot += red[q * (kDdsW + 1) + col];
    const float rstd = 1.f / sqrtf(tot / (float)C + 1e-5f);
#pragma unroll
    for (int j = 0; j < kDdsPer; ++j) {
      const int c = cg + kDdsCG * j;
      if (c < C) {
        float y = (v[j] - mean) * rstd * gamma[c] + beta[c];  // (LDS)
        y = gelu_erf_d(y);
        if (second) xs[c * kDdsW + col] = (y + xs[c * kDdsW + col]) * mkv;
        else ys[c * kDdsW + col] = y;
      }
    }
    __syncthreads();
  };

  const int mtiles = C >> 5;            // 32-row tiles of the 1x1 conv
  const int G = p.G;                    // groups of 8 input channels
  const bool mma_wave = wave < (NW / 32) * mtiles;
  const int mtile = wave % mtiles, ntile = wave / mtiles;
  int dil = 1;
  for (int i = 0; i < 3; ++i) {
    const float* lp = prm + (size_t)i * 9 * C;
    // the first A fragments of this layer's 1x1 conv are requested now and land during the dwconv / LayerNorm
    constexpr int R = 8;
    float4 ring[R];
    const float4* abase = reinterpret_cast<const float4*>(p.wpk[i]) + ((int64_t)mtile * G) * 64 + lane;
    if (mma_wave) {
#pragma unroll
      for (int j = 0; j < R; ++j) ring[j] = abase[(int64_t)(j < G ? j : G - 1) * 64];
    }
    // ---- depthwise conv, k = 3, on x * mask (dwconv_kernel's tap order; taps outside the sequence are skipped) ----
    {
      int ccs[3];
      float mvs[3];
      bool use[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int cc = col + (j - 1) * dil, tt = t + (j - 1) * dil;
        const bool in_tile = cc >= 0 && cc < kDdsW;  // beyond the tile: only stale halo columns see it
        use[j] = tt >= 0 && tt < T;
        ccs[j] = in_tile ? cc : col;
        mvs[j] = in_tile ? mk[ccs[j]] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < kDdsPer; ++q) {
        const int c = cg + kDdsCG * q;
        if (c < C) {
          float acc = lp[3 * C + c];
#pragma unroll
          for (int j = 0; j < 3; ++j)
            if (use[j]) acc += lp[j * C + c] * (xs[c * kDdsW + ccs[j]] * mvs[j]);
          ys[c * kDdsW + col] = acc;
        }
      }
    }
    __syncthreads();
    layer_norm(lp + 4 * C, lp + 5 * C, false, 1.f);
    // ---- 1x1 conv on the matrix cores: wave w -> rows 32 (w % mtiles), columns 32 (w / mtiles).  This is what
    // bounds the kernel: 2 C^2 x NW flops per layer on ONE CU's f32 matrix pipe (7.7 us at C = 192, NW = 64) ------
    f32x16d acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mma_wave) {
      const float* bcol = ys + half * kDdsW + ntile * 32 + l32;
      for (int g0 = 0; g0 < G; g0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int g = g0 + j;
          const float4 a = ring[j];
          const int gn = g + R < G ? g + R : G - 1;  // unconditional refill (clamped)
          ring[j] = abase[(int64_t)gn * 64];
          if (g < G) {
            const float* brow0 = bcol + (size_t)(8 * g) * kDdsW;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const float bv = brow0[s * 2 * kDdsW];
              const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
              acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
            }
          }
        }
      }
    }
    __syncthreads();  // every wave has finished reading ys
    if (mma_wave) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mtile * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        ys[row * kDdsW + ntile * 32 + l32] = acc[r] + lp[8 * C + row];
      }
    }
    __syncthreads();
    layer_norm(lp + 6 * C, lp + 7 * C, true, i == 2 ? mk[col] : 1.f);
    dil *= 3;
  }
  if (col >= kDdsHalo && col < kDdsHalo + kDdsValid && inside) {
    float* og = p.out + (int64_t)b * C * T;
    for (int c = cg; c < C; c += kDdsCG) og[(int64_t)c * T + t] = xs[c * kDdsW + col];
  }
}

// mode 1: only while the narrow tiles leave CUs idle anyway (a short utterance at B = 1: eleven blocks where two
// 64-column tiles would be two) -- there the one launch is faster than twelve; mode 2: any size up to 1024 wide tiles
// (measured no faster than the launches it replaces; kept for the tests of the 64-column instantiation)
bool dds_fused_supported(int mode, int C, int Cw, int nchunks, int B, int T) {
  if (!(C == Cw && C % 32 == 0 && C >= 32 && C <= kDdsCG * kDdsPer && nchunks * 16 == C)) return false;
  if (mode == 1) return (int64_t)B * cdiv(T, 32 - 2 * kDdsHalo) <= 128;
  return mode == 2 && (int64_t)B * cdiv(T, 64 - 2 * kDdsHalo) <= 1024;
}

template <int NW>
static int32_t launch_dds(const DdsFusedParams& p, hipStream_t s) {
  const size_t lds = ((size_t)2 * p.C * NW + kDdsCG * (NW + 1) + NW + (size_t)27 * p.C) * sizeof(float);
  static signed char opt_in[64] = {};
  if (lds > 64 * 1024 && !lds_opt_in(reinterpret_cast<const void*>(&dds_fused_kernel<NW>), opt_in)) {
    set_error("dds_fused_kernel: the device refused the %zu-byte dynamic LDS opt-in", lds);
    return WETTS_E_HIP;
  }
  hipLaunchKernelGGL(dds_fused_kernel<NW>, dim3(cdiv(p.T, NW - 2 * kDdsHalo), p.B), dim3(NW * kDdsCG), lds, s, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

int32_t k_dds_fused(DdsFusedParams p, hipStream_t s) {
  if (p.B * p.T == 0) return WETTS_OK;
  if ((int64_t)p.B * cdiv(p.T, 32 - 2 * kDdsHalo) <= 128) return launch_dds<32>(p, s);
  return launch_dds<64>(p, s);
}

}  // namespace wetts
