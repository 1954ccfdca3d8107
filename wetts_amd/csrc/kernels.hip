// Non-GEMM kernels of the VITS infer() path for gfx950.  All are HBM/latency-bound byte movers:
// lanes run along the time axis (stride-1 in [B,C,T]) so every global access is coalesced; small
// reductions over channels use LDS or wave shuffles.  No kernel here is reshaped into a GEMM.
#include "kernels.h"

namespace wetts {

static inline dim3 grid1d(int64_t n, int threads) {
  return dim3((unsigned)((n + threads - 1) / threads));
}

// ---------------------------------------------------------------------------------------------
__global__ void embed_mask_kernel(const int64_t* __restrict__ ids,
                                  const int64_t* __restrict__ lengths,
                                  const float* __restrict__ emb, int n_vocab, int B, int H, int T,
                                  float scale, float* __restrict__ x_out,
                                  float* __restrict__ mask_out, int32_t* __restrict__ status) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * H * T;
  if (idx >= total) return;
  int t = (int)(idx % T);
  int c = (int)((idx / T) % H);
  int b = (int)(idx / ((int64_t)T * H));
  bool valid = (int64_t)t < lengths[b];
  float v = 0.f;
  if (valid) {
    int64_t id = ids[(int64_t)b * T + t];
    // nn.Embedding raises IndexError here (encoders.py:48); the clamp keeps the access in bounds
    // and the status bit lets the host raise
    if (id < 0 || id >= n_vocab) {
      if (status && c == 0) atomicOr(status, WETTS_STATUS_PHONE_ID_RANGE);
      id = id < 0 ? 0 : n_vocab - 1;
    }
    v = emb[id * H + c] * scale;
  }
  x_out[idx] = v;
  if (c == 0) mask_out[(int64_t)b * T + t] = valid ? 1.f : 0.f;
}

int32_t k_embed_mask(const int64_t* ids, const int64_t* lengths, const float* emb, int n_vocab,
                     int B, int H, int T, float* x_out, float* mask_out, int32_t* status,
                     hipStream_t s) {
  int64_t n = (int64_t)B * H * T;
  if (n == 0) return WETTS_OK;
  float scale = (float)sqrt((double)H);  // math.sqrt(hidden_channels), encoders.py:48
  hipLaunchKernelGGL(embed_mask_kernel, grid1d(n, 256), dim3(256), 0, s, ids, lengths, emb,
                     n_vocab, B, H, T, scale, x_out, mask_out, status);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels.  Block = 64 time lanes x 4 channel groups; each thread owns C/4
// channels of one (b,t) column; partial moments are combined through LDS.
__device__ __forceinline__ float gelu_erf(float x) {
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}

__global__ __launch_bounds__(256) void layernorm_kernel(
    const float* __restrict__ a, const float* __restrict__ add, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, const float* __restrict__ mask,
    int gelu, int B, int C, int T, float eps, float* __restrict__ out) {
  // block = 16 time lanes x 16 channel groups: the tensors here are tiny (B*T columns), so the
  // kernel is latency-bound and wants many short threads rather than few long ones
  constexpr int TL = 16, CG = 16;
  __shared__ float red[CG][TL + 1];
  const int tl = threadIdx.x % TL, cg = threadIdx.x / TL;
  const int tblocks = (T + TL - 1) / TL;
  const int b = blockIdx.x / tblocks;
  const int t = (blockIdx.x % tblocks) * TL + tl;
  const bool ok = t < T;
  const int64_t base = (int64_t)b * C * T + t;
  const int c0 = (C * cg) / CG, c1 = (C * (cg + 1)) / CG;

  float sum = 0.f;
  if (ok) {
    for (int c = c0; c < c1; ++c) {
      float v = a[base + (int64_t)c * T];
      if (add) v += add[base + (int64_t)c * T];
      sum += v;
    }
  }
  red[cg][tl] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < CG; ++q) tot += red[q][tl];
  const float mean = tot / (float)C;
  __syncthreads();
  float sq = 0.f;
  if (ok) {
    for (int c = c0; c < c1; ++c) {
      float v = a[base + (int64_t)c * T];
      if (add) v += add[base + (int64_t)c * T];
      float d = v - mean;
      sq += d * d;
    }
  }
  red[cg][tl] = sq;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int q = 0; q < CG; ++q) tot += red[q][tl];
  const float var = tot / (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  if (!ok) return;
  const float mk = mask ? mask[(int64_t)b * T + t] : 1.f;
  for (int c = c0; c < c1; ++c) {
    float v = a[base + (int64_t)c * T];
    if (add) v += add[base + (int64_t)c * T];
    float y = (v - mean) * rstd * gamma[c] + beta[c];
    if (gelu) y = gelu_erf(y);
    if (res) y += res[base + (int64_t)c * T];
    out[base + (int64_t)c * T] = y * mk;
  }
}

// The same block shape with the column held in registers (channel c = cg + 16 j, j < PER): one pass over
// memory with all loads in flight at once instead of three passes of dependent-latency loops -- at B = 1
// this kernel is launched 36 times per utterance and each pass is a round trip to L2
// (profiles/r02_b1_anatomy.txt: 11.5 us per launch before).
template <int PER>
__global__ __launch_bounds__(256) void layernorm_reg_kernel(
    const float* __restrict__ a, const float* __restrict__ add, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ res, const float* __restrict__ mask,
    int gelu, int B, int C, int T, float eps, float* __restrict__ out) {
  constexpr int TL = 16, CG = 16;
  __shared__ float red[CG][TL + 1];
  const int tl = threadIdx.x % TL, cg = threadIdx.x / TL;
  const int tblocks = (T + TL - 1) / TL;
  const int b = blockIdx.x / tblocks;
  const int t = (blockIdx.x % tblocks) * TL + tl;
  const bool ok = t < T;
  const int64_t base = (int64_t)b * C * T + (ok ? t : 0);
  float v[PER], av[PER], rv[PER], gm[PER], bt[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = cg + CG * j;
    const int cc = c < C ? c : 0;  // clamped: every load is issued, the value is dropped below
    v[j] = a[base + (int64_t)cc * T];
    rv[j] = res ? res[base + (int64_t)cc * T] : 0.f;
    gm[j] = gamma[cc];
    bt[j] = beta[cc];
  }
  // the second summand in its own pass behind ONE branch: as `if (add) v[j] += add[..]` inside the loop above, every j
  // became load, load, s_waitcnt vmcnt(0), add -- PER serialised round trips, which is what "all loads in flight" was
  // written to avoid (found in the ISA, round 5; same additions, bit-identical)
  if (add) {
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int c = cg + CG * j;
      av[j] = add[base + (int64_t)(c < C ? c : 0) * T];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) v[j] += av[j];
  }
  const float mk = (mask && ok) ? mask[(int64_t)b * T + t] : 1.f;
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) sum += (cg + CG * j < C) ? v[j] : 0.f;
  red[cg][tl] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < CG; ++q) tot += red[q][tl];
  const float mean = tot / (float)C;
  __syncthreads();
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float d = v[j] - mean;
    sq += (cg + CG * j < C) ? d * d : 0.f;
  }
  red[cg][tl] = sq;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int q = 0; q < CG; ++q) tot += red[q][tl];
  const float rstd = 1.f / sqrtf(tot / (float)C + eps);
  if (!ok) return;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int c = cg + CG * j;
    if (c < C) {
      float y = (v[j] - mean) * rstd * gm[j] + bt[j];
      if (gelu) y = gelu_erf(y);
      out[base + (int64_t)c * T] = (y + rv[j]) * mk;
    }
  }
}

int32_t k_layernorm(const float* a, const float* add, const float* gamma, const float* beta,
                    const float* res, const float* mask, int gelu, int B, int C, int T, float* out,
                    hipStream_t s) {
  if (B * T == 0) return WETTS_OK;
  int blocks = B * cdiv(T, 16);
  if (C <= 512) {
    const dim3 g(blocks), blk(256);
    if (C <= 192)
      hipLaunchKernelGGL(layernorm_reg_kernel<12>, g, blk, 0, s, a, add, gamma, beta, res, mask, gelu, B, C, T,
                         1e-5f, out);
    else if (C <= 256)
      hipLaunchKernelGGL(layernorm_reg_kernel<16>, g, blk, 0, s, a, add, gamma, beta, res, mask, gelu, B, C, T,
                         1e-5f, out);
    else
      hipLaunchKernelGGL(layernorm_reg_kernel<32>, g, blk, 0, s, a, add, gamma, beta, res, mask, gelu, B, C, T,
                         1e-5f, out);
    WETTS_LAUNCH_CHECK();
    return WETTS_OK;
  }
  hipLaunchKernelGGL(layernorm_kernel, dim3(blocks), dim3(256), 0, s, a, add, gamma, beta, res,
                     mask, gelu, B, C, T, 1e-5f, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Tv: the valid length of a row (zero padding starts there); T: the row stride.  Tv < T for the ConvNeXt stack, whose
// rows are padded to a multiple of 4 columns (columns >= Tv hold finite junk that must not leak into column Tv - 1)
__global__ void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                              const float* __restrict__ w, const float* __restrict__ bias, int k,
                              int dil, int B, int C, int T, int Tv, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * C * T;
  if (idx >= total) return;
  int t = (int)(idx % T);
  int c = (int)((idx / T) % C);
  int b = (int)(idx / ((int64_t)T * C));
  const float* xr = x + ((int64_t)b * C + c) * T;
  const float* mr = mask ? mask + (int64_t)b * T : nullptr;  // null: no mask (ConvNeXt dw_conv)
  int pad = (k * dil - dil) / 2;
  float acc = bias[c];
  for (int j = 0; j < k; ++j) {
    int tt = t + j * dil - pad;
    if (tt >= 0 && tt < Tv) acc += w[c * k + j] * (mr ? xr[tt] * mr[tt] : xr[tt]);
  }
  out[idx] = acc;
}

int32_t k_dwconv(const float* x, const float* mask, const float* w, const float* bias, int k,
                 int dil, int B, int C, int T, float* out, hipStream_t s, int Tvalid) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(dwconv_kernel, grid1d(n, 256), dim3(256), 0, s, x, mask, w, bias, k, dil, B,
                     C, T, Tvalid > 0 ? Tvalid : T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// ConvNeXtLayer front half (decoders.py:241-243): y = LayerNorm_C(dw_conv_k3(x)) in ONE pass -- x read once, y written
// once (the two-kernel form is four tensor passes and two launches per layer).  Block = 16 waves over 62 output
// columns: lane l of every wave holds column t0 - 1 + l (lanes 0 / 63 are the halo, their neighbours' taps come by
// DPP wave shifts), wave w owns channels w * PER .. and keeps its PER conv outputs in registers; mean and variance
// are the exact two-pass form, reduced over the 16 waves through LDS.  Rows: stride T, zero padding from column Tv.
template <int PER>
__global__ __launch_bounds__(1024) void convnext_dwln_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ wb, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, int C, int T, int Tv, float eps,
                                                            float* __restrict__ out) {
  __shared__ float red[2][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tblocks = (T + 61) / 62;
  const int b = blockIdx.x / tblocks;
  const int t = (blockIdx.x % tblocks) * 62 - 1 + lane;
  const bool inr = t >= 0 && t < Tv;
  const int c0 = wave * PER;
  const float* xb = x + ((int64_t)b * C + c0) * T + (inr ? t : 0);
  float xv[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) xv[j] = xb[(int64_t)j * T];  // every load issued (clamped column), dropped below
  float y[PER];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float x0 = inr ? xv[j] : 0.f;
    const float xm = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x0), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
    const float xp = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x0), 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
    const int c = c0 + j;
    y[j] = wb[c] + w[c * 3] * xm + w[c * 3 + 1] * x0 + w[c * 3 + 2] * xp;
    sum += y[j];
  }
  red[0][wave][lane] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) tot += red[0][q][lane];
  const float mean = tot / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const float d = y[j] - mean;
    sq += d * d;
  }
  red[1][wave][lane] = sq;
  __syncthreads();
  tot = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) tot += red[1][q][lane];
  const float rstd = 1.f / sqrtf(tot / (float)C + eps);
  if (lane == 0 || lane == 63 || t >= T) return;
  float* ob = out + ((int64_t)b * C + c0) * T + t;
#pragma unroll
  for (int j = 0; j < PER; ++j) ob[(int64_t)j * T] = (y[j] - mean) * rstd * gamma[c0 + j] + beta[c0 + j];
}

// returns false when the shape is not covered (the caller then runs k_dwconv + k_layernorm)
bool k_convnext_dwln(const float* x, const float* w, const float* wb, const float* gamma, const float* beta, int B, int C,
                     int T, int Tvalid, float* out, hipStream_t s, int32_t* rc) {
  *rc = WETTS_OK;
  if (B * T == 0) return true;
  if (C % 16 != 0 || (C / 16 != 32 && C / 16 != 16 && C / 16 != 4)) return false;
  const dim3 grid((unsigned)(B * cdiv(T, 62))), blk(1024);
  if (C / 16 == 32)
    hipLaunchKernelGGL(convnext_dwln_kernel<32>, grid, blk, 0, s, x, w, wb, gamma, beta, C, T, Tvalid, 1e-5f, out);
  else if (C / 16 == 16)
    hipLaunchKernelGGL(convnext_dwln_kernel<16>, grid, blk, 0, s, x, w, wb, gamma, beta, C, T, Tvalid, 1e-5f, out);
  else
    hipLaunchKernelGGL(convnext_dwln_kernel<4>, grid, blk, 0, s, x, w, wb, gamma, beta, C, T, Tvalid, 1e-5f, out);
  if (hipGetLastError() != hipSuccess) {
    set_error("convnext_dwln_kernel launch failed");
    *rc = WETTS_E_HIP;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// one wave per (b, c): lanes stride over K, shuffle-reduce
__global__ __launch_bounds__(256) void cond_linear_kernel(const float* __restrict__ g,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias, int B,
                                                          int Cout, int K,
                                                          float* __restrict__ out) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (wave >= B * Cout) return;
  int c = wave % Cout, b = wave / Cout;
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += W[(int64_t)c * K + k] * g[(int64_t)b * K + k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) out[(int64_t)b * Cout + c] = acc + (bias ? bias[c] : 0.f);
}

int32_t k_cond_linear(const float* g, const float* W, const float* bias, int B, int Cout, int K,
                      float* out, hipStream_t s) {
  int64_t n = (int64_t)B * Cout * 64;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(cond_linear_kernel, grid1d(n, 256), dim3(256), 0, s, g, W, bias, B, Cout, K,
                     out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void gather_rows_kernel(const int64_t* __restrict__ idx, const float* __restrict__ table,
                                   int n_rows, int B, int C, float* __restrict__ out,
                                   int32_t* __restrict__ status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  int b = i / C, c = i % C;
  int64_t r = idx[b];
  if (r < 0 || r >= n_rows) {  // emb_g(sid) raises IndexError in the reference (models.py:239)
    if (status && c == 0) atomicOr(status, WETTS_STATUS_SPEAKER_ID_RANGE);
    r = r < 0 ? 0 : n_rows - 1;
  }
  out[i] = table[r * C + c];
}

int32_t k_gather_rows(const int64_t* idx, const float* table, int n_rows, int B, int C, float* out,
                      int32_t* status, hipStream_t s) {
  if (B * C == 0) return WETTS_OK;
  hipLaunchKernelGGL(gather_rows_kernel, grid1d((int64_t)B * C, 256), dim3(256), 0, s, idx, table,
                     n_rows, B, C, out, status);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void add_bias_b_kernel(float* __restrict__ x, const float* __restrict__ v, int64_t total,
                                  int T) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  x[idx] += v[idx / T];
}

int32_t k_add_bias_b(float* x, const float* v, int B, int C, int T, hipStream_t s) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(add_bias_b_kernel, grid1d(n, 256), dim3(256), 0, s, x, v, n, T);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void scale_kernel(const float* __restrict__ a, float scale, int64_t n,
                             float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = a[idx] * scale;
}

int32_t k_scale(const float* a, float scale, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(scale_kernel, grid1d(n, 256), dim3(256), 0, s, a, scale, n, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void convflow_pre_kernel(const float* __restrict__ z, int ch0,
                                    const float* __restrict__ w, const float* __restrict__ bias,
                                    const float* __restrict__ g, int B, int C, int T,
                                    float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * C * T;
  if (idx >= total) return;
  int t = (int)(idx % T);
  int c = (int)((idx / T) % C);
  int b = (int)(idx / ((int64_t)T * C));
  float x0 = z[((int64_t)b * 2 + ch0) * T + t];
  out[idx] = (w[c] * x0 + bias[c]) + g[idx];
}

int32_t k_convflow_pre(const float* z, int ch0, const float* w, const float* bias, const float* g,
                       int B, int C, int T, float* out, hipStream_t s) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(convflow_pre_kernel, grid1d(n, 256), dim3(256), 0, s, z, ch0, w, bias, g, B,
                     C, T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Rational-quadratic spline, inverse direction, linear tails.  One thread per (b,t).
// Mirrors transforms.py:47-97 (tails) and :100-187 (inverse branch) operation by operation.
constexpr int kMaxBins = 16;

__device__ __forceinline__ float softplus_f(float x) {
  return x > 20.f ? x : log1pf(expf(x));  // F.softplus(beta=1, threshold=20)
}

__global__ void spline_inverse_kernel(float* __restrict__ z, int ch0, int ch1,
                                      const float* __restrict__ h, const float* __restrict__ mask,
                                      int nb, float tail, float div, int B, int T,
                                      int32_t* __restrict__ status) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * T) return;
  const int t = idx % T, b = idx / T;
  const float mk = mask[idx];
  float* z0 = z + ((int64_t)b * 2 + ch0) * T + t;
  float* z1 = z + ((int64_t)b * 2 + ch1) * T + t;
  const float x = *z1;
  const float* hp = h + (int64_t)b * (3 * nb - 1) * T + t;
  float outv = x;
  if (x >= -tail && x <= tail) {
    const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
    const float one_minus = (float)(1.0 - 1e-3 * (double)nb);  // python double, then f32 (transforms.py:126)
    float cw[kMaxBins + 1], chh[kMaxBins + 1];
    // widths: softmax -> min + (1-min*nb)*p -> cumsum -> affine -> pin ends
    {
      float u[kMaxBins];
      float mx = -INFINITY;
      for (int i = 0; i < nb; ++i) {
        u[i] = hp[(int64_t)i * T] / div;
        mx = fmaxf(mx, u[i]);
      }
      float sm = 0.f;
      for (int i = 0; i < nb; ++i) {
        u[i] = expf(u[i] - mx);
        sm += u[i];
      }
      float cum = 0.f;
      cw[0] = -tail;
      for (int i = 0; i < nb; ++i) {
        float wdt = min_w + one_minus * (u[i] / sm);
        cum += wdt;
        cw[i + 1] = (2.f * tail) * cum + (-tail);
      }
      cw[0] = -tail;
      cw[nb] = tail;
    }
    {
      float u[kMaxBins];
      float mx = -INFINITY;
      for (int i = 0; i < nb; ++i) {
        u[i] = hp[(int64_t)(nb + i) * T] / div;
        mx = fmaxf(mx, u[i]);
      }
      float sm = 0.f;
      for (int i = 0; i < nb; ++i) {
        u[i] = expf(u[i] - mx);
        sm += u[i];
      }
      float cum = 0.f;
      chh[0] = -tail;
      for (int i = 0; i < nb; ++i) {
        float hgt = min_h + one_minus * (u[i] / sm);
        cum += hgt;
        chh[i + 1] = (2.f * tail) * cum + (-tail);
      }
      chh[0] = -tail;
      chh[nb] = tail;
    }
    // searchsorted(cumheights, x): last edge += 1e-6; count(x >= edge) - 1
    int bin = -1;
    for (int i = 0; i <= nb; ++i) {
      float e = chh[i];
      if (i == nb) e += 1e-6f;
      bin += (x >= e) ? 1 : 0;
    }
    if (bin < 0) bin = 0;
    if (bin > nb - 1) bin = nb - 1;
    const float in_cw = cw[bin];
    const float in_bw = cw[bin + 1] - cw[bin];
    const float in_ch = chh[bin];
    const float in_h = chh[bin + 1] - chh[bin];
    const float in_delta = in_h / in_bw;
    // derivatives: pad both ends with log(exp(1-min_d)-1)
    const float cst = (float)0.5408847652626036;  // np.log(np.exp(1 - 1e-3) - 1)
    float ud0 = (bin == 0) ? cst : hp[(int64_t)(2 * nb + bin - 1) * T];
    float ud1 = (bin == nb - 1) ? cst : hp[(int64_t)(2 * nb + bin) * T];
    const float d0 = min_d + softplus_f(ud0);
    const float d1 = min_d + softplus_f(ud1);

    const float dx = x - in_ch;
    const float s2 = d0 + d1 - 2.f * in_delta;
    const float qa = dx * s2 + in_h * (in_delta - d0);
    const float qb = in_h * d0 - dx * s2;
    const float qc = -in_delta * dx;
    const float disc = qb * qb - 4.f * qa * qc;
    if (!(disc >= 0.f)) {
      if (status) atomicOr(status, WETTS_STATUS_SPLINE_DOMAIN);
    }
    const float root = (2.f * qc) / (-qb - sqrtf(disc));
    outv = root * in_bw + in_cw;
  }
  *z1 = outv * mk;
  *z0 = (*z0) * mk;
}

int32_t k_spline_inverse(float* z, int ch0, int ch1, const float* h, const float* mask,
                         int num_bins, float tail_bound, float div, int B, int T, int32_t* status,
                         hipStream_t s) {
  WETTS_REQUIRE(num_bins <= kMaxBins, "num_bins %d > %d", num_bins, kMaxBins);
  if (B * T == 0) return WETTS_OK;
  hipLaunchKernelGGL(spline_inverse_kernel, grid1d((int64_t)B * T, 128), dim3(128), 0, s, z, ch0,
                     ch1, h, mask, num_bins, tail_bound, div, B, T, status);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void affine_reverse_kernel(const float* __restrict__ z, int ch,
                                      const float* __restrict__ m, const float* __restrict__ logs,
                                      int pidx, const float* __restrict__ mask, int B, int T,
                                      float* __restrict__ logw) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * T) return;
  int t = idx % T, b = idx / T;
  float x = z[((int64_t)b * 2 + ch) * T + t];
  logw[idx] = (x - m[pidx]) * expf(-logs[pidx]) * mask[idx];
}

int32_t k_affine_reverse(const float* z, int ch, const float* m, const float* logs, int param_idx,
                         const float* mask, int B, int T, float* logw, hipStream_t s) {
  if (B * T == 0) return WETTS_OK;
  hipLaunchKernelGGL(affine_reverse_kernel, grid1d((int64_t)B * T, 256), dim3(256), 0, s, z, ch, m,
                     logs, param_idx, mask, B, T, logw);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ void gate_kernel(const float* __restrict__ a, int B, int H, int T,
                            float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * H * T;
  if (idx >= total) return;
  int64_t ht = (int64_t)H * T;
  int64_t b = idx / ht, r = idx % ht;
  float ta = a[b * 2 * ht + r];
  float sa = a[b * 2 * ht + ht + r];
  out[idx] = tanhf(ta) * (1.f / (1.f + expf(-sa)));
}

int32_t k_gate(const float* a, int B, int H, int T, float* out, hipStream_t s) {
  int64_t n = (int64_t)B * H * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(gate_kernel, grid1d(n, 256), dim3(256), 0, s, a, B, H, T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void wn_update_kernel(const float* __restrict__ rs, float* __restrict__ h,
                                 float* __restrict__ skip, const float* __restrict__ mask, int last,
                                 int first, int B, int H, int T) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * H * T;
  if (idx >= total) return;
  int64_t ht = (int64_t)H * T;
  int64_t b = idx / ht, r = idx % ht;
  int t = (int)(r % T);
  float sk;
  if (!last) {
    float res = rs[b * 2 * ht + r];
    sk = rs[b * 2 * ht + ht + r];
    h[idx] = (h[idx] + res) * mask[b * T + t];
  } else {
    sk = rs[idx];
  }
  skip[idx] = first ? sk : skip[idx] + sk;
}

int32_t k_wn_update(const float* rs, float* h, float* skip, const float* mask, int last, int first,
                    int B, int H, int T, hipStream_t s) {
  int64_t n = (int64_t)B * H * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(wn_update_kernel, grid1d(n, 256), dim3(256), 0, s, rs, h, skip, mask, last,
                     first, B, H, T);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void coupling_flip_kernel(const float* __restrict__ xin, const float* __restrict__ m,
                                     const float* __restrict__ mask, int B, int C, int T,
                                     float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * C * T;
  if (idx >= total) return;
  int t = (int)(idx % T);
  int c = (int)((idx / T) % C);
  int b = (int)(idx / ((int64_t)T * C));
  const int half = C / 2;
  float v = xin[((int64_t)b * C + (C - 1 - c)) * T + t];
  if (c >= half) v = (v - m[((int64_t)b * half + (c - half)) * T + t]) * mask[(int64_t)b * T + t];
  out[idx] = v;
}

int32_t k_coupling_flip(const float* xin, const float* m, const float* mask, int B, int C, int T,
                        float* out, hipStream_t s) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(coupling_flip_kernel, grid1d(n, 256), dim3(256), 0, s, xin, m, mask, B, C, T,
                     out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// conv_post: each thread produces 4 consecutive samples; weights (C*k floats) sit in LDS.
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ w, int k,
                                                             int B, int C, int T,
                                                             float* __restrict__ out,
                                                             const int64_t* __restrict__ lens, int len_mul) {
  extern __shared__ float wsh[];
  for (int i = threadIdx.x; i < C * k; i += blockDim.x) wsh[i] = w[i];
  __syncthreads();
  const int per_b = (T + 3) / 4;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * per_b) return;
  const int b = (int)(idx / per_b);
  const int t0 = (int)(idx % per_b) * 4;
  const int pad = (k - 1) / 2;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* xb = x + (int64_t)b * C * T;
  // ragged batch: utterance b ends at Tb (it is convolved as if alone); the rest of its row reads as 0
  const int Tb = lens ? min((int)lens[b] * len_mul, T) : T;
  if (t0 >= Tb) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (t0 + o < T) out[(int64_t)b * T + t0 + o] = 0.f;
    return;
  }
  // Interior threads of 16-byte-aligned rows (all but the first and last of a sequence): the k + 3 <= 10-sample window of
  // a channel is three aligned 16-byte loads (t0 - 4 .. t0 + 7) instead of ten scalar ones, four channels in flight --
  // the kernel streams 32 channel planes (404 MB at the headline shape) and ran at 1.2 TB/s on the scalar form.  Same
  // products, same order (channel-major, tap-minor) as the general loop below: bit-identical.
  if (k == 7 && (T & 3) == 0 && t0 >= 4 && t0 + 8 <= Tb && (C & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(xb) & 15) == 0) {
    for (int c = 0; c < C; c += 4) {
      float4 v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float4* xr4 = reinterpret_cast<const float4*>(xb + (int64_t)(c + u) * T + (t0 - 4));
        v[u][0] = xr4[0];
        v[u][1] = xr4[1];
        v[u][2] = xr4[2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // win[i] = lrelu(x[t0 - 3 + i]), i = 0 .. 9  (pad = 3)
        const float raw[10] = {v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y,
                               v[u][1].z, v[u][1].w, v[u][2].x, v[u][2].y, v[u][2].z};
        float win[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) win[i] = raw[i] > 0.f ? raw[i] : raw[i] * 0.01f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const float wv = wsh[(c + u) * 7 + j];
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] += wv * win[o + j];
        }
      }
    }
  } else
  for (int c = 0; c < C; ++c) {
    const float* xr = xb + (int64_t)c * T;
    float win[4 + 15];
    for (int i = 0; i < 3 + k; ++i) {
      int tt = t0 - pad + i;
      float v = (tt >= 0 && tt < Tb) ? xr[tt] : 0.f;
      win[i] = v > 0.f ? v : v * 0.01f;  // F.leaky_relu default slope, decoders.py:78
    }
    for (int j = 0; j < k; ++j) {
      float wv = wsh[c * k + j];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] += wv * win[o + j];
    }
  }
#pragma unroll
  for (int o = 0; o < 4; ++o)
    if (t0 + o < T) out[(int64_t)b * T + t0 + o] = (t0 + o < Tb) ? tanhf(acc[o]) : 0.f;
}

// conv_post for launches too small to fill the chip (a streaming window is 15 k samples = 15 blocks of
// the kernel above, each thread walking all C channels: 63 us, profiles/r02_b1_anatomy.txt): 8 channel
// groups x 32 samples per block, the k-tap windows of a thread's C/8 channels loaded up front, partial
// sums combined through LDS in a fixed order.
template <int K>
__global__ __launch_bounds__(256) void conv_post_tanh_small_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ w, int B,
                                                                   int C, int T, float* __restrict__ out) {
  constexpr int TL = 32, CG = 8, MAXC = 8;  // C <= 64
  __shared__ float red[CG][TL + 1];
  const int tl = threadIdx.x % TL, cg = threadIdx.x / TL;
  const int tblocks = (T + TL - 1) / TL;
  const int b = blockIdx.x / tblocks;
  const int t = (blockIdx.x % tblocks) * TL + tl;
  constexpr int pad = (K - 1) / 2;
  const float* xb = x + (int64_t)b * C * T;
  float win[MAXC][K], wv[MAXC][K];
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = cg + CG * j;
    const int cc = c < C ? c : 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const int tt = t - pad + i;
      const bool ok = c < C && tt >= 0 && tt < T;
      win[j][i] = xb[(int64_t)cc * T + (ok ? tt : 0)];
      wv[j][i] = ok ? w[cc * K + i] : 0.f;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j)
#pragma unroll
    for (int i = 0; i < K; ++i) {
      const float v = win[j][i];
      acc += wv[j][i] * (v > 0.f ? v : v * 0.01f);  // F.leaky_relu default slope, decoders.py:78
    }
  red[cg][tl] = acc;
  __syncthreads();
  if (cg == 0 && t < T) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < CG; ++q) tot += red[q][tl];
    out[(int64_t)b * T + t] = tanhf(tot);
  }
}

int32_t k_conv_post_tanh(const float* x, const float* w, int k, int B, int C, int T, float* out,
                         hipStream_t s, const int64_t* lens, int len_mul) {
  WETTS_REQUIRE(k <= 15, "conv_post kernel size %d unsupported", k);
  int64_t n = (int64_t)B * ((T + 3) / 4);
  if (n == 0) return WETTS_OK;
  if (k == 7 && C <= 64 && n <= 64 * 256 && !lens) {  // fewer than 64 blocks of the kernel above
    hipLaunchKernelGGL(conv_post_tanh_small_kernel<7>, dim3((unsigned)(B * cdiv(T, 32))), dim3(256), 0, s, x, w,
                       B, C, T, out);
    WETTS_LAUNCH_CHECK();
    return WETTS_OK;
  }
  hipLaunchKernelGGL(conv_post_tanh_kernel, grid1d(n, 256), dim3(256), (size_t)C * k * 4, s, x, w,
                     k, B, C, T, out, lens, len_mul);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// durations -> lengths: one wave per utterance; inclusive scan with wave shuffles.
__global__ __launch_bounds__(64) void durations_kernel(const float* __restrict__ logw,
                                                       const float* __restrict__ mask,
                                                       float length_scale, int B, int T,
                                                       float* __restrict__ w_ceil,
                                                       float* __restrict__ cum,
                                                       int64_t* __restrict__ y_lengths,
                                                       int32_t* __restrict__ status) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float carry = 0.f;
  for (int t0 = 0; t0 < T; t0 += 64) {
    int t = t0 + lane;
    float wc = 0.f;
    if (t < T) {
      float w = expf(logw[(int64_t)b * T + t]) * mask[(int64_t)b * T + t] * length_scale;
      wc = ceilf(w);
      w_ceil[(int64_t)b * T + t] = wc;
    }
    float v = wc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      float n = __shfl_up(v, off, 64);
      if (lane >= off) v += n;
    }
    v += carry;
    if (t < T) cum[(int64_t)b * T + t] = v;
    carry = __shfl(v, 63, 64);
  }
  if (lane == 0) {
    float tot = carry < 1.f ? 1.f : carry;  // clamp_min(sum, 1)
    // NaN / inf durations (a spline-domain failure upstream, or exp overflow): the float -> int64
    // cast is undefined for them; flag and fall back to the clamp value
    if (!(tot <= 9.0e15f)) {
      if (status) atomicOr(status, WETTS_STATUS_DURATION_NONFINITE);
      tot = 1.f;
    }
    y_lengths[b] = (int64_t)tot;
  }
}

int32_t k_durations_to_lengths(const float* logw, const float* mask, float length_scale, int B,
                               int T, float* w_ceil, float* cum, int64_t* y_lengths,
                               int32_t* status, hipStream_t s) {
  if (B == 0) return WETTS_OK;
  hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(64), 0, s, logw, mask, length_scale, B, T,
                     w_ceil, cum, y_lengths, status);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// frame -> phoneme index map (replaces generate_path's dense 0/1 tensor as the working form)
__global__ void frame_index_kernel(const float* __restrict__ cum,
                                   const int64_t* __restrict__ y_lengths, int B, int Tx, int Ty,
                                   int32_t* __restrict__ f2p, float* __restrict__ y_mask) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Ty) return;
  int ty = idx % Ty, b = idx / Ty;
  bool valid = (int64_t)ty < y_lengths[b];
  int res = -1;
  if (valid) {
    // first tx with cum[tx] > ty  (path[tx] = (ty < cum[tx]) - (ty < cum[tx-1]), commons.py:128-134)
    const float* cr = cum + (int64_t)b * Tx;
    int lo = 0, hi = Tx;
    float fy = (float)ty;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (cr[mid] > fy) hi = mid; else lo = mid + 1;
    }
    if (lo < Tx) res = lo;
  }
  f2p[idx] = res;
  y_mask[idx] = valid ? 1.f : 0.f;
}

int32_t k_frame_index(const float* cum, const int64_t* y_lengths, int B, int Tx, int Ty,
                      int32_t* frame2phone, float* y_mask, hipStream_t s) {
  if (B * Ty == 0) return WETTS_OK;
  hipLaunchKernelGGL(frame_index_kernel, grid1d((int64_t)B * Ty, 256), dim3(256), 0, s, cum,
                     y_lengths, B, Tx, Ty, frame2phone, y_mask);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void expand_prior_kernel(const float* __restrict__ stats,
                                    const int32_t* __restrict__ f2p, const float* __restrict__ eps,
                                    int64_t eps_bs, int64_t eps_cs, float noise_scale, int B, int C, int Tx, int Ty,
                                    float* __restrict__ m_exp, float* __restrict__ logs_exp,
                                    float* __restrict__ z_p) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * C * Ty;
  if (idx >= total) return;
  int ty = (int)(idx % Ty);
  int c = (int)((idx / Ty) % C);
  int b = (int)(idx / ((int64_t)Ty * C));
  int tx = f2p[(int64_t)b * Ty + ty];
  float m = 0.f, ls = 0.f;
  if (tx >= 0) {
    m = stats[((int64_t)b * 2 * C + c) * Tx + tx];
    ls = stats[((int64_t)b * 2 * C + C + c) * Tx + tx];
  }
  if (m_exp) m_exp[idx] = m;
  if (logs_exp) logs_exp[idx] = ls;
  const float e = eps[(int64_t)b * eps_bs + (int64_t)c * eps_cs + ty];
  z_p[idx] = m + e * expf(ls) * noise_scale;  // models.py:267
}

int32_t k_expand_prior(const float* stats, const int32_t* frame2phone, const float* eps,
                       int64_t eps_bs, int64_t eps_cs, float noise_scale, int B, int C, int Tx, int Ty, float* m_exp,
                       float* logs_exp, float* z_p, hipStream_t s) {
  int64_t n = (int64_t)B * C * Ty;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(expand_prior_kernel, grid1d(n, 256), dim3(256), 0, s, stats, frame2phone, eps,
                     eps_bs, eps_cs, noise_scale, B, C, Tx, Ty, m_exp, logs_exp, z_p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void attn_path_kernel(const int32_t* __restrict__ f2p, int B, int Tx, int Ty,
                                 float* __restrict__ attn) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = (int64_t)B * Ty * Tx;
  if (idx >= total) return;
  int tx = (int)(idx % Tx);
  int64_t bty = idx / Tx;
  attn[idx] = (f2p[bty] == tx) ? 1.f : 0.f;
}

int32_t k_attn_path(const int32_t* frame2phone, int B, int Tx, int Ty, float* attn, hipStream_t s) {
  int64_t n = (int64_t)B * Ty * Tx;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(attn_path_kernel, grid1d(n, 256), dim3(256), 0, s, frame2phone, B, Tx, Ty,
                     attn);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void audio_to_int16_kernel(const float* __restrict__ audio,
                                                             const int64_t* __restrict__ lengths,
                                                             int64_t L, int16_t* __restrict__ pcm) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const float* a = audio + (int64_t)b * L;
  int64_t n = lengths ? lengths[b] : L;
  if (n > L) n = L;
  float mx = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x) mx = fmaxf(mx, fabsf(a[i]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  // audio *= 32767 / max(0.01, max|audio|) * 0.6   (inference.py:101)
  const float scale = (32767.f / fmaxf(0.01f, mx)) * 0.6f;
  for (int64_t i = threadIdx.x; i < L; i += blockDim.x) {
    float v = a[i] * scale;
    v = fminf(fmaxf(v, -32767.f), 32767.f);
    pcm[(int64_t)b * L + i] = (int16_t)v;  // numpy astype(int16) truncates toward zero
  }
}

int32_t k_audio_to_int16(const float* audio, const int64_t* lengths, int B, int64_t L, int16_t* pcm,
                         hipStream_t s) {
  if (B == 0 || L == 0) return WETTS_OK;
  hipLaunchKernelGGL(audio_to_int16_kernel, dim3(B), dim3(256), 0, s, audio, lengths, L, pcm);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}


// ---------------------------------------------------------------------------------------------
// VocosGenerator pieces (decoders.py:251-308)
// ---------------------------------------------------------------------------------------------
__global__ void vocos_pad_kernel(const float* __restrict__ z, int64_t z_bs, int64_t z_cs,
                                 const float* __restrict__ mask, int64_t mask_stride, int B, int C,
                                 int L, int Fs, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int F = L + 1;
  if (idx >= (int64_t)B * C * Fs) return;
  const int f = (int)(idx % Fs);
  const int c = (int)((idx / Fs) % C);
  const int b = (int)(idx / ((int64_t)Fs * C));
  float v = 0.f;  // columns F .. Fs - 1: row padding (kept finite)
  if (f < F) {
    const int t = f == 0 ? 1 : f - 1;  // reflect: padded[0] = x[1]
    v = z[(int64_t)b * z_bs + (int64_t)c * z_cs + t];
    if (mask) v *= mask[(int64_t)b * mask_stride + t];
  }
  out[idx] = v;
}

int32_t k_vocos_pad(const float* z, int64_t z_bs, int64_t z_cs, const float* mask,
                    int64_t mask_stride, int B, int C, int L, float* out, hipStream_t s, int Fs) {
  if (Fs <= 0) Fs = L + 1;
  int64_t n = (int64_t)B * C * Fs;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(vocos_pad_kernel, grid1d(n, 256), dim3(256), 0, s, z, z_bs, z_cs, mask,
                     mask_stride, B, C, L, Fs, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void vocos_spec_kernel(const float* __restrict__ spec, int B, int half, int F, int64_t ri_bs,
                                  float* __restrict__ ri) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * half * F) return;
  const int f = (int)(idx % F);
  const int k = (int)((idx / F) % half);
  const int b = (int)(idx / ((int64_t)F * half));
  const int64_t base = (int64_t)b * 2 * half * F, obase = (int64_t)b * ri_bs;
  const float lm = spec[base + (int64_t)k * F + f];
  const float ph = spec[base + (int64_t)(half + k) * F + f];
  const float mag = fminf(expf(lm), 1e2f);  // mag.exp().clamp_max(1e2)
  ri[obase + (int64_t)k * F + f] = mag * cosf(ph);
  ri[obase + (int64_t)(half + k) * F + f] = mag * sinf(ph);
}

int32_t k_vocos_spec(const float* spec, int B, int half, int F, float* ri, hipStream_t s, int64_t ri_bs) {
  int64_t n = (int64_t)B * half * F;
  if (n == 0) return WETTS_OK;
  if (ri_bs <= 0) ri_bs = (int64_t)2 * half * F;
  hipLaunchKernelGGL(vocos_spec_kernel, grid1d(n, 256), dim3(256), 0, s, spec, B, half, F, ri_bs, ri);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// w[n][c]: c < half -> coefficient of Re S[c]; c >= half -> coefficient of Im S[c-half]
//   irfft: x[n] = (1/N) sum_k a_k (Re S_k cos(2 pi k n / N) - Im S_k sin(2 pi k n / N)),
//   a_0 = a_{N/2} = 1, else 2; times the periodic hann window (torch.hann_window default)
// inv_scale != 0 selects OnnxSTFT's inverse basis (utils/stft.py:272-290): pinv(scale * F)^T * hann with F the
// stacked [Re; Im] rows 0..N/2 of the DFT matrix and scale = n_fft / hop.  F^T F = (N/2) I + (1 1^T + a a^T) / 2
// (a_n = (-1)^n), whose inverse is (2/N) I - (1 1^T + a a^T) / N^2, so pinv(F) = (F^T F)^-1 F^T is EXACTLY the irfft
// matrix above (weights 1, 2, .., 2, 1 over N; the Im rows of k = 0 and N/2 are zero) and pinv(scale F) = irfft /
// scale.  The reference rounds the float64 pinv to float32 and multiplies by the float32 window in float32
// (stft.py:279-287); the same two roundings are made here.
__global__ void istft_basis_kernel(int n_fft, double inv_scale, float* __restrict__ w) {
  const int half = n_fft / 2 + 1;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_fft * 2 * half) return;
  const int c = (int)(idx % (2 * half));
  const int n = (int)(idx / (2 * half));
  const int k = c < half ? c : c - half;
  const double pi2 = 6.283185307179586476925286766559;
  const double win = 0.5 - 0.5 * cos(pi2 * n / n_fft);
  const double a = (k == 0 || k == n_fft / 2) ? 1.0 : 2.0;
  // reduce k*n mod N before the trig call to keep the argument small
  const double ang = pi2 * (double)((int64_t)k * n % n_fft) / n_fft;
  const double v = c < half ? cos(ang) : -sin(ang);
  if (inv_scale != 0.0)
    w[idx] = (float)(a * v / n_fft * inv_scale) * (float)win;
  else
    w[idx] = (float)(win * a * v / n_fft);
}

int32_t k_istft_basis(int n_fft, float* w, hipStream_t s, double inv_scale) {
  int64_t n = (int64_t)n_fft * (n_fft + 2);
  hipLaunchKernelGGL(istft_basis_kernel, grid1d(n, 256), dim3(256), 0, s, n_fft, inv_scale, w);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

template <bool ENV>
__global__ void istft_ola_kernel(const float* __restrict__ frames, int B, int n_fft, int hop, int F, int Fs,
                                 float* __restrict__ audio) {
  const int64_t Ls = (int64_t)(F - 1) * hop;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * Ls) return;
  const int b = (int)(idx / Ls);
  const int64_t n = idx % Ls;
  const int64_t m = n + n_fft / 2;  // position in the un-trimmed overlap-add buffer
  int64_t f_hi = m / hop;
  if (f_hi > F - 1) f_hi = F - 1;
  int64_t f_lo = (m - n_fft + hop) / hop;  // ceil((m - n_fft + 1) / hop) for m - n_fft + 1 > 0
  if (m - n_fft + 1 <= 0) f_lo = 0;
  const float* fb = frames + (int64_t)b * n_fft * Fs;
  const float pi2 = 6.283185307179586f;
  float y = 0.f, env = 0.f;
  for (int64_t f = f_lo; f <= f_hi; ++f) {
    const int j = (int)(m - f * hop);  // 0 <= j < n_fft
    y += fb[(int64_t)j * Fs + f];
    if (ENV) {
      const float wv = 0.5f - 0.5f * cosf(pi2 * (float)j / (float)n_fft);
      env += wv * wv;
    }
  }
  // torch.istft: y / window_envelope (asserted > 1e-11 there); OnnxSTFT.inverse (stft.py:325-340): the bare
  // overlap-add of conv_transpose1d, trimmed
  audio[idx] = ENV ? y / env : y;
}

int32_t k_istft_ola(const float* frames, int B, int n_fft, int hop, int F, float* audio,
                    hipStream_t s, int Fs, int envelope) {
  int64_t n = (int64_t)B * (F - 1) * hop;
  if (n <= 0) return WETTS_OK;
  if (envelope)
    hipLaunchKernelGGL(istft_ola_kernel<true>, grid1d(n, 256), dim3(256), 0, s, frames, B, n_fft, hop, F,
                       Fs > 0 ? Fs : F, audio);
  else
    hipLaunchKernelGGL(istft_ola_kernel<false>, grid1d(n, 256), dim3(256), 0, s, frames, B, n_fft, hop, F,
                       Fs > 0 ? Fs : F, audio);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void scale_rows_kernel(const float* __restrict__ a, const float* __restrict__ scale,
                                  int rows, int cols, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)rows * cols) return;
  out[idx] = a[idx] * scale[idx / cols];
}

int32_t k_scale_rows(const float* a, const float* scale, int rows, int cols, float* out,
                     hipStream_t s) {
  int64_t n = (int64_t)rows * cols;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(scale_rows_kernel, grid1d(n, 256), dim3(256), 0, s, a, scale, rows, cols, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}


__global__ void flip_half_kernel(const float* __restrict__ x, const float* __restrict__ mask, int B,
                                 int C, int T, float* __restrict__ x0, float* __restrict__ x0m) {
  const int half = C / 2;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * half * T) return;
  const int t = (int)(idx % T);
  const int c = (int)((idx / T) % half);
  const int b = (int)(idx / ((int64_t)T * half));
  const float v = x[((int64_t)b * C + (C - 1 - c)) * T + t];
  x0[idx] = v;
  x0m[idx] = v * mask[(int64_t)b * T + t];
}

int32_t k_flip_half(const float* x, const float* mask, int B, int C, int T, float* x0, float* x0m,
                    hipStream_t s) {
  int64_t n = (int64_t)B * (C / 2) * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(flip_half_kernel, grid1d(n, 256), dim3(256), 0, s, x, mask, B, C, T, x0, x0m);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// MonoTransformerFlowLayer reverse, input side (flows.py:287-290,303-304): x0 = x[:, :C/2] * sc copied out, raw and masked
__global__ void mono_split_kernel(const float* __restrict__ x, const float* __restrict__ mask, int B, int C, int T,
                                  float sc, float* __restrict__ x0, float* __restrict__ x0m) {
  const int half = C / 2;
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * half * T) return;
  const int t = (int)(idx % T);
  const int c = (int)((idx / T) % half);
  const int b = (int)(idx / ((int64_t)T * half));
  const float v = x[((int64_t)b * C + c) * T + t] * sc;
  x0[idx] = v;
  x0m[idx] = v * mask[(int64_t)b * T + t];
}

int32_t k_mono_split(const float* x, const float* mask, int B, int C, int T, float sc, float* x0, float* x0m,
                     hipStream_t s) {
  int64_t n = (int64_t)B * (C / 2) * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(mono_split_kernel, grid1d(n, 256), dim3(256), 0, s, x, mask, B, C, T, sc, x0, x0m);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// MonoTransformerFlowLayer reverse, output side (flows.py:297-299,320-322 with mean_only => logs = 0):
//   out[c] = c < half ? x[c] * sc : (x[c] - m[c - half]) * sc * mask;  sc = 1 (inter) or 1/2 (post: x0 / 2 and
//   1 / (1 + exp(-0)), both exact as a multiply by 0.5)
__global__ void mono_coupling_kernel(const float* __restrict__ x, const float* __restrict__ m,
                                     const float* __restrict__ mask, int B, int C, int T, float sc,
                                     float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * C * T) return;
  const int t = (int)(idx % T);
  const int c = (int)((idx / T) % C);
  const int b = (int)(idx / ((int64_t)T * C));
  const int half = C / 2;
  float v = x[idx];
  if (c >= half) v = ((v - m[((int64_t)b * half + (c - half)) * T + t]) * sc) * mask[(int64_t)b * T + t];
  else v = v * sc;
  out[idx] = v;
}

int32_t k_mono_coupling(const float* x, const float* m, const float* mask, int B, int C, int T, float sc, float* out,
                        hipStream_t s) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(mono_coupling_kernel, grid1d(n, 256), dim3(256), 0, s, x, m, mask, B, C, T, sc, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// dst[r][c] = c < cols_src ? src[r][c] : 0 for c < cols_dst: rows re-strided (padded to a multiple of 4 columns
// on the way in, trimmed on the way out)
__global__ void copy_rows_kernel(const float* __restrict__ src, int64_t src_stride, int cols_src,
                                 float* __restrict__ dst, int64_t dst_stride, int cols_dst, int64_t rows) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * cols_dst) return;
  const int c = (int)(idx % cols_dst);
  const int64_t r = idx / cols_dst;
  dst[r * dst_stride + c] = c < cols_src ? src[r * src_stride + c] : 0.f;
}

int32_t k_copy_rows(const float* src, int64_t src_stride, int cols_src, float* dst, int64_t dst_stride, int cols_dst,
                    int64_t rows, hipStream_t s) {
  int64_t n = rows * cols_dst;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(copy_rows_kernel, grid1d(n, 256), dim3(256), 0, s, src, src_stride, cols_src, dst, dst_stride,
                     cols_dst, rows);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                           float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) out[idx] = a[idx] + b[idx];
}

int32_t k_add(const float* a, const float* b, int64_t n, float* out, hipStream_t s) {
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(add_kernel, grid1d(n, 256), dim3(256), 0, s, a, b, n, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}


__global__ void add_bias_b_mask_kernel(float* __restrict__ x, const float* __restrict__ v,
                                       const float* __restrict__ mask, int64_t total, int C, int T) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int t = (int)(idx % T);
  const int64_t bc = idx / T;
  x[idx] = (x[idx] + v[bc]) * mask[(bc / C) * T + t];
}

int32_t k_add_bias_b_mask(float* x, const float* v, const float* mask, int B, int C, int T,
                          hipStream_t s) {
  int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(add_bias_b_mask_kernel, grid1d(n, 256), dim3(256), 0, s, x, v, mask, n, C, T);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
// Standard-normal draws on the device: Philox4x32-10 counter RNG (Salmon et al., SC'11) + Box-Muller.
// Replaces the two torch.randn calls of the reference (duration_predictors.py:257, models.py:267);
// element i of a draw depends only on (seed, offset + i/4), so results do not depend on the launch
// geometry.  One thread produces four values.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__global__ void randn_kernel(float* __restrict__ out, int64_t n, uint64_t seed, uint64_t offset) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 outputs
  if (q * 4 >= n) return;
  const uint64_t ctr = offset + (uint64_t)q;
  uint32_t r[4];
  philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  float v[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // u1 in (0, 1], u2 in [0, 1): 24-bit mantissas, log(u1) finite
    const float u1 = ((float)(r[2 * h] >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(r[2 * h + 1] >> 8) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.28318530717958647692f * u2, &sn, &cs);
    v[2 * h] = rad * cs;
    v[2 * h + 1] = rad * sn;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (q * 4 + i < n) out[q * 4 + i] = v[i];
}

int32_t k_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s) {
  if (n <= 0) return WETTS_OK;
  hipLaunchKernelGGL(randn_kernel, grid1d((n + 3) / 4, 256), dim3(256), 0, s, out, n, seed, offset);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// out[b,c,t] = x[b,c,t] * mask[b,t]   (`z * y_mask` of infer_encoder, models.py:322-331)
__global__ void mask_rows_kernel(const float* __restrict__ x, const float* __restrict__ mask,
                                 int64_t total, int C, int T, float* __restrict__ out) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int64_t b = idx / ((int64_t)C * T);
  out[idx] = x[idx] * mask[b * T + idx % T];
}

int32_t k_mask_rows(const float* x, const float* mask, int B, int C, int T, float* out,
                    hipStream_t s) {
  const int64_t n = (int64_t)B * C * T;
  if (n == 0) return WETTS_OK;
  hipLaunchKernelGGL(mask_rows_kernel, grid1d(n, 256), dim3(256), 0, s, x, mask, n, C, T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
