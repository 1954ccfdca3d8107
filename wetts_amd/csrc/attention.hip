// Windowed relative-position self-attention of the VITS text encoder (attentions.py:235-282),
// written in its direct banded form instead of the reference's pad/reshape skew
// (attentions.py:284-358): with window w the relative-key term adds (q_i/sqrt(dk)) . E_k[j-i+w]
// to score (i,j) only for |j-i| <= w, and the relative-value term adds
// sum_{|r|<=w} P[i,i+r] * E_v[r+w] to the output -- 2w+1 extra dot products per query.
//
// Scores are kept TRANSPOSED in the workspace: S[b,h,j,i] with the query index i fastest, so in
// all three kernels consecutive lanes touch consecutive addresses (q, S and the output are
// stride-1 in i; k / v / E values are wave-uniform broadcasts).
#include "kernels.h"

namespace wetts {

// grid: (ceil(T/64), T(j), B*H)   block 64
__global__ __launch_bounds__(64) void attn_scores_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask,
    const float* __restrict__ emb_rel_k, int window, int n_heads, int dk, int T, float qdiv,
    float* __restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y;
  const int bh = blockIdx.z;
  const int b = bh / n_heads, h = bh % n_heads;
  if (i >= T) return;
  const float* qb = q + ((int64_t)b * n_heads + h) * dk * T;
  const float* kb = k + ((int64_t)b * n_heads + h) * dk * T;
  const int r = j - i;
  const bool in_band = (r >= -window) && (r <= window);
  const float* er = emb_rel_k + (int64_t)(in_band ? (r + window) : 0) * dk;
  float acc = 0.f, rel = 0.f;
  for (int d = 0; d < dk; ++d) {
    float qs = qb[(int64_t)d * T + i] / qdiv;  // query / math.sqrt(k_channels)
    acc += qs * kb[(int64_t)d * T + j];
    if (in_band) rel += qs * er[d];
  }
  float sc = acc + rel;
  const float mi = mask[(int64_t)b * T + i], mj = mask[(int64_t)b * T + j];
  if (mi * mj == 0.f) sc = -1e4f;  // masked_fill(mask == 0, -1e4)
  S[(((int64_t)bh * T) + j) * T + i] = sc;
}

// softmax over j for every (b,h,i); block = 16 query lanes x 16 key groups (latency-bound: T is a
// few hundred, so many short threads); grid: (ceil(T/16), B*H)
__global__ __launch_bounds__(256) void attn_softmax_kernel(int T, float* __restrict__ S) {
  constexpr int TL = 16, JG = 16;
  __shared__ float red[JG][TL + 1];
  const int il = threadIdx.x % TL, jg = threadIdx.x / TL;
  const int i = blockIdx.x * TL + il;
  const int bh = blockIdx.y;
  const bool ok = i < T;
  float* col = S + (int64_t)bh * T * T + (ok ? i : 0);
  const int j0 = (T * jg) / JG, j1 = (T * (jg + 1)) / JG;
  float mx = -INFINITY;
  if (ok)
    for (int j = j0; j < j1; ++j) mx = fmaxf(mx, col[(int64_t)j * T]);
  red[jg][il] = mx;
  __syncthreads();
  mx = -INFINITY;
#pragma unroll
  for (int q = 0; q < JG; ++q) mx = fmaxf(mx, red[q][il]);
  __syncthreads();
  float sm = 0.f;
  if (ok)
    for (int j = j0; j < j1; ++j) {
      float e = expf(col[(int64_t)j * T] - mx);
      col[(int64_t)j * T] = e;
      sm += e;
    }
  red[jg][il] = sm;
  __syncthreads();
  sm = 0.f;
#pragma unroll
  for (int q = 0; q < JG; ++q) sm += red[q][il];
  if (ok)
    for (int j = j0; j < j1; ++j) col[(int64_t)j * T] = col[(int64_t)j * T] / sm;
}

// out[b, h*dk+d, i] = sum_j P[j,i] v[d,j] + sum_r P[i+r,i] E_v[r+w][d];  grid (ceil(T/64), dk, B*H)
__global__ __launch_bounds__(64) void attn_pv_kernel(const float* __restrict__ P,
                                                     const float* __restrict__ v,
                                                     const float* __restrict__ emb_rel_v,
                                                     int window, int n_heads, int dk, int T,
                                                     float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int d = blockIdx.y;
  const int bh = blockIdx.z;
  if (i >= T) return;
  const float* Pc = P + (int64_t)bh * T * T + i;
  const float* vr = v + ((int64_t)bh * dk + d) * T;
  float acc = 0.f;
  for (int j = 0; j < T; ++j) acc += Pc[(int64_t)j * T] * vr[j];
  float rel = 0.f;
  for (int r = -window; r <= window; ++r) {
    int j = i + r;
    if (j >= 0 && j < T) rel += Pc[(int64_t)j * T] * emb_rel_v[(int64_t)(r + window) * dk + d];
  }
  out[((int64_t)bh * dk + d) * T + i] = acc + rel;
}

int32_t k_rel_attention(const float* q, const float* k, const float* v, const float* mask,
                        const float* emb_rel_k, const float* emb_rel_v, int window, int B,
                        int n_heads, int dk, int T, float* scores, float* out, hipStream_t s) {
  if (B * T == 0) return WETTS_OK;
  WETTS_REQUIRE(T <= 65535, "attention length %d too large", T);
  const float qdiv = (float)sqrt((double)dk);
  int tb = cdiv(T, 64);
  hipLaunchKernelGGL(attn_scores_kernel, dim3(tb, T, B * n_heads), dim3(64), 0, s, q, k, mask,
                     emb_rel_k, window, n_heads, dk, T, qdiv, scores);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_softmax_kernel, dim3(cdiv(T, 16), B * n_heads), dim3(256), 0, s, T, scores);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_pv_kernel, dim3(tb, dk, B * n_heads), dim3(64), 0, s, scores, v,
                     emb_rel_v, window, n_heads, dk, T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
