// Windowed relative-position self-attention of the VITS text encoder (attentions.py:235-282),
// written in its direct banded form instead of the reference's pad/reshape skew
// (attentions.py:284-358): with window w the relative-key term adds (q_i/sqrt(dk)) . E_k[j-i+w]
// to score (i,j) only for |j-i| <= w, and the relative-value term adds
// sum_{|r|<=w} P[i,i+r] * E_v[r+w] to the output -- 2w+1 extra dot products per query.
//
// Scores are kept TRANSPOSED in the workspace: S[b,h,j,i] with the query index i fastest, so in
// all three kernels consecutive lanes touch consecutive addresses (q, S and the output are
// stride-1 in i; k / v / E values are wave-uniform broadcasts).
#include <stdlib.h>

#include "kernels.h"

namespace wetts {

// grid: (ceil(T/64), T(j), B*H)   block 64
__global__ __launch_bounds__(64) void attn_scores_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask,
    const float* __restrict__ emb_rel_k, int window, int n_heads, int dk, int T, float qdiv,
    int64_t qbs, float* __restrict__ S) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int j = blockIdx.y;
  const int bh = blockIdx.z;
  const int b = bh / n_heads, h = bh % n_heads;
  if (i >= T) return;
  const float* qb = q + (int64_t)b * qbs + (int64_t)h * dk * T;
  const float* kb = k + (int64_t)b * qbs + (int64_t)h * dk * T;
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

// softmax over j for every (b,h,i); block = 16 query lanes x 16 key groups; grid: (ceil(T/16), B*H).
// Two passes over S instead of three: each thread keeps a running (max, sum) over its keys
// (online rescaling), the 16 groups are merged through LDS, then P = exp(s - M) / Z is written.
__global__ __launch_bounds__(256) void attn_softmax_kernel(int T, float* __restrict__ S) {
  constexpr int TL = 16, JG = 16;
  __shared__ float red_m[JG][TL + 1];
  __shared__ float red_s[JG][TL + 1];
  const int il = threadIdx.x % TL, jg = threadIdx.x / TL;
  const int i = blockIdx.x * TL + il;
  const int bh = blockIdx.y;
  const bool ok = i < T;
  float* col = S + (int64_t)bh * T * T + (ok ? i : 0);
  const int j0 = (T * jg) / JG, j1 = (T * (jg + 1)) / JG;
  float mx = -INFINITY, sm = 0.f;
  if (ok) {
    for (int j = j0; j < j1; ++j) {
      const float x = col[(int64_t)j * T];
      if (x > mx) {
        sm = sm * expf(mx - x);  // exp(-inf) = 0 on the first element
        mx = x;
      }
      sm += expf(x - mx);
    }
  }
  red_m[jg][il] = mx;
  red_s[jg][il] = sm;
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int q = 0; q < JG; ++q) M = fmaxf(M, red_m[q][il]);
  float Z = 0.f;
#pragma unroll
  for (int q = 0; q < JG; ++q) {
    const float mq = red_m[q][il];
    if (mq > -INFINITY) Z += red_s[q][il] * expf(mq - M);
  }
  if (ok)
    for (int j = j0; j < j1; ++j) col[(int64_t)j * T] = expf(col[(int64_t)j * T] - M) / Z;
}

// out[b, h*dk+d, i] = sum_j P[j,i] v[d,j] + sum_r P[i+r,i] E_v[r+w][d];  grid (ceil(T/64), dk, B*H)
__global__ __launch_bounds__(64) void attn_pv_kernel(const float* __restrict__ P,
                                                     const float* __restrict__ v,
                                                     const float* __restrict__ emb_rel_v,
                                                     int window, int n_heads, int dk, int T,
                                                     int64_t qbs, float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int d = blockIdx.y;
  const int bh = blockIdx.z;
  if (i >= T) return;
  const float* Pc = P + (int64_t)bh * T * T + i;
  const float* vr = v + (int64_t)(bh / n_heads) * qbs + ((int64_t)(bh % n_heads) * dk + d) * T;
  float acc = 0.f;
  for (int j = 0; j < T; ++j) acc += Pc[(int64_t)j * T] * vr[j];
  float rel = 0.f;
  for (int r = -window; r <= window; ++r) {
    int j = i + r;
    if (j >= 0 && j < T) rel += Pc[(int64_t)j * T] * emb_rel_v[(int64_t)(r + window) * dk + d];
  }
  out[((int64_t)bh * dk + d) * T + i] = acc + rel;
}

// ---------------------------------------------------------------------------------------------
// MFMA path for window-less attention over long sequences (the VITS2 flow encoders run over Ty
// ~ 800 frames, flows.py:111-119): both contractions on v_mfma_f32_32x32x2_f32 straight from the
// L2-resident q / k / v (160 KB per head) -- no packing: with S kept transposed the A and B
// operand loads are stride-1 across lanes.
// ---------------------------------------------------------------------------------------------
typedef float f32x16a __attribute__((ext_vector_type(16)));

// S[b,h,j,i] = sum_d k[d][j] * (q[d][i] / sqrt(dk)), masked_fill(mask_i * mask_j == 0, -1e4)
// block = 4 waves = 4 j-blocks of 32 x one i-block of 32;  grid (ceil(T/32), ceil(T/128), B*H)
// rel[bh][r][i] = (q_i / sqrt(dk)) . E_k[r]   for the 2w+1 relative positions (attentions.py:246-252)
__global__ void attn_relk_kernel(const float* __restrict__ q, const float* __restrict__ emb_rel_k,
                                 int nrel, int n_heads, int dk, int T, float qdiv, int64_t qbs,
                                 float* __restrict__ rel) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const int r = blockIdx.y, bh = blockIdx.z;
  if (i >= T) return;
  const float* qb = q + (int64_t)(bh / n_heads) * qbs + (int64_t)(bh % n_heads) * dk * T;
  const float* er = emb_rel_k + (int64_t)r * dk;
  float acc = 0.f;
  int d0 = 0;
  for (; d0 + 16 <= dk; d0 += 16) {  // 16 loads in flight, then their FMAs (same sum order as a plain loop)
    float qv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) qv[u] = qb[(int64_t)(d0 + u) * T + i];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += (qv[u] / qdiv) * er[d0 + u];
  }
  for (int d = d0; d < dk; ++d) acc += (qb[(int64_t)d * T + i] / qdiv) * er[d];
  rel[((int64_t)bh * nrel + r) * T + i] = acc;
}

// out[bh][d][i] += sum_{|r|<=w} P[i+r][i] * E_v[r+w][d]   (attentions.py:273-279)
// block = 64 query lanes x 4 channel groups: a thread keeps its 2w+1 probabilities in registers and walks
// its quarter of the channels (E_v in LDS, broadcast reads), instead of one thread per (i, d) re-reading
// the probabilities dk times
constexpr int kRelvMaxW = 8;
__global__ __launch_bounds__(256) void attn_relv_add_kernel(const float* __restrict__ P,
                                                            const float* __restrict__ emb_rel_v, int window,
                                                            int dk, int T, float* __restrict__ out) {
  extern __shared__ float esh[];  // [2w+1][dk]
  const int nrel = 2 * window + 1;
  for (int q = threadIdx.x; q < nrel * dk; q += blockDim.x) esh[q] = emb_rel_v[q];
  __syncthreads();
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), dg = threadIdx.x >> 6, bh = blockIdx.y;
  if (i >= T) return;
  const float* Pc = P + (int64_t)bh * T * T + i;
  float pv[2 * kRelvMaxW + 1];
#pragma unroll
  for (int q = 0; q < 2 * kRelvMaxW + 1; ++q) {
    const int j = i + q - window;
    pv[q] = (q < nrel && j >= 0 && j < T) ? Pc[(int64_t)j * T] : 0.f;
  }
  const int d0 = (dk * dg) / 4, d1 = (dk * (dg + 1)) / 4;
  float* ob = out + (int64_t)bh * dk * T + i;
  for (int db = d0; db < d1; db += 8) {  // eight read-modify-writes in flight at a time
    float o[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) o[u] = ob[(int64_t)(db + u < d1 ? db + u : d0) * T];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = db + u < d1 ? db + u : d0;
      float rel = 0.f;
#pragma unroll
      for (int q = 0; q < 2 * kRelvMaxW + 1; ++q)
        if (q < nrel) {
          const int j = i + q - window;
          if (j >= 0 && j < T) rel += pv[q] * esh[q * dk + d];  // same terms, same order as the (i, d) form
        }
      o[u] += rel;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (db + u < d1) ob[(int64_t)(db + u) * T] = o[u];
  }
}

__global__ __launch_bounds__(256) void attn_scores_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ mask,
    const float* __restrict__ rel, int window, int n_heads, int dk, int T, float qdiv, int64_t qbs,
    float* __restrict__ S) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int bh = blockIdx.z, b = bh / n_heads;
  const int i = blockIdx.x * 32 + (lane & 31);
  const int jb = (blockIdx.y * 4 + wave) * 32;
  if (jb >= T) return;
  const int j = jb + (lane & 31);
  const float* qb = q + (int64_t)b * qbs + (int64_t)(bh % n_heads) * dk * T;
  const float* kb = k + (int64_t)b * qbs + (int64_t)(bh % n_heads) * dk * T;
  const bool iok = i < T, jok = j < T;
  f32x16a acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int nks = (dk + 1) / 2;
  const int jcl = jok ? j : 0, icl = iok ? i : 0;
  for (int k0 = 0; k0 < nks; k0 += 8) {  // loads of 8 k-steps in flight, then their MFMAs
    float a[8], bq[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int d = 2 * (k0 + u) + half;
      const bool dv = d < dk;
      const int dc = dv ? d : 0;
      const float av = kb[(int64_t)dc * T + jcl];
      const float qv = qb[(int64_t)dc * T + icl];
      a[u] = (jok && dv) ? av : 0.f;
      bq[u] = (iok && dv) ? qv / qdiv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bq[u], acc, 0, 0, 0);
  }
  if (!iok) return;
  const float mi = mask[(int64_t)b * T + i];
  float* Sb = S + (int64_t)bh * T * T + i;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int jj = jb + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (jj < T) {
      float sc = acc[r];
      const int rr = jj - i + window;  // relative-key term inside the band (window >= 0 only)
      if (rel && rr >= 0 && rr <= 2 * window) sc += rel[((int64_t)bh * (2 * window + 1) + rr) * T + i];
      if (mi * mask[(int64_t)b * T + jj] == 0.f) sc = -1e4f;
      Sb[(int64_t)jj * T] = sc;
    }
  }
}

// vT[bh][j][d] = v[bh][d][j]
__global__ void attn_transpose_v_kernel(const float* __restrict__ v, int n_heads, int dk, int T,
                                        int64_t qbs, float* __restrict__ vT, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int d = (int)(idx % dk);
  const int j = (int)((idx / dk) % T);
  const int64_t bh = idx / ((int64_t)dk * T);
  vT[idx] = v[(bh / n_heads) * qbs + ((bh % n_heads) * dk + d) * T + j];
}

// out[bh][d][i] = sum_j vT[j][d] * P[j][i]
// block = 4 waves = 4 i-blocks of 32 x one d-block of 32;  grid (ceil(T/128), ceil(dk/32), B*H)
__global__ __launch_bounds__(256) void attn_pv_mfma_kernel(const float* __restrict__ P,
                                                           const float* __restrict__ vT, int dk,
                                                           int T, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5;
  const int bh = blockIdx.z;
  const int ib = (blockIdx.x * 4 + wave) * 32;
  if (ib >= T) return;
  const int i = ib + (lane & 31);
  const int d = blockIdx.y * 32 + (lane & 31);
  const bool iok = i < T, dok = d < dk;
  const float* Pb = P + (int64_t)bh * T * T + (iok ? i : 0);
  const float* vb = vT + (int64_t)bh * T * dk + (dok ? d : 0);
  f32x16a acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // 8 k-steps per trip: all 16 loads are issued before the first MFMA (the plain loop exposed one
  // L2 round trip per k-step), and two accumulators halve the dependent-MFMA chain.  Out-of-range
  // steps read a clamped address and contribute a zero operand.
  f32x16a acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
  const int nks = (T + 1) / 2;
  for (int k0 = 0; k0 < nks; k0 += 8) {
    float a[8], bp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = 2 * (k0 + u) + half;
      const bool jv = j < T;
      const int jc = jv ? j : 0;
      const float av = vb[(int64_t)jc * dk];
      const float bv = Pb[(int64_t)jc * T];
      a[u] = (dok && jv) ? av : 0.f;
      bp[u] = (iok && jv) ? bv : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bp[u], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u + 1], bp[u + 1], acc2, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
  if (!iok) return;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dd = blockIdx.y * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (dd < dk) out[((int64_t)bh * dk + dd) * T + i] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------------
// Window-less attention in one kernel (flash style): the S^T tile of 32 keys x 32 queries leaves
// the score MFMAs in exactly the lane layout the P.V MFMA wants for its B operand -- lane (i, half)
// holds P[j][i] for j = jmap(r, half), r = 0..15 -- so k-step r of the second contraction takes
// acc[r] as is and loads v at the same 16 key positions: no S round trip through HBM, no
// cross-lane movement.  Online softmax per query column (max / sum combined across the two lane
// halves); the block's 4 waves split the key range and merge their (m, l, o) through LDS.
// Masking follows attentions.py:253-256: masked pairs score -1e4 (a fully padded query row
// therefore averages v over all T keys, like the reference), keys >= T do not exist.
// grid (ceil(T/32), B*H), block 256;  NKS = k-steps over dk (2 per step), NDB = 32-row d-blocks
// ---------------------------------------------------------------------------------------------
template <int NKS, int NDB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NKS <= 24 ? 2 : 1))) void attn_flash_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ vT,
    const float* __restrict__ mask, int n_heads, int dk, int T, float qdiv, int64_t qbs,
    float* __restrict__ out) {
  constexpr int NR = NDB * 16;
  __shared__ float sm_m[4][32];
  __shared__ float sm_l[4][32];
  __shared__ float sm_o[4 * NR * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, col = lane & 31;
  const int bh = blockIdx.y, b = bh / n_heads;
  const int i = blockIdx.x * 32 + col;
  const bool iok = i < T;
  const int icl = iok ? i : T - 1;
  const float* qb = q + (int64_t)b * qbs + (int64_t)(bh % n_heads) * dk * T;
  const float* kb = k + (int64_t)b * qbs + (int64_t)(bh % n_heads) * dk * T;
  const float* vb = vT + (int64_t)bh * T * dk;
  const float* mb = mask + (int64_t)b * T;
  float qr[NKS];
#pragma unroll
  for (int u = 0; u < NKS; ++u) {
    const int d = 2 * u + half;
    // unconditional load (clamped row) times a 0/1 factor: a select would make the load itself
    // conditional and serialise the 24 round trips
    const float qv = qb[(int64_t)(d < dk ? d : 0) * T + icl];
    qr[u] = (qv / qdiv) * (d < dk ? 1.f : 0.f);  // query / math.sqrt(k_channels)
  }
  const float mi = mb[icl];
  float m_run = -INFINITY, l_run = 0.f;
  f32x16a o[NDB];
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

  for (int jb = wave * 32; jb < T; jb += 128) {
    const int jcl = min(jb + col, T - 1);
    // the tile's 32 key-mask values as one bit mask (one coalesced load instead of 16 per lane)
    const unsigned kbits = (unsigned)__ballot(mb[jcl] != 0.f);
    f32x16a sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.f;
#pragma unroll
    for (int u0 = 0; u0 < NKS; u0 += 8) {
      float a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int d = 2 * (u0 + u) + half;
        a[u] = kb[(int64_t)(d < dk ? d : 0) * T + jcl];  // d >= dk meets qr = 0
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], qr[u0 + u], sc, 0, 0, 0);
    }
    float p[16];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = (r & 3) + 8 * (r >> 2) + 4 * half;
      float x = sc[r];
      if (mi == 0.f || !((kbits >> jl) & 1u)) x = -1e4f;  // masked_fill(mask == 0, -1e4)
      if (jb + jl >= T) x = -INFINITY;
      p[r] = x;
      tmax = fmaxf(tmax, x);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);  // finite: key jb < T is in this tile
    const float scale = expf(m_run - m_new);  // exp(-inf) = 0 on the first tile
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = expf(p[r] - m_new);
      psum += p[r];
    }
    psum += __shfl_xor(psum, 32);
    l_run = l_run * scale + psum;
    m_run = m_new;
#pragma unroll
    for (int db = 0; db < NDB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= scale;
#pragma unroll
    for (int s0 = 0; s0 < 16; s0 += 8) {
      float av[NDB][8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int r = s0 + u;
        const int jj = min(jb + (r & 3) + 8 * (r >> 2) + 4 * half, T - 1);  // p = 0 beyond T
#pragma unroll
        for (int db = 0; db < NDB; ++db) {
          const int d = db * 32 + col;
          av[db][u] = vb[(int64_t)jj * dk + (d < dk ? d : 0)] * (d < dk ? 1.f : 0.f);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int db = 0; db < NDB; ++db)
          o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[db][u], p[s0 + u], o[db], 0, 0, 0);
    }
  }

  // merge the four key ranges: o = sum_w e_w o_w / sum_w e_w l_w,  e_w = exp(m_w - M)
  if (half == 0) {
    sm_m[wave][col] = m_run;
    sm_l[wave][col] = l_run;
  }
#pragma unroll
  for (int db = 0; db < NDB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) sm_o[((wave * NR) + db * 16 + r) * 64 + lane] = o[db][r];
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int w = 0; w < 4; ++w) M = fmaxf(M, sm_m[w][col]);
  float e[4], L = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    e[w] = expf(sm_m[w][col] - M);  // a wave without keys has m = -inf, l = 0, o = 0
    L += e[w] * sm_l[w][col];
  }
  const float inv = 1.f / L;
#pragma unroll
  for (int q4 = 0; q4 < NR / 4; ++q4) {
    const int rr = wave * (NR / 4) + q4;
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) acc += e[w] * sm_o[((w * NR) + rr) * 64 + lane];
    const int r = rr & 15;
    const int dd = (rr >> 4) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (iok && dd < dk) out[((int64_t)bh * dk + dd) * T + i] = acc * inv;
  }
}

static int attn_flash_enabled() { return 1; }  // (the A/B against separate scores / softmax / P.V kernels is settled: profiles/r01_ab_attn_flash.txt)

int64_t attn_score_elems(int window, int dk, int B, int n_heads, int T) {
  const bool mfma_path = window < 0 || T >= 64;
  if (mfma_path && window < 0 && dk <= 48 && attn_flash_enabled()) return 0;  // flash: no S in HBM
  return (int64_t)B * n_heads * T * T;
}

struct AttnSmallLayout {
  int Tp, Tk, SS;   // keys padded to 4; + the relative columns padded to 4; row stride of S (4 x odd: conflict-free
  size_t lds;       // 16-byte reads across the 32 query lanes)
};
__host__ __device__ static inline AttnSmallLayout attn_small_layout(int T, int dk, int window) {
  AttnSmallLayout L;
  const int nrel = 2 * window + 1;
  L.Tp = (T + 3) & ~3;
  L.Tk = L.Tp + ((nrel + 3) & ~3);
  L.SS = L.Tk + (((L.Tk >> 2) & 1) ? 0 : 4);
  L.lds = ((size_t)dk * 32 + (size_t)2 * dk * L.Tk + (size_t)32 * L.SS + (size_t)32 * 36) * sizeof(float);
  return L;
}

// ---------------------------------------------------------------------------------------------
// Short sequences (T <= 128: a sentence's phonemes): the WHOLE windowed attention of one head for a strip of 32
// queries in ONE launch -- scores (+ the banded relative-key term), mask, softmax, P.V (+ the banded relative-value
// term) -- where the general path takes six (relk table, scores, softmax, v transpose, P.V, relv add: 53 us per
// encoder layer at B = 1, T = 64, each launch a few microseconds of work on a handful of CUs,
// profiles/r02_b1_anatomy.txt).  k, v and the query strip are staged in LDS once; lanes run along the query index,
// so q / P reads are conflict-free and k / v / E reads are broadcasts.  Exact f32, the scalar kernels' formulas.
// grid (ceil(T/32), B*H), 1024 threads = 32 queries x 32 groups of four columns; LDS per attn_small_layout().
// (256-thread blocks with 4-byte LDS reads took 48 us a launch at T = 64: 2000 dependent LDS reads per thread and
// one wave per SIMD to hide them, profiles/r03_b1_anatomy.txt)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void attn_small_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const float* __restrict__ mask, const float* __restrict__ emb_rel_k, const float* __restrict__ emb_rel_v,
    int window, int n_heads, int dk, int T, float qdiv, int64_t qbs, float* __restrict__ out) {
  extern __shared__ float sm_a[];
  constexpr int QT = 32, NG = 32, NTH = QT * NG, RS = NG + 4;
  const AttnSmallLayout L = attn_small_layout(T, dk, window);
  const int nrel = 2 * window + 1, Tp = L.Tp, Tk = L.Tk, SS = L.SS;
  float* qs = sm_a;                        // [dk][QT]   q / sqrt(dk)
  float* kk = qs + (size_t)dk * QT;        // [dk][Tk]   keys, then E_k as columns Tp.. (the relative-key logits are
  float* vv = kk + (size_t)dk * Tk;        // [dk][Tk]   nrel more dot products of the same loop); values, then E_v
  float* S = vv + (size_t)dk * Tk;         // [QT][SS]   rel logits, then P with the band of P in columns Tp..
  float* red = S + (size_t)QT * SS;        // [QT][RS]   per-group max, then per-group sum
  const int tid = threadIdx.x;
  const int il = tid & (QT - 1), grp = tid >> 5;
  const int i0 = blockIdx.x * QT, i = i0 + il;
  const int bh = blockIdx.y, b = bh / n_heads, h = bh % n_heads;
  const float* qb = q + (int64_t)b * qbs + (int64_t)h * dk * T;
  const float* kb = k + (int64_t)b * qbs + (int64_t)h * dk * T;
  const float* vb = v + (int64_t)b * qbs + (int64_t)h * dk * T;
  const float* mb = mask + (int64_t)b * T;
  // ---- stage (all loads of a thread issued before its first LDS store: one memory round trip) --------------------
  {
    const bool vec = ((T & 3) == 0) && (((uintptr_t)kb | (uintptr_t)vb) & 15) == 0;
    if (vec) {
      const float4* k4 = reinterpret_cast<const float4*>(kb);
      const float4* v4 = reinterpret_cast<const float4*>(vb);
      const int tq = T >> 2, n4 = dk * tq;
      for (int e0 = tid; e0 < n4; e0 += NTH * 2) {
        float4 a[2], c[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = min(e0 + NTH * u, n4 - 1);  // clamped: unconditional loads
          a[u] = k4[e];
          c[u] = v4[e];
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int e = e0 + NTH * u;
          if (e < n4) {
            const int d = e / tq, jq = e - d * tq;
            *reinterpret_cast<float4*>(kk + (size_t)d * Tk + 4 * jq) = a[u];
            *reinterpret_cast<float4*>(vv + (size_t)d * Tk + 4 * jq) = c[u];
          }
        }
      }
    } else {
      for (int e = tid; e < dk * T; e += NTH) {
        const int d = e / T, j = e - d * T;
        kk[(size_t)d * Tk + j] = kb[e];
        vv[(size_t)d * Tk + j] = vb[e];
      }
    }
    // columns T..Tk-1: zero padding up to Tp, E_k / E_v (transposed) behind it, zero again up to Tk
    const int nx = Tk - T;
    for (int e = tid; e < dk * nx; e += NTH) {
      const int d = e / nx, c = T + (e - d * nx), r = c - Tp;
      const bool isrel = r >= 0 && r < nrel;
      kk[(size_t)d * Tk + c] = isrel ? emb_rel_k[(size_t)r * dk + d] : 0.f;
      vv[(size_t)d * Tk + c] = isrel ? emb_rel_v[(size_t)r * dk + d] : 0.f;
    }
    for (int e = tid; e < dk * QT; e += NTH) {
      const int d = e / QT, ii = i0 + (e & (QT - 1));
      qs[e] = ii < T ? qb[(int64_t)d * T + ii] / qdiv : 0.f;
    }
  }
  __syncthreads();
  // ---- scores: a thread owns four consecutive columns (one 16-byte LDS read feeds four chains).  Key columns stay
  // in registers for the softmax below; relative-key columns go to S for the threads whose band they fall in ------
  float xq[4] = {0.f, 0.f, 0.f, 0.f};
  const int nkq = Tp >> 2;  // <= NG: every key quad is some group's first unit
  for (int unit = grp; unit < (Tk >> 2); unit += NG) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const float* kr = kk + 4 * unit;
#pragma unroll 8
    for (int d = 0; d < dk; ++d) {
      const float qv = qs[d * QT + il];
      const float4 k4 = *reinterpret_cast<const float4*>(kr + (size_t)d * Tk);
      acc[0] += qv * k4.x;
      acc[1] += qv * k4.y;
      acc[2] += qv * k4.z;
      acc[3] += qv * k4.w;
    }
    if (unit < nkq) {
#pragma unroll
      for (int u = 0; u < 4; ++u) xq[u] = acc[u];
    } else {
      *reinterpret_cast<float4*>(S + (size_t)il * SS + 4 * unit) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
  }
  __syncthreads();
  // ---- softmax over the keys: exp(x - max) / sum, groups merged through LDS ------------------------------
  const bool own = grp < nkq;
  const float mi = i < T ? mb[i] : 0.f;
  float mx = -INFINITY;
  if (own) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = 4 * grp + u;
      float a = xq[u];
      if (j < T) {
        const int r = j - i;
        if (r >= -window && r <= window) a += S[(size_t)il * SS + Tp + r + window];
        if (mi * mb[j] == 0.f) a = -1e4f;  // masked_fill(mask == 0, -1e4)
      } else {
        a = -INFINITY;
      }
      xq[u] = a;
      mx = fmaxf(mx, a);
    }
  }
  red[il * RS + grp] = mx;
  __syncthreads();
  float M = -INFINITY;
#pragma unroll
  for (int g = 0; g < NG; g += 4) {
    const float4 m4 = *reinterpret_cast<const float4*>(red + il * RS + g);
    M = fmaxf(fmaxf(fmaxf(M, m4.x), fmaxf(m4.y, m4.z)), m4.w);
  }
  float sum = 0.f;
  if (own) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xq[u] = expf(xq[u] - M);  // exp(-inf) = 0 for the padding columns
      sum += xq[u];
    }
  }
  __syncthreads();  // every thread has read the maxima
  red[il * RS + grp] = sum;
  __syncthreads();
  float Z = 0.f;
#pragma unroll
  for (int g = 0; g < NG; g += 4) {
    const float4 s4 = *reinterpret_cast<const float4*>(red + il * RS + g);
    Z += (s4.x + s4.y) + (s4.z + s4.w);
  }
  if (own)
    *reinterpret_cast<float4*>(S + (size_t)il * SS + 4 * grp) =
        make_float4(xq[0] / Z, xq[1] / Z, xq[2] / Z, xq[3] / Z);
  __syncthreads();
  // band of P behind the keys: column Tp + r + w holds P[i + r] (zero outside the sequence), so the relative-value
  // term below is the same loop as P.V over the E_v columns
  for (int c = grp; c < Tk - Tp; c += NG) {
    const int j = i + c - window;
    S[(size_t)il * SS + Tp + c] = (c < nrel && j >= 0 && j < T) ? S[(size_t)il * SS + j] : 0.f;
  }
  __syncthreads();
  // ---- out[d][i] = sum_j P[j][i] v[d][j] + sum_r P[i+r][i] E_v[r+w][d], four channels and four keys a step ----------
  if (i < T) {
    const float* pr = S + (size_t)il * SS;
    for (int db = grp; db < dk; db += 4 * NG) {
      const float* vr[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) vr[u] = vv + (size_t)min(db + u * NG, dk - 1) * Tk;
      float acc[4] = {0.f, 0.f, 0.f, 0.f}, rel[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
      for (int j = 0; j < Tp; j += 4) {
        const float4 p4 = *reinterpret_cast<const float4*>(pr + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 v4 = *reinterpret_cast<const float4*>(vr[u] + j);
          acc[u] += p4.x * v4.x;
          acc[u] += p4.y * v4.y;
          acc[u] += p4.z * v4.z;
          acc[u] += p4.w * v4.w;
        }
      }
      for (int j = Tp; j < Tk; j += 4) {
        const float4 p4 = *reinterpret_cast<const float4*>(pr + j);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float4 v4 = *reinterpret_cast<const float4*>(vr[u] + j);
          rel[u] += p4.x * v4.x;
          rel[u] += p4.y * v4.y;
          rel[u] += p4.z * v4.z;
          rel[u] += p4.w * v4.w;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int d = db + u * NG;
        if (d < dk) out[((int64_t)bh * dk + d) * T + i] = acc[u] + rel[u];
      }
    }
  }
}

// longest sequence the one-launch kernel covers (longer ones take the multi-kernel paths)
static int attn_small_max_t() { return 128; }

int32_t k_rel_attention(const float* q, const float* k, const float* v, int64_t qkv_batch_stride,
                        const float* mask, const float* emb_rel_k, const float* emb_rel_v,
                        int window, int B, int n_heads, int dk, int T, float* scores, float* out,
                        hipStream_t s) {
  const int64_t qbs = qkv_batch_stride;
  if (B * T == 0) return WETTS_OK;
  WETTS_REQUIRE(T <= 65535, "attention length %d too large", T);
  const float qdiv = (float)sqrt((double)dk);
  if (window >= 0 && T <= min(attn_small_max_t(), 128)) {  // 32 groups x 4 keys
    const size_t lds = attn_small_layout(T, dk, window).lds;
    static signed char opt_in[64] = {};
    // (a device that refuses the large-LDS opt-in takes the general path below)
    if (lds <= 150 * 1024 && (lds <= 64 * 1024 || lds_opt_in((const void*)attn_small_kernel, opt_in))) {
      hipLaunchKernelGGL(attn_small_kernel, dim3(cdiv(T, 32), B * n_heads), dim3(1024), lds, s, q, k, v, mask,
                         emb_rel_k, emb_rel_v, window, n_heads, dk, T, qdiv, qbs, out);
      WETTS_LAUNCH_CHECK();
      return WETTS_OK;
    }
  }
  if (window < 0 || T >= 64) {
    // Matrix-core path: window-less attention (VITS2 flow encoders) and every relative-position
    // attention long enough to fill 32x32 tiles.  The relative-key term is a [2w+1] x T table
    // added inside the band by the score epilogue, the relative-value term a 2w+1-tap pass over
    // P after the P.V contraction.  Behind the scores the workspace holds the transposed v
    // (B*H*dk*T floats) and that table (B*H*(2w+1)*T).
    float* vT = scores + attn_score_elems(window, dk, B, n_heads, T);
    // (heads wider than 48 channels -- none in the reference's configs -- keep the three-kernel path)
    if (window < 0 && dk <= 48 && attn_flash_enabled()) {
      const int64_t nv = (int64_t)B * n_heads * dk * T;
      hipLaunchKernelGGL(attn_transpose_v_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0,
                         s, v, n_heads, dk, T, qbs, vT, nv);
      WETTS_LAUNCH_CHECK();
      hipLaunchKernelGGL((attn_flash_kernel<24, 2>), dim3(cdiv(T, 32), B * n_heads), dim3(256), 0,
                         s, q, k, vT, mask, n_heads, dk, T, qdiv, qbs, out);
      WETTS_LAUNCH_CHECK();
      return WETTS_OK;
    }
    float* rel = nullptr;
    const int nrel = 2 * window + 1;
    if (window >= 0) {
      rel = vT + (int64_t)B * n_heads * dk * T;
      hipLaunchKernelGGL(attn_relk_kernel, dim3(cdiv(T, 64), nrel, B * n_heads), dim3(64), 0, s, q,
                         emb_rel_k, nrel, n_heads, dk, T, qdiv, qbs, rel);
      WETTS_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(attn_scores_mfma_kernel, dim3(cdiv(T, 32), cdiv(T, 128), B * n_heads),
                       dim3(256), 0, s, q, k, mask, rel, window < 0 ? 0 : window, n_heads, dk, T,
                       qdiv, qbs, scores);
    WETTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_softmax_kernel, dim3(cdiv(T, 16), B * n_heads), dim3(256), 0, s, T, scores);
    WETTS_LAUNCH_CHECK();
    const int64_t nv = (int64_t)B * n_heads * dk * T;
    hipLaunchKernelGGL(attn_transpose_v_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s,
                       v, n_heads, dk, T, qbs, vT, nv);
    WETTS_LAUNCH_CHECK();
    hipLaunchKernelGGL(attn_pv_mfma_kernel, dim3(cdiv(T, 128), cdiv(dk, 32), B * n_heads),
                       dim3(256), 0, s, scores, vT, dk, T, out);
    WETTS_LAUNCH_CHECK();
    if (window >= 0) {
      WETTS_REQUIRE(window <= kRelvMaxW, "relative attention window %d > %d", window, kRelvMaxW);
      hipLaunchKernelGGL(attn_relv_add_kernel, dim3(cdiv(T, 64), B * n_heads), dim3(256),
                         (size_t)(2 * window + 1) * dk * sizeof(float), s, scores, emb_rel_v, window, dk, T, out);
      WETTS_LAUNCH_CHECK();
    }
    return WETTS_OK;
  }
  int tb = cdiv(T, 64);
  hipLaunchKernelGGL(attn_scores_kernel, dim3(tb, T, B * n_heads), dim3(64), 0, s, q, k, mask,
                     emb_rel_k, window, n_heads, dk, T, qdiv, qbs, scores);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_softmax_kernel, dim3(cdiv(T, 16), B * n_heads), dim3(256), 0, s, T, scores);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_pv_kernel, dim3(tb, dk, B * n_heads), dim3(64), 0, s, scores, v,
                     emb_rel_v, window, n_heads, dk, T, qbs, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
