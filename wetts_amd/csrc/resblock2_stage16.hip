// A whole MRF stage of ResBlock2 blocks in ONE launch (16-bit decoder; the v3 recipes):
//
//   out = ( rb_0(x) + rb_1(x) + ... ) / n,   rb_j(x) = t + c2_j(lrelu(t)),  t = x + c1_j(lrelu(x))
//
// (reference decoders.py:63-82 `xs += resblock(x); x = xs / num_kernels` with ResBlock2.forward, decoders.py:205-214).
// Chain by chain (resblock16.hip, RB2) a stage moves x in three times and the running sum in and out three times:
// at C = 32 that is 6.1 GB for 0.8 GB of input, and the class sat at 2.6 TB/s of real traffic.  Here a block keeps
// the running sum of its time tile in registers across the chains: x is read from HBM once (the later chains' tiles
// are L2 hits of rows this block staged a moment ago), the sum is written once.
//
// Per chain the block does exactly what resblock_pair16_kernel<RB2> does -- stage lrelu(x) channel-last into LDS, c1
// on the matrix cores with the raw x as accumulator init, t rounded to 16 bit and kept in the accumulators,
// lrelu(t) written h2 rows down the tile, c2 with accumulator init t (+ the running sum), + bias, (/ n on the last
// chain), round -- with ONE change: every chain uses the same column -> time mapping, column c <-> time
// n0 - o + c with o = the widest c2 halo of the stage, so that the chains' valid columns [h2_j, NTC - h2_j) all
// contain the common output range [o, NTC - o).  Operation order and rounding points per output element are those
// of the chain-by-chain launches, so the results are bit-identical to them
// (tests/test_gpu_parity.py::test_fused_resblock_pair_bit_identical runs both).
#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

template <int C, bool F16, int NR, int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void rb2_stage16_kernel(const ResStage2Params p) {
  constexpr int WM = C / 32, WN = 4 / WM, NB = 4;
  constexpr int NTH = 256;
  constexpr int NTC = 32 * NB * WN;
  constexpr int CKB = C >= 64 ? 64 : 32;
  constexpr int NCH = C / CKB, KS = CKB / 16;
  constexpr int SEG = C / 8;
  constexpr int RS = C * 2 + 16;
  constexpr int MAXU = ((NTC + RESPAIR2_MAX_SPAN32) * SEG + NTH - 1) / NTH;
  static_assert(NTH % SEG == 0, "piece index must not depend on the unit");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  const int o = p.origin;
  const int NTO = NTC - 2 * o;
  int bid = blockIdx.x;
  {  // XCD-aware tile order (see resblock16.hip)
    const int per = (p.nblocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.nblocks) return;
  }
  const int ntile = bid % p.ntiles;
  const int b = bid / p.ntiles;
  const int n0 = ntile * NTO;
  const unsigned short* xb = p.x + (int64_t)b * p.T * C;
  const int co_blk = wm * 32;
  const int wcol = wn * (32 * NB) + (lane & 31);
  const unsigned char* bcol = smem_r + (size_t)wcol * RS + half * 16;

  uint4 sum[NB][2];  // the running MRF sum of this lane's outputs, packed 16 bit (rounded after every chain)
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) sum[j][i] = make_uint4(0u, 0u, 0u, 0u);

  for (int ch = 0; ch < p.nchain; ++ch) {
    const int ktaps = p.ktaps[ch], dil1 = p.dil1[ch], dil2 = p.dil2[ch];
    const int h1 = (ktaps - 1) / 2 * dil1, h2 = (ktaps - 1) / 2 * dil2;
    const int W1 = NTC + 2 * (h1 > h2 ? h1 : h2);
    const int tx0 = n0 - o - h1;  // time of LDS row 0 of the x tile
    const int G = NCH * ktaps;
    const uint4* abase1 = reinterpret_cast<const uint4*>(p.wpk1[ch]) + ((int64_t)wm * G * KS) * 64 + lane;
    const uint4* abase2 = reinterpret_cast<const uint4*>(p.wpk2[ch]) + ((int64_t)wm * G * KS) * 64 + lane;
    uint4 aa[NR][KS];
    auto a_prologue = [&](const uint4* abase) {
#pragma unroll
      for (int r = 0; r < NR - 1; ++r)
#pragma unroll
        for (int s = 0; s < KS; ++s) aa[r][s] = abase[((int64_t)(r < G ? r : 0) * KS + s) * 64];
#pragma unroll
      for (int s = 0; s < KS; ++s) aa[NR - 1][s] = aa[0][s];
    };
    a_prologue(abase1);

    // raw x at c1's columns initialises c1's accumulators (requested first: the oldest load in flight)
    uint4 rres[NB][2];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = n0 - o + wcol + 32 * j;
      const bool ok = t >= 0 && t < p.T;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (ok) v = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + co_blk + 16 * i + 8 * half);
        rres[j][i] = v;
      }
    }

    // ---- 1. stage lrelu(x) (chains after the first: L2 hits) ---------------------------------------------
    if (ch > 0) __syncthreads();  // the previous chain's c2 has finished reading the tile
    {
      const int useg = tid % SEG;
      uint4 st[MAXU];
#pragma unroll
      for (int i = 0; i < MAXU; ++i) {
        const int row = (tid + NTH * i) / SEG;
        const int t = tx0 + row;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < W1 && t >= 0 && t < p.T) v = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + useg * 8);
        st[i] = v;
      }
#pragma unroll
      for (int i = 0; i < MAXU; ++i) {
        const int row = (tid + NTH * i) / SEG;
        if (row < W1) {
          uint4 v = st[i];
          v.x = lrelu_pk<F16>(v.x, p.slope); v.y = lrelu_pk<F16>(v.y, p.slope);
          v.z = lrelu_pk<F16>(v.z, p.slope); v.w = lrelu_pk<F16>(v.w, p.slope);
          *reinterpret_cast<uint4*>(smem_r + (size_t)row * RS + useg * 16) = v;
        }
      }
    }
    __syncthreads();

    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned w4[4] = {rres[j][i].x, rres[j][i].y, rres[j][i].z, rres[j][i].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc[j][8 * i + 2 * e] = lo16<F16>(w4[e]);
          acc[j][8 * i + 2 * e + 1] = hi16<F16>(w4[e]);
        }
      }

    // one conv over the LDS tile (the loop of resblock16.hip at MB = 1: unconditional A prefetch NR - 1 groups
    // ahead, B fragments one k-step ahead)
    uint4 bq[2][NB];
    auto b_load = [&](uint4 (&dst)[NB], const unsigned char* bb, int s) {
#pragma unroll
      for (int j = 0; j < NB; ++j) dst[j] = *reinterpret_cast<const uint4*>(bb + (size_t)(32 * j) * RS + s * 32);
    };
    auto mma_group = [&](const uint4 (&av)[KS], const unsigned char* cur, const unsigned char* nxt) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + 1 < KS) b_load(bq[(s + 1) & 1], cur, s + 1);
        else b_load(bq[(s + 1) & 1], nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma16<F16>(av[s], bq[s & 1][j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto conv_loop = [&](const uint4* abase, int dil, int row0) {
      int chunk = 0, tap = 0, g = 0;
      auto bpos = [&](int tp, int cq) { return bcol + (size_t)(row0 + tp * dil) * RS + cq * (CKB * 2); };
      auto advance = [&](int& tp, int& cq) { if (++tp == ktaps) { tp = 0; ++cq; } };
      b_load(bq[0], bpos(0, 0), 0);
      for (; g + NR <= G; g += NR) {
#pragma unroll
        for (int par = 0; par < NR; ++par) {
          int gn = g + par + NR - 1;
          gn = gn < G ? gn : G - 1;
#pragma unroll
          for (int s = 0; s < KS; ++s) aa[(par + NR - 1) % NR][s] = abase[((int64_t)gn * KS + s) * 64];
          __builtin_amdgcn_sched_barrier(0);
          {
            int ntap = tap, nchunk = chunk;
            advance(ntap, nchunk);
            if (nchunk >= NCH) { ntap = tap; nchunk = chunk; }
            mma_group(aa[par], bpos(tap, chunk), bpos(ntap, nchunk));
          }
          advance(tap, chunk);
        }
      }
#pragma unroll
      for (int par = 0; par < NR - 1; ++par) {
        if (g + par < G) {
          {
            int ntap = tap, nchunk = chunk;
            advance(ntap, nchunk);
            if (nchunk >= NCH) { ntap = tap; nchunk = chunk; }
            mma_group(aa[par], bpos(tap, chunk), bpos(ntap, nchunk));
          }
          advance(tap, chunk);
        }
      }
    };

    // ---- 2. c1: column c <-> time n0 - o + c reads x rows c + tap * dil1 (row 0 <-> time n0 - o - h1) ------
    conv_loop(abase1, dil1, 0);
    a_prologue(abase2);

    // ---- 3. t = round16(c1 + b1) stays in the accumulators; lrelu(t) goes h2 rows down the tile ------------
    {
      float bia[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[r] = p.bias1[ch][co_blk + 16 * (r >> 3) + 8 * half + (r & 7)];
      __syncthreads();  // every wave has finished reading lrelu(x)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = wcol + 32 * j;
        const int t = n0 - o + col;
        const bool inside = t >= 0 && t < p.T;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned r16 = pk2<F16>(acc[j][8 * i + 2 * e] + bia[8 * i + 2 * e],
                                          acc[j][8 * i + 2 * e + 1] + bia[8 * i + 2 * e + 1]);
            acc[j][8 * i + 2 * e] = lo16<F16>(r16);
            acc[j][8 * i + 2 * e + 1] = hi16<F16>(r16);
            // lrelu_pk(r16) on the values just unpacked (same operations, two unpacks less)
            w[e] = inside ? pk2<F16>(lrelu_max(acc[j][8 * i + 2 * e], p.slope), lrelu_max(acc[j][8 * i + 2 * e + 1], p.slope)) : 0u;
          }
          *reinterpret_cast<uint4*>(smem_r + (size_t)(col + h2) * RS + (co_blk + 16 * i + 8 * half) * 2) =
              make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      __syncthreads();
    }

    // ---- 4. c2: accumulator = t (+ the running sum), rows c + tap * dil2 of the shifted tile ---------------
    if (ch > 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned o4[4] = {sum[j][i].x, sum[j][i].y, sum[j][i].z, sum[j][i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[j][8 * i + 2 * e] += lo16<F16>(o4[e]);
            acc[j][8 * i + 2 * e + 1] += hi16<F16>(o4[e]);
          }
        }
    }
    conv_loop(abase2, dil2, 0);

    // ---- 5. + b2 (/ n on the last chain), round: the new running sum ---------------------------------------
    {
      float bia[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[r] = p.bias2[ch][co_blk + 16 * (r >> 3) + 8 * half + (r & 7)];
      const bool dodiv = ch == p.nchain - 1 && p.out_div != 1.f;
      const bool fastdiv = p.out_div == 3.f || p.out_div == 2.f;  // (num_kernels of every recipe: 3)
      const float dinv = 1.f / p.out_div;
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[j][8 * i + e] + bia[8 * i + e];
            if (dodiv) v[e] = fastdiv ? div_small_const(v[e], p.out_div, dinv) : v[e] / p.out_div;
          }
          sum[j][i].x = pk2<F16>(v[0], v[1]); sum[j][i].y = pk2<F16>(v[2], v[3]);
          sum[j][i].z = pk2<F16>(v[4], v[5]); sum[j][i].w = pk2<F16>(v[6], v[7]);
        }
    }
  }

  // ---- store the common valid columns [o, NTC - o) -------------------------------------------------------------
  unsigned short* ob = p.out + (int64_t)b * p.T * C;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int col = wcol + 32 * j;
    const int t = n0 - o + col;
    if (col < o || col >= NTC - o || t < 0 || t >= p.T) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      *reinterpret_cast<uint4*>(ob + (int64_t)t * C + co_blk + 16 * i + 8 * half) = sum[j][i];
  }
}

// valid output columns per block, or 0 when the stage is not covered (shapes, halos, waste above max_waste_pct)
int resblock2_stage16_nto(const PackedConvB* const* c1, const PackedConvB* const* c2, int nchain, int max_waste_pct) {
  if (nchain < 1 || nchain > RESSTAGE2_MAX_CHAINS) return 0;
  const int C = c1[0]->Cin;
  if (C != 32 && C != 64) return 0;
  int o = 0;
  for (int j = 0; j < nchain; ++j) {
    const PackedConvB &a = *c1[j], &b = *c2[j];
    if (a.Cin != C || a.Cout != C || b.Cin != C || b.Cout != C || a.up || b.up || !a.wpk || !b.wpk) return 0;
    if (a.ktaps != b.ktaps || (a.ktaps & 1) == 0 || a.f16 != c1[0]->f16 || b.f16 != c1[0]->f16) return 0;
    if (a.pad != (a.ktaps - 1) / 2 * a.dil || b.pad != (b.ktaps - 1) / 2 * b.dil) return 0;
    const int h1 = (a.ktaps - 1) / 2 * a.dil, h2 = (b.ktaps - 1) / 2 * b.dil;
    if (2 * (h1 > h2 ? h1 : h2) > RESPAIR2_MAX_SPAN32) return 0;
    if (h2 > o) o = h2;
  }
  const int NTC = 128 * (4 / (C / 32));
  if (2 * o * 100 > max_waste_pct * NTC) return 0;
  return NTC - 2 * o;
}

int32_t launch_resblock2_stage16(const PackedConvB* const* c1, const PackedConvB* const* c2, int nchain,
                                 ResStage2Params p, hipStream_t stream) {
  const int nto = resblock2_stage16_nto(c1, c2, nchain, 100);
  WETTS_REQUIRE(nto > 0, "ResBlock2 stage not supported by the fused kernel");
  const int C = c1[0]->Cin;
  const int NTC = 128 * (4 / (C / 32)), RS = C * 2 + 16;
  p.nchain = nchain;
  int wmax = 0;
  for (int j = 0; j < nchain; ++j) {
    p.wpk1[j] = c1[j]->wpk; p.bias1[j] = c1[j]->bias;
    p.wpk2[j] = c2[j]->wpk; p.bias2[j] = c2[j]->bias;
    p.ktaps[j] = c1[j]->ktaps;
    p.dil1[j] = c1[j]->dil;
    p.dil2[j] = c2[j]->dil;
    const int h1 = (p.ktaps[j] - 1) / 2 * p.dil1[j], h2 = (p.ktaps[j] - 1) / 2 * p.dil2[j];
    if (h1 > wmax) wmax = h1;
    if (h2 > wmax) wmax = h2;
  }
  p.origin = (NTC - nto) / 2;
  p.ntiles = cdiv(p.T, nto);
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock stage grid too large");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  const size_t lds = (size_t)(NTC + 2 * wmax) * RS;
  const bool f16 = c1[0]->f16 != 0;
  if (C == 32) {
    if (f16) hipLaunchKernelGGL((rb2_stage16_kernel<32, true, 2, 2>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((rb2_stage16_kernel<32, false, 2, 2>), dim3(grid), dim3(256), lds, stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((rb2_stage16_kernel<64, true, 2, 2>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((rb2_stage16_kernel<64, false, 2, 2>), dim3(grid), dim3(256), lds, stream, p);
  }
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
