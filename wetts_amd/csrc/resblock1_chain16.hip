// A whole ResBlock1 of the 16-bit decoder in ONE launch (C <= 64, short kernels):
//   for p in pairs:  x = x + c2_p(lrelu(c1_p(lrelu(x))))        out = (x [+ out]) / div
// (reference decoders.py:157-170, ResBlock1.forward) -- the 16-bit twin of resblock_chain32.hip.
//
// One pair per launch (resblock16.hip) moves x in and x out through HBM for every pair, and at C <= 64 with k = 3 a
// block's life is mostly the fixed latency of a tile (HBM round trip of the staging loads, two epilogues, the store:
// profiles/r03_pair16_phase_timeline.txt -- the two MFMA loops are a quarter of it).  Here the block stages its x tile
// once, runs all the pairs over it in LDS / registers and stores once: x read once, the MRF sum written once (SURVEY
// 8(d) "resblock-fused" bytes), one staging and one store phase instead of three.
//
// Every conv of the chain uses the SAME column -> time mapping (column c <-> time n0 - H + c, LDS row c + hm), so the
// value a lane's accumulators hold after c2_p -- x_{p+1} at its own (channels, columns) -- is exactly what it needs as
// the residual of pair p+1: it stays in registers (packed 16-bit), and lrelu(x_{p+1}) goes back to the same LDS rows
// for c1_{p+1}.  A conv with halo h reads rows c + hm - h .. c + hm + h; columns within h of the tile edge see rows
// that were not updated for them, so the correct region shrinks by the conv's halo on each side per conv and the
// block's valid output is the middle NTO = NTC - 2 H columns, H = sum_p (h1_p + h2).  (2 H / NTC of the MFMA work is
// redone by the neighbours: 9 % at C = 64, k = 3; the host only takes shapes under a waste limit.)
//
// Arithmetic -- operation order, rounding points, zero padding of every conv's input outside [0, T) -- is that of the
// pair kernel launched once per pair, so the two paths agree bit for bit (tests/test_gpu_parity.py).
#include <stdlib.h>

#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

template <int C, bool F16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3)))
void resblock1_chain16_kernel(const ResChain16Params p) {
  constexpr int WM = C / 32, WN = 4 / WM, NB = 4;
  constexpr int NTH = 256;
  constexpr int NTC = 32 * NB * WN;
  constexpr int CKB = C >= 64 ? 64 : 32;
  constexpr int KS = CKB / 16;
  constexpr int SEG = C / 8;
  constexpr int RS = C * 2 + 16;
  constexpr int MAXU = ((NTC + 2 * RESCHAIN16_MAX_HALO) * SEG + NTH - 1) / NTH;
  static_assert(C / CKB == 1, "one K chunk");
  static_assert(NTH % SEG == 0, "piece index must not depend on the unit");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  const int h2 = (p.ktaps - 1) / 2;
  const int H = p.halo, hm = p.margin;
  const int NTO = NTC - 2 * H;
  int bid = blockIdx.x;
  {  // XCD-aware tile order (see resblock16.hip)
    const int per = (p.nblocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.nblocks) return;
  }
  const int ntile = bid % p.ntiles;
  const int b = bid / p.ntiles;
  const int n0 = ntile * NTO;
  const int W = NTC + 2 * hm;
  const int tx0 = n0 - H - hm;  // time of LDS row 0

  const unsigned short* xb = p.x + (int64_t)b * p.T * C;
  const int G = p.ktaps;
  const int64_t aoff = ((int64_t)wm * G * KS) * 64 + lane;
  // A fragments of one (tap) group; each k-step's registers are refilled for the next group as soon as its MFMAs
  // are issued (an in-place ring: KS - 1 k-steps of MFMA work cover the L2 round trip, and the registers a second
  // group would take are what keeps this kernel at three waves per SIMD)
  uint4 aa[KS];
  auto a_prologue = [&](const uint4* abase) {
#pragma unroll
    for (int s = 0; s < KS; ++s) aa[s] = abase[(int64_t)s * 64];
  };
  a_prologue(reinterpret_cast<const uint4*>(p.wpk1[0]) + aoff);

  const int co_blk = wm * 32;
  const int wcol = wn * (32 * NB) + (lane & 31);
  // ---- stage lrelu(x), zero outside [0, T) ------------------------------------------------------
  {
    const int useg = tid % SEG;
    uint4 st[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int row = (tid + NTH * i) / SEG;
      const int t = tx0 + row;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (row < W && t >= 0 && t < p.T) v = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + useg * 8);
      st[i] = v;
    }
#pragma unroll
    for (int i = 0; i < MAXU; ++i) {
      const int row = (tid + NTH * i) / SEG;
      if (row < W) {
        uint4 v = st[i];
        v.x = lrelu_pk<F16>(v.x, p.slope); v.y = lrelu_pk<F16>(v.y, p.slope);
        v.z = lrelu_pk<F16>(v.z, p.slope); v.w = lrelu_pk<F16>(v.w, p.slope);
        *reinterpret_cast<uint4*>(smem_c + (size_t)row * RS + useg * 16) = v;
      }
    }
  }
  // raw x at this lane's (channels, columns): the residual of pair 0 (L2 hits: the rows were staged a moment ago;
  // requested after the staging registers are dead)
  uint4 rres[NB][2];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int t = n0 - H + wcol + 32 * j;
    const bool ok = t >= 0 && t < p.T;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (ok) v = *reinterpret_cast<const uint4*>(xb + (int64_t)t * C + co_blk + 16 * i + 8 * half);
      rres[j][i] = v;
    }
  }

  __syncthreads();

  f32x16 acc[NB];
  // rows of this lane's columns (n-block 0): a conv with halo h starts h rows above row wcol + hm
  const unsigned char* bmid = smem_c + (size_t)(wcol + hm) * RS + half * 16;

  uint4 bq[NB];  // one buffer: a second one (B reads a k-step ahead) costs the registers of the third wave per SIMD
  // one conv over the tile: tap t reads rows shifted by (t - (k-1)/2) * dil.  The refill of a k-step's A registers
  // is issued UNCONDITIONALLY (the last group re-reads itself): exact s_waitcnt counts (see resblock16.hip)
  auto conv_loop = [&](const uint4* abase, int dil, int h) {
    const unsigned char* b0 = bmid - (size_t)h * RS;
    for (int g = 0; g < G; ++g) {
      const int gn = g + 1 < G ? g + 1 : g;
      const unsigned char* cur = b0 + (size_t)(g * dil) * RS;
      const uint4* an = abase + (int64_t)gn * KS * 64;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
#pragma unroll
        for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const uint4*>(cur + (size_t)(32 * j) * RS + s * 32);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma16<F16>(aa[s], bq[j], acc[j]);
        aa[s] = an[(int64_t)s * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  unsigned short* ob = p.out + (int64_t)b * p.T * C;
  for (int pr = 0; pr < p.npairs; ++pr) {
    const bool last = pr == p.npairs - 1;
    const int dil = p.dil[pr], h1 = h2 * dil;
    // ---- c1 ------------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    conv_loop(reinterpret_cast<const uint4*>(p.wpk1[pr]) + aoff, dil, h1);
    a_prologue(reinterpret_cast<const uint4*>(p.wpk2[pr]) + aoff);  // lands during the epilogue
    // ---- ft = lrelu(round16(c1 + b1)) over the tile (zero outside [0, T): c2 pads ITS input) ---------------
    {
      float bia[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[r] = p.bias1[pr][co_blk + 16 * (r >> 3) + 8 * half + (r & 7)];
      __syncthreads();  // every wave has finished reading the tile
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = wcol + 32 * j;
        const int t = n0 - H + col;
        const bool inside = t >= 0 && t < p.T;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned r16 = pk2<F16>(acc[j][8 * i + 2 * e] + bia[8 * i + 2 * e],
                                          acc[j][8 * i + 2 * e + 1] + bia[8 * i + 2 * e + 1]);
            w[e] = inside ? lrelu_pk<F16>(r16, p.slope) : 0u;
          }
          *reinterpret_cast<uint4*>(smem_c + (size_t)(col + hm) * RS + (co_blk + 16 * i + 8 * half) * 2) =
              make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      __syncthreads();
    }
    // ---- c2, accumulator = residual (+ the running MRF sum on the last pair) -------------------------------
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol + 32 * j;
      const int t = n0 - H + col;
      const bool ok = col >= H && col < NTC - H && t < p.T;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned w4[4] = {rres[j][i].x, rres[j][i].y, rres[j][i].z, rres[j][i].w};
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] = lo16<F16>(w4[e]);
          v[2 * e + 1] = hi16<F16>(w4[e]);
        }
        if (last && p.accum && ok) {
          const uint4 oo = *reinterpret_cast<const uint4*>(ob + (int64_t)t * C + co_blk + 16 * i + 8 * half);
          const unsigned o4[4] = {oo.x, oo.y, oo.z, oo.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[2 * e] += lo16<F16>(o4[e]);
            v[2 * e + 1] += hi16<F16>(o4[e]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][8 * i + e] = v[e];
      }
    }
    conv_loop(reinterpret_cast<const uint4*>(p.wpk2[pr]) + aoff, 1, h2);
    float bia[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bia[r] = p.bias2[pr][co_blk + 16 * (r >> 3) + 8 * half + (r & 7)];
    if (!last) {
      // x_{p+1} = round16(x_p + c2 + b2): the next residual (registers) and, through lrelu, the next c1's input
      a_prologue(reinterpret_cast<const uint4*>(p.wpk1[pr + 1]) + aoff);
      __syncthreads();  // every wave has finished reading ft
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = wcol + 32 * j;
        const int t = n0 - H + col;
        const bool inside = t >= 0 && t < p.T;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned r16[4], w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r16[e] = pk2<F16>(acc[j][8 * i + 2 * e] + bia[8 * i + 2 * e],
                              acc[j][8 * i + 2 * e + 1] + bia[8 * i + 2 * e + 1]);
            if (!inside) r16[e] = 0u;  // what the next launch would read from beyond the sequence
            w[e] = lrelu_pk<F16>(r16[e], p.slope);
          }
          rres[j][i] = make_uint4(r16[0], r16[1], r16[2], r16[3]);
          *reinterpret_cast<uint4*>(smem_c + (size_t)(col + hm) * RS + (co_blk + 16 * i + 8 * half) * 2) =
              make_uint4(w[0], w[1], w[2], w[3]);
        }
      }
      __syncthreads();
    } else {
      const bool dodiv = p.out_div != 1.f;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = wcol + 32 * j;
        const int t = n0 - H + col;
        if (col < H || col >= NTC - H || t >= p.T) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] = acc[j][8 * i + e] + bia[8 * i + e];
            if (dodiv) v[e] = v[e] / p.out_div;
          }
          uint4 o;
          o.x = pk2<F16>(v[0], v[1]); o.y = pk2<F16>(v[2], v[3]);
          o.z = pk2<F16>(v[4], v[5]); o.w = pk2<F16>(v[6], v[7]);
          *reinterpret_cast<uint4*>(ob + (int64_t)t * C + co_blk + 16 * i + 8 * half) = o;
        }
      }
    }
  }
}

static int chain16_ntc(int C) { return 128 * (4 / (C / 32)); }

static bool chain16_geometry(const PackedConvB* c1, const PackedConvB* c2, int npairs, int* halo, int* margin) {
  if (npairs < 1 || npairs > RESCHAIN16_MAX_PAIRS) return false;
  const int C = c1[0].Cin, k = c1[0].ktaps;
  if (!(C == 32 || C == 64) || (k & 1) == 0) return false;
  int H = 0, hm = (k - 1) / 2;
  for (int q = 0; q < npairs; ++q) {
    const PackedConvB &a = c1[q], &b = c2[q];
    if (a.Cin != C || a.Cout != C || b.Cin != C || b.Cout != C || a.up || b.up) return false;
    if (a.ktaps != k || b.ktaps != k || b.dil != 1 || a.f16 != c1[0].f16 || b.f16 != c1[0].f16) return false;
    if (a.pad != (k - 1) / 2 * a.dil || b.pad != (k - 1) / 2) return false;
    if (!a.wpk || !b.wpk) return false;
    const int h1 = (k - 1) / 2 * a.dil;
    H += h1 + (k - 1) / 2;
    if (h1 > hm) hm = h1;
  }
  if (hm > RESCHAIN16_MAX_HALO) return false;
  *halo = H;
  *margin = hm;
  return true;
}

// valid output columns per block, or 0 when the chain kernel does not take the shape
int resblock1_chain16_nto(const PackedConvB* c1, const PackedConvB* c2, int npairs, int max_waste_pct) {
  int H = 0, hm = 0;
  if (!chain16_geometry(c1, c2, npairs, &H, &hm)) return 0;
  const int NTC = chain16_ntc(c1[0].Cin);
  if (2 * H * 100 > max_waste_pct * NTC) return 0;
  return NTC - 2 * H;
}

int32_t launch_resblock1_chain16(const PackedConvB* c1, const PackedConvB* c2, int npairs, ResChain16Params p,
                                 hipStream_t stream) {
  int H = 0, hm = 0;
  WETTS_REQUIRE(chain16_geometry(c1, c2, npairs, &H, &hm), "ResBlock1 chain shape not supported by the 16-bit chain kernel");
  const int C = c1[0].Cin, NTC = chain16_ntc(C), NTO = NTC - 2 * H;
  WETTS_REQUIRE(NTO > 0, "chain halo exceeds the tile");
  p.npairs = npairs;
  p.ktaps = c1[0].ktaps;
  p.halo = H;
  p.margin = hm;
  for (int q = 0; q < npairs; ++q) {
    p.wpk1[q] = c1[q].wpk; p.bias1[q] = c1[q].bias;
    p.wpk2[q] = c2[q].wpk; p.bias2[q] = c2[q].bias;
    p.dil[q] = c1[q].dil;
  }
  p.ntiles = cdiv(p.T, NTO);
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock grid too large");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  const size_t lds = (size_t)(NTC + 2 * hm) * (C * 2 + 16);
  const bool f16 = c1[0].f16 != 0;
  if (C == 32) {
    if (f16) hipLaunchKernelGGL((resblock1_chain16_kernel<32, true>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((resblock1_chain16_kernel<32, false>), dim3(grid), dim3(256), lds, stream, p);
  } else {
    if (f16) hipLaunchKernelGGL((resblock1_chain16_kernel<64, true>), dim3(grid), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((resblock1_chain16_kernel<64, false>), dim3(grid), dim3(256), lds, stream, p);
  }
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
