// ResBlock1 chains at float32 in ONE launch: P = 1 (one (c1, c2) pair) or P = 3 (the whole
// ResBlock1, reference decoders.py:157-170) -- the "resblock-fused" traffic figure of SURVEY 8(d):
// x is read once and the block's output written once, every intermediate stays on the CU.
//
//   for p in 0..P-1:   x = x + c2_p(lrelu(c1_p(lrelu(x))))        c1_p at dilation d_p, c2_p at 1
//   out = (x [+ out]) / div
//
// Second generation of resblock32.hip's pair kernel, written around what the round-2
// micro-benchmarks showed (profiles/r02_mfma_valu_coissue.txt, r02_mfma_loop_ingredients.txt,
// r02_pair32_ablation.txt):
//   * the f32 matrix pipe and the vector ALU are one resource on gfx950: every VALU instruction of
//     any wave on the SIMD costs the MFMA stream ~4 cycles.  So the non-MFMA phases are written for
//     instruction count: 16-byte aligned staging loads with uniform row bases, leaky-relu as
//     mul + max (two ops, not three), 32-bit offsets against uniform bases for the residual loads
//     and the stores, no per-element 64-bit address arithmetic, bounds handling only in the two
//     edge tiles of a sequence;
//   * the residual stream lives in accumulator registers for the whole chain: c2 accumulates straight
//     into it (x += c2(..)), so there is no residual re-read and no register copy between pairs;
//   * B fragments are software-pipelined one k-step ahead (pinned by sched_barriers).
//
// Geometry.  All 2P convs share ONE column -> time mapping: accumulator column c of the tile is time
// t0 + c, t0 = n0 - S.  LDS tile [C][Wp]: LDS column L is time t0 - M + L (M = left margin >= the widest
// half-width, chosen per tile so that the staged window starts on a 16-byte boundary).  Conv 0 reads
// real x in both margins, so all NTC of its outputs are exact; every later conv i loses h_i columns
// per side (its inputs there would come from a neighbouring tile), leaving NTO = NTC - 2S valid
// outputs, S = sum_{i>=1} h_i.  Positions outside [0, T) are written as zeros at every stage (each
// conv pads ITS input).  Operation order and rounding points per output element equal the
// conv-by-conv path (same packed weights, same group order), so results are bit-identical to it.
#include <stdlib.h>

#include "common.h"
#include "resblock32.h"

namespace wetts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// Buffer addressing (uniform descriptor + uniform byte offset + one 32-bit lane offset + immediate):
// the address arithmetic of every global access of this kernel happens in the scalar unit / the
// instruction encoding, none of it in the vector ALU the MFMAs need.
constexpr int kBufRsrcDword3 = 0x00020000;  // gfx9 family raw buffer: 32-bit data format, no swizzle
// (the b32 builtins traffic in `unsigned`: bit casts, not value conversions)
__device__ __forceinline__ float buf_load_f32(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
__device__ __forceinline__ void buf_store_f32(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}

// max(x, slope*x) == (x > 0 ? x : slope*x) for 0 <= slope <= 1; v_max_f32 through asm so that no
// canonicalising v_max x, x is put in front of it
__device__ __forceinline__ float lrelu2(float x, float slope) {
  const float m = x * slope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}

// NB = 32-column accumulator blocks per wave.  4: tiles of 128 * WN columns, two blocks per CU (LDS and 177 registers).
// 3: tiles of 96 * WN columns -- 25 % fewer accumulator registers (<= 168: three waves per SIMD) and an LDS tile of at most
// 53 KB, i.e. THREE blocks per CU where the chain's margins allow it: the chain is a sequence of short MFMA runs
// separated by barriers and LDS write passes (k = 3 at C = 32: 192 MFMAs per wave between them), and a third resident
// block is what hides those phases; the price is a larger share of halo columns (2 S of 96 * WN instead of 128 * WN).
template <int C, int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NB == 2 ? 4 : NB == 3 ? 3 : 2, NB == 2 ? 4 : NB == 3 ? 3 : 2)))
void resblock_chain32_kernel(const ResChain32Params p) {
  constexpr int WM = C / 32, WN = 4 / WM;
  constexpr int NTC = 32 * NB * WN;
  constexpr int NCH = C / kConvCK;
  constexpr int RPW = C / 4;                                   // staged rows per wave
  constexpr int QN = (NTC + 2 * RESCHAIN32_MAX_M + 8 + 255) / 256;  // 16-byte pieces per lane per row

  extern __shared__ __attribute__((aligned(16))) float smem_c[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  int bid = blockIdx.x;
  {  // XCD-aware tile order: neighbouring tiles (shared halos) on one XCD's L2
    const int per = (p.nblocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.nblocks) return;
  }
  const int ntile = bid % p.ntiles;
  int b = bid / p.ntiles;
  // ragged batches arrive sorted by length: an XCD's contiguous run of (utterance, tile) pairs would hold the few
  // longest (or shortest) utterances and the XCDs would finish far apart.  Walk the utterances with a stride
  // coprime to B instead, so that every run mixes long and short ones.
  if (p.lens) b = (int)(((int64_t)b * p.bstride) % p.B);
  const int n0 = ntile * p.NTO;
  const int t0 = __builtin_amdgcn_readfirstlane(n0 - p.S);  // time of accumulator column 0
  const int M = p.Mmin + ((t0 - p.Mmin) & 3);      // left margin: (t0 - M) % 4 == 0
  const int tx0 = t0 - M;                          // time of LDS column 0
  const int Wp = p.Wp;
  const int T = p.T;  // row stride of the [C][T] planes
  // Tb: where THIS utterance ends (ragged batches: it is convolved as if alone, zero padding at its own end)
  const int Tb = p.lens ? __builtin_amdgcn_readfirstlane(min((int)p.lens[b] * p.len_mul, T)) : T;
  if (n0 >= Tb) return;
  const bool interior = tx0 >= 0 && tx0 + Wp <= Tb;

  // (b, t0, ... derive from blockIdx through an integer division, which lives in vector registers: the
  // readfirstlane round trips tell the compiler what it cannot prove -- these are uniform -- so that
  // descriptors and scalar offsets go to SGPRs instead of per-instruction waterfall loops)
  const float* xb = p.x + (int64_t)__builtin_amdgcn_readfirstlane(b) * C * T;
  float* ob = p.out + (int64_t)__builtin_amdgcn_readfirstlane(b) * C * T;
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0, C * T * 4, kBufRsrcDword3);
  // descriptors whose base is (row 0, time t0): accumulator-shaped accesses then use non-negative
  // offsets in every tile (t0 < 0 in the first tile of a sequence; columns with t < 0 are never
  // accessed, so the base lying in front of the tensor there is harmless)
  const __amdgpu_buffer_rsrc_t rsx0 = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb) + t0, 0, (C * T - t0) * 4, kBufRsrcDword3);
  const __amdgpu_buffer_rsrc_t rso0 = __builtin_amdgcn_make_buffer_rsrc(ob + t0, 0, (C * T - t0) * 4,
                                                                        kBufRsrcDword3);

  const int co_blk = wm * 32;
  const int wcol = wn * (32 * NB) + (lane & 31);   // this lane's accumulator column for j = 0
  const int G = NCH * p.ktaps * 2;

  // ---- residual stream x at this lane's 4 x 16 accumulator positions -------------------------
  // rows co_blk + (r&3) + 8*(r>>2) + 4*half, columns wcol + 32*j.  Uniform row bases + one 32-bit
  // lane offset: no per-element address arithmetic.
  f32x16 xr[NB];
  const int lane_off = ((co_blk + 4 * half) * T + wcol) * 4;  // bytes from (row 0, time t0)
  float mcol[NB];  // 1 inside [0, T), 0 outside (edge tiles only)
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int t = t0 + wcol + 32 * j;
    mcol[j] = (t >= 0 && t < Tb) ? 1.f : 0.f;
  }
  // ---- 1. stage lrelu(x): wave w takes rows w, w+4, ...; a lane takes aligned 16-byte pieces ----
  {
    const int ppr = Wp >> 2;  // pieces per row
    constexpr int RBT = RPW > 8 ? 8 : RPW;  // rows per batch (bounds the registers in flight)
#pragma unroll 1
    for (int r0 = 0; r0 < RPW; r0 += RBT) {
      float4 st[RBT][QN];
      if (interior) {
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
          const int soff = ((wave + 4 * (r0 + r)) * T + tx0) * 4;  // uniform, 16-byte aligned
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int seg = lane + 64 * q;
            st[r][q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (seg < ppr) {
              const f32x4v u = __builtin_amdgcn_raw_buffer_load_b128(rsx, lane * 16 + q * 1024, soff, 0);
              st[r][q] = make_float4(u.x, u.y, u.z, u.w);
            }
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < RBT; ++r) {
          const float* xrow = xb + (int64_t)(wave + 4 * (r0 + r)) * T;
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int seg = lane + 64 * q;
            const int t = tx0 + 4 * seg;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (seg < ppr) {  // sequence edge: element-wise, zero outside [0, T)
              if (t >= 0 && t < Tb) v.x = xrow[t];
              if (t + 1 >= 0 && t + 1 < Tb) v.y = xrow[t + 1];
              if (t + 2 >= 0 && t + 2 < Tb) v.z = xrow[t + 2];
              if (t + 3 >= 0 && t + 3 < Tb) v.w = xrow[t + 3];
            }
            st[r][q] = v;
          }
        }
      }
#pragma unroll
      for (int r = 0; r < RBT; ++r) {
        float* lrow = smem_c + (size_t)(wave + 4 * (r0 + r)) * Wp;
#pragma unroll
        for (int q = 0; q < QN; ++q) {
          const int seg = lane + 64 * q;
          if (seg < ppr) {
            float4 v = st[r][q];
            v.x = lrelu2(v.x, p.slope);
            v.y = lrelu2(v.y, p.slope);
            v.z = lrelu2(v.z, p.slope);
            v.w = lrelu2(v.w, p.slope);
            *reinterpret_cast<float4*>(lrow + 4 * seg) = v;
          }
        }
      }
    }
  }
  // the residual stream is first needed by c2: its loads land behind c1's MFMA work
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int soff = ((r & 3) + 8 * (r >> 2)) * T * 4;  // uniform
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      xr[j][r] = 0.f;
      if (interior || mcol[j] != 0.f)
        xr[j][r] = buf_load_f32(rsx0, lane_off + 128 * j, soff);
    }
  }

  __syncthreads();

  // ---- the conv loop: acc += W (*) tile, B fragments one k-step ahead ---------------------------
  // bcol: LDS address of (row = half, accumulator column wcol); a tap at offset o reads column + o
  const float* bcol = smem_c + (size_t)half * Wp + wcol + M;
  // A fragments: FOUR register sets.  Group g multiplies from set g % 4 while the load of group g + 2 lands in set
  // (g + 2) % 4, whose last reader was group g - 2 -- no set is read and written in the same group, so nothing has to be
  // copied out of the way (with two sets the compiler rotated them with 8 v_mov per tap, and every VALU instruction costs
  // the f32 matrix pipe ~3 cycles).  The tap loop runs two taps = four groups per iteration so that the set indices are
  // static; the number of (chunk, tap) steps is even because C / 16 is.
  float4 aa[4];
  auto conv_loop = [&](f32x16 (&acc)[NB], const float4* abase, int dil, int h) {
    const float* b0 = bcol - h;  // tap 0
    float bv[2][NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) bv[0][j] = b0[32 * j];
    auto group = [&](const float4& av, float4& apref, const float4* anext, const float* cur, const float* nxt) {
      apref = *anext;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float* src = s < 3 ? cur + (size_t)(2 * (s + 1)) * Wp : nxt;
#pragma unroll
        for (int j = 0; j < NB; ++j) bv[(s + 1) & 1][j] = src[32 * j];
        __builtin_amdgcn_sched_barrier(0);
        const float a = s == 0 ? av.x : s == 1 ? av.y : s == 2 ? av.z : av.w;
#pragma unroll
        for (int j = 0; j < NB; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[s & 1][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    auto rows = [&](int ch, int tp) { return b0 + (size_t)(ch * kConvCK) * Wp + tp * dil; };  // hp = 0 rows of a tap
    auto advance = [&](int& ch, int& tp) {
      if (++tp == p.ktaps) { tp = 0; ++ch; }
    };
    static_assert(NCH % 2 == 0, "an even number of (chunk, tap) steps");
    const int nsteps = NCH * p.ktaps;
    int g = 0, chunk = 0, tap = 0;
    for (int t = 0; t < nsteps; t += 2) {
      const float* r0 = rows(chunk, tap);
      const float* r1 = r0 + (size_t)8 * Wp;  // hp = 1 rows
      advance(chunk, tap);
      const float* s0 = rows(chunk, tap);
      const float* s1 = s0 + (size_t)8 * Wp;
      advance(chunk, tap);
      const float* rn = (t + 2 < nsteps) ? rows(chunk, tap) : s1;
      group(aa[0], aa[2], abase + (int64_t)(g + 2) * 64, r0, r1);
      group(aa[1], aa[3], abase + (int64_t)(g + 3) * 64, r1, s0);
      group(aa[2], aa[0], abase + (int64_t)(g + 4) * 64, s0, s1);
      group(aa[3], aa[1], abase + (int64_t)(g + 5) * 64, s1, rn);
      g += 4;
    }
  };

  const int hk = (p.ktaps - 1) / 2;
  float* wrow = smem_c + (size_t)(co_blk + 4 * half) * Wp + wcol + M;  // this lane's write position

  for (int pr = 0; pr < p.npairs; ++pr) {
    const float4* ab1 = reinterpret_cast<const float4*>(p.wpk[2 * pr]) + ((int64_t)wm * G) * 64 + lane;
    const float4* ab2 = reinterpret_cast<const float4*>(p.wpk[2 * pr + 1]) + ((int64_t)wm * G) * 64 + lane;
    const int d1 = p.dil[pr];
    const bool add_prev = p.accum && pr + 1 == p.npairs;
    // ---- c1 -----------------------------------------------------------------------------------
    aa[0] = ab1[0];
    aa[1] = ab1[64];
    f32x16 acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    conv_loop(acc, ab1, d1, hk * d1);
    aa[0] = ab2[0];  // c2's first two groups land during the write pass
    aa[1] = ab2[64];
    // ---- ft = lrelu(c1 + b1) over the tile (zero outside [0, T): c2 pads ITS input) -------------
    {
      float bia[16];
      const float* bp = p.bias[2 * pr] + co_blk + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[r] = bp[(r & 3) + 8 * (r >> 2)];
      __syncthreads();  // every wave has finished reading the previous stage
      if (interior) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            wrow[(size_t)((r & 3) + 8 * (r >> 2)) * Wp + 32 * j] = lrelu2(acc[j][r] + bia[r], p.slope);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
          for (int j = 0; j < NB; ++j)
            wrow[(size_t)((r & 3) + 8 * (r >> 2)) * Wp + 32 * j] =
                lrelu2(acc[j][r] + bia[r], p.slope) * mcol[j];
      }
      __syncthreads();
    }
    // ---- c2 accumulates into the residual stream: x += c2(ft) ------------------------------------
    // running MRF sum: the conv-by-conv path initialises c2's accumulator with residual + out (in this
    // order) before the MFMAs.  (Requesting these loads before c1 would hide their latency but holds
    // 64 more registers across c1's loop: the kernel then spills.)
    if (add_prev) {
      __builtin_amdgcn_sched_barrier(0);  // keep these loads out of c1's loop (register budget)
#pragma unroll
      for (int jj = 0; jj < NB; jj += 2) {
        float pv[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int soff = ((r & 3) + 8 * (r >> 2)) * T * 4;  // uniform
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (jj + u >= NB) continue;
            const int col = wcol + 32 * (jj + u);
            const bool ok = col >= p.S && col < NTC - p.S && t0 + col < Tb;
            pv[u][r] = 0.f;
            if (ok) pv[u][r] = buf_load_f32(rso0, lane_off + 128 * (jj + u), soff);
          }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (jj + u >= NB) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) xr[jj + u][r] += pv[u][r];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    conv_loop(xr, ab2, 1, hk);
    {
      float bia[16];
      const float* bp = p.bias[2 * pr + 1] + co_blk + 4 * half;
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[r] = bp[(r & 3) + 8 * (r >> 2)];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[j][r] += bia[r];
      if (pr + 1 < p.npairs) {  // next pair's input: lrelu(x) over the tile
        __syncthreads();
        if (interior) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j)
              wrow[(size_t)((r & 3) + 8 * (r >> 2)) * Wp + 32 * j] = lrelu2(xr[j][r], p.slope);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < NB; ++j)
              wrow[(size_t)((r & 3) + 8 * (r >> 2)) * Wp + 32 * j] = lrelu2(xr[j][r], p.slope) * mcol[j];
        }
        __syncthreads();
      }
    }
  }

  // ---- epilogue: out = (x [+ out]) / div on the valid middle columns -------------------------------
  // (three straight-line variants behind uniform branches: a select between the quotient forms inside the unrolled loop
  // would compute the ~10-instruction IEEE sequence for every element whether it is used or not)
  auto store_all = [&](auto fin) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol + 32 * j;
      const int t = t0 + col;
      const bool ok = col >= p.S && col < NTC - p.S && t < Tb;  // t >= n0 >= 0 inside the window
      if (!ok) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        buf_store_f32(fin(xr[j][r]), rso0, lane_off + 128 * j, ((r & 3) + 8 * (r >> 2)) * T * 4);
    }
  };
  const float dv = p.out_div, dinv = 1.f / p.out_div;
  if (dv == 1.f) store_all([](float v) { return v; });
  else if (mrf_div_fast(dv)) store_all([=](float v) { return div_small_const(v, dv, dinv); });
  else store_all([=](float v) { return v / dv; });
}

// per-side half-widths of the chain's convs, in execution order
static void chain_geometry(int ktaps, const int* dil, int npairs, int* S, int* Mmin) {
  const int hk = (ktaps - 1) / 2;
  int s = 0, m = 0;
  for (int pr = 0; pr < npairs; ++pr) {
    const int h1 = hk * dil[pr], h2 = hk;
    if (pr > 0) s += h1;  // conv 0 reads real x in the margins
    s += h2;
    if (h1 > m) m = h1;
    if (h2 > m) m = h2;
  }
  *S = s;
  *Mmin = m;
}

// Accumulator blocks per wave of a launch (see the kernel): 3 at C = 32 where THREE blocks of the narrower tile fit a
// CU's LDS, else 4.  Measured (profiles/r05_chain_narrow_tiles_ab.txt, C = 32): k = 3 whole chain 97.7 -> 98.6 TF/s,
// k = 7 as three pairs 113.0 -> 116.8 (and ahead of the whole chain's 112.3, which the 15 % halo bound now rules out by
// itself), k = 11 pairs 124.2 -> 125.8; headline 61.14 -> 60.9 ms/step same box.  C = 64 loses 1 % (its halo share
// doubles twice as fast) and keeps 4.  (Two blocks per wave, four resident blocks: measured too, no better than 3 --
// profiles/r05_chain_narrow_tiles_ab.txt; the A/B switch is gone.)
static int chain_nb(int C, int Mmin) {
  if (C > 32) return 4;
  const int want = 3, blocks = 3;
  const int ntc = 32 * want * (4 / (C / 32));
  const int wp = (ntc + 2 * Mmin + 4 + 3) & ~3;
  return blocks * (int64_t)C * wp * 4 <= 160 * 1024 ? want : 4;
}

bool resblock_chain32_supported(const PackedConv* c1, const PackedConv* c2, int npairs, int max_lds_bytes,
                                int max_waste_pct) {
  if (npairs < 1 || npairs > RESCHAIN32_MAX_PAIRS) return false;
  const int C = c1[0].Cin, k = c1[0].ktaps;
  if (!(C == 32 || C == 64 || C == 128) || (k & 1) == 0) return false;
  int dil[RESCHAIN32_MAX_PAIRS];
  for (int pr = 0; pr < npairs; ++pr) {
    const PackedConv &a = c1[pr], &b = c2[pr];
    if (a.Cin != C || a.Cout != C || b.Cin != C || b.Cout != C || a.M != C || b.M != C) return false;
    if (a.up || b.up || a.ktaps != k || b.ktaps != k || b.dil != 1) return false;
    if (a.pad != (k - 1) / 2 * a.dil || b.pad != (k - 1) / 2) return false;
    if (!a.bias || !b.bias) return false;
    dil[pr] = a.dil;
  }
  int S, Mmin;
  chain_geometry(k, dil, npairs, &S, &Mmin);
  if (Mmin > RESCHAIN32_MAX_M) return false;
  const int NTC = 32 * chain_nb(C, Mmin) * (4 / (C / 32));
  if (2 * S * 100 > max_waste_pct * NTC || NTC - 2 * S <= 0) return false;
  const int Wp = (NTC + 2 * Mmin + 4 + 3) & ~3;
  return (int64_t)C * Wp * 4 <= max_lds_bytes;
}

int resblock_chain32_nto(int C, int ktaps, const int* dil, int npairs) {
  int S, Mmin;
  chain_geometry(ktaps, dil, npairs, &S, &Mmin);
  return 32 * chain_nb(C, Mmin) * (4 / (C / 32)) - 2 * S;
}

template <int C, int NB>
static int32_t launch_chain(ResChain32Params p, hipStream_t stream) {
  constexpr int NTC = 32 * NB * (4 / (C / 32));
  chain_geometry(p.ktaps, p.dil, p.npairs, &p.S, &p.Mmin);
  p.NTO = NTC - 2 * p.S;
  WETTS_REQUIRE(p.NTO > 0, "chain halo exceeds the tile");
  p.ntiles = cdiv(p.T, p.NTO);
  p.Wp = (NTC + 2 * p.Mmin + 4 + 3) & ~3;
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock grid too large");
  // the staged window starts on a 16-byte boundary only if rows do
  WETTS_REQUIRE(p.T % 4 == 0 && ((uintptr_t)p.x & 15) == 0, "chain kernel needs 16-byte aligned rows");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  const size_t lds = (size_t)C * p.Wp * sizeof(float);
  int dev = 0;
  (void)hipGetDevice(&dev);
  static bool attr_done[64] = {};  // per device: tiles above the default 64 KB dynamic-LDS limit
  if (dev >= 0 && dev < 64 && !attr_done[dev]) {
    WETTS_HIP_CHECK(hipFuncSetAttribute((const void*)resblock_chain32_kernel<C, NB>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((resblock_chain32_kernel<C, NB>), dim3(grid), dim3(256), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

int32_t launch_resblock_chain32(const PackedConv* c1, const PackedConv* c2, int npairs,
                                ResChain32Params p, hipStream_t stream) {
  WETTS_REQUIRE(resblock_chain32_supported(c1, c2, npairs, 160 * 1024, 100),
                "resblock chain shape not supported by the fused f32 kernel");
  p.npairs = npairs;
  p.ktaps = c1[0].ktaps;
  p.bstride = 1;
  if (p.lens) {  // smallest stride >= 8 coprime to B (8 XCDs take the runs)
    auto gcd = [](int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; };
    int st = p.B > 8 ? 8 : 1;
    while (st < p.B && gcd(st, p.B) != 1) ++st;
    p.bstride = st < p.B ? st : 1;
  }
  for (int pr = 0; pr < npairs; ++pr) {
    WETTS_REQUIRE(c1[pr].wpk && c2[pr].wpk, "conv weight not packed");
    p.wpk[2 * pr] = c1[pr].wpk;
    p.bias[2 * pr] = c1[pr].bias;
    p.wpk[2 * pr + 1] = c2[pr].wpk;
    p.bias[2 * pr + 1] = c2[pr].bias;
    p.dil[pr] = c1[pr].dil;
  }
  int S_, Mmin_;
  chain_geometry(p.ktaps, p.dil, npairs, &S_, &Mmin_);
  const int nb = chain_nb(c1[0].Cin, Mmin_);
  switch (c1[0].Cin) {
    case 32: return nb == 3 ? launch_chain<32, 3>(p, stream) : nb == 2 ? launch_chain<32, 2>(p, stream) : launch_chain<32, 4>(p, stream);
    case 64: return launch_chain<64, 4>(p, stream);
    default: return launch_chain<128, 4>(p, stream);
  }
}

}  // namespace wetts
