// Fused ResBlock1 pair for the 16-bit decoder:   out = (x + c2(lrelu(c1(lrelu(x)))) [+ out]) / div
// (reference decoders.py:157-170, one (c1, c2) iteration of ResBlock1.forward) in ONE kernel.
//
// Unfused, a pair moves five tensors through HBM (x in, ft out, ft in, x again as the residual,
// out) and the 16-bit convs at C <= 128 are then memory/latency bound (profiles/r01_conv16_*).
// Here the block keeps the intermediate ft in LDS:
//
//   1. stage lrelu(x) for times [n0-h2-h1, n0-h2-h1 + NTC + 2*h1) as channel-last rows in LDS
//      (h1 = (k-1)/2*d is c1's halo, h2 = (k-1)/2 is c2's; ALL C channels of a row, C <= 128)
//   2. c1 on MFMA for the NTC columns t = n0-h2 .. (A fragments stream from L2, B from LDS)
//   3. barrier; ft = lrelu(round16(c1 + b1)) is written over the x tile (zero outside [0,T):
//      c2 pads ITS input with zeros); barrier
//   4. c2 on MFMA for columns t' = n0 .. n0+NTC, accumulator initialised with the raw residual x
//      (+ the running MRF sum); columns >= NTO = NTC - 2*h2 see a truncated ft window and are
//      discarded, so consecutive blocks advance by NTO
//   5. + b2, / div, round, 16-byte channel-last stores
//
// HBM traffic per pair: x once (+ halo, mostly L2 hits), the residual re-read (L2 hit) and out
// once -- 2 tensor passes instead of 5.  Arithmetic (operation order, rounding points) is exactly
// that of two conv_bf16_kernel launches, so the fused and unfused paths agree bit for bit
// (tests/test_gpu_parity.py::test_fused_resblock_pair_bit_identical).
#include <stdlib.h>

#include <vector>

#include <type_traits>

#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

// NR = depth of the A-fragment register ring in (chunk, tap) groups: the fragments of group g+NR-1
// are requested while group g computes, which has to cover a loaded L2 round trip (~1-2 us);
// OCC = waves per SIMD the register allocation is held to.
// RB2 = true: a whole ResBlock2 (decoders.py:205-214: both convs residual, c2 at dilation p.dil2) --
// c2 shares c1's column -> time mapping, so the rounded t1 a lane needs as c2's residual is what its
// accumulators just produced; lrelu(t1) is written h2 rows down the tile, valid outputs are the
// middle columns [h2, NTC - h2)  (see resblock32.hip).
// MB = 32-row m-blocks per wave.  MB = 2 (C >= 64; round 3): one B fragment -- one ds_read_b128 -- feeds TWO MFMAs.
// The round-1 ablation of the 16-bit loop (profiles/r01_conv16_ablation.txt) put the loop with everything but
// the MFMAs and the LDS B reads removed at 53 % of the bf16 peak: with one LDS read per 32-cycle MFMA the LDS
// issue, not the matrix pipe, sets the pace.  A wave now owns 64 rows x 128 columns (8 accumulators), a block is
// two waves (C = 128: 2 x 1, C = 64: 1 x 2) on the same 128 / 256-column tile as before, so the LDS tile, the
// column -> time mapping, the K order of every accumulator and therefore the results are unchanged.
// Waves per block (MB = 1).  ResBlock1 pairs: 4 (three blocks per CU, three waves per SIMD).  ResBlock2 chains: 8, one block
// per CU (round 6; configs[2] 7.50 -> 7.38 ms/step same box) -- its c2 halo ((k - 1) / 2 * dilation, up to 36 columns a
// side) is then amortised over twice the columns.  Eight waves for the ResBlock1 pairs measured WORSE (configs[4] 13.75 ->
// 15.09 ms/step, v1 bf16 11.13 -> 11.94): their halo is 1-5 columns, and two waves per SIMD instead of three is all it buys.
constexpr int pair16_waves(bool rb2) { return rb2 ? 8 : 4; }
template <int C, bool F16, int NR, int OCC, bool RB2, int MB>
__global__ __launch_bounds__(64 * (C / (32 * MB)) * (MB == 2 ? (C == 128 ? 1 : 2) : pair16_waves(RB2) / (C / 32)))
__attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void resblock_pair16_kernel(const ResPairParams p) {
  constexpr int WM = C / (32 * MB), WN = MB == 2 ? (C == 128 ? 1 : 2) : pair16_waves(RB2) / (C / 32), NB = 4;
  constexpr int NTH = 64 * WM * WN;            // threads per block
  constexpr int NTC = 32 * NB * WN;            // columns computed per conv
  constexpr int CKB = C >= 64 ? 64 : 32;       // K chunk of the packed weights (pack_bf16_kernel)
  constexpr int NCH = C / CKB, KS = CKB / 16;
  constexpr int SEG = C / 8;                   // 16-byte pieces per row
  constexpr int RS = C * 2 + 16;               // padded LDS row stride (bytes)
  // widest halo the staging loop is sized for; the C = 32 ResBlock2 chain also takes v3's k = 7 block
  // (dilations 3 / 12: 72 columns, 14 % of the 512-column tile)
  constexpr int MAXSPAN = (RB2 && C == 32) ? RESPAIR2_MAX_SPAN32 : RESPAIR_MAX_SPAN;
  constexpr int RPP = NTH / SEG;               // rows per staging pass
  constexpr int MAXU = (NTC + MAXSPAN + RPP - 1) / RPP;
  static_assert(NTH % SEG == 0, "piece index must not depend on the pass");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_r[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  const int dil2 = RB2 ? p.dil2 : 1;
  const int h2 = (p.ktaps - 1) / 2 * dil2, h1 = (p.ktaps - 1) / 2 * p.dil;
  const int NTO = NTC - 2 * h2;
  // XCD-aware tile order: the 8 XCDs take blocks round-robin, so give each XCD a contiguous run
  // of time tiles (neighbours share their halo rows through that XCD's L2)
  int bid = blockIdx.x;
  {
    const int per = (p.nblocks + 7) >> 3;
    bid = (bid & 7) * per + (bid >> 3);
    if (bid >= p.nblocks) return;
  }
  const int ntile = bid % p.ntiles;
  const int b = bid / p.ntiles;
  const int n0 = ntile * NTO;
  const int W1 = NTC + 2 * (h1 > h2 ? h1 : h2);  // also backs c2's reads of its discarded columns
  const int tx0 = n0 - h2 - h1;  // time of LDS row 0 of the x tile

  const unsigned short* xb = p.x + (int64_t)b * p.T * C;
  // the utterance's plane as a raw buffer: rows outside it load zeros (conv16_dev.h: plane_rsrc)
  const __amdgpu_buffer_rsrc_t rsx = plane_rsrc(xb, p.T * C * 2);
  const f32x2v slope2 = {p.slope, p.slope};
  // ---- A streams -----------------------------------------------------------------------------
  const int G = NCH * p.ktaps;
  const uint4* abase1 = reinterpret_cast<const uint4*>(p.wpk1) + ((int64_t)(wm * MB) * G * KS) * 64 + lane;
  const uint4* abase2 = reinterpret_cast<const uint4*>(p.wpk2) + ((int64_t)(wm * MB) * G * KS) * 64 + lane;
  const int64_t mstride = (int64_t)G * KS * 64;  // uint4 elements between consecutive m-blocks
  uint4 aa[NR][MB][KS];
  auto a_prologue = [&](const uint4* abase) {  // groups 0 .. NR-2 in flight
#pragma unroll
    for (int r = 0; r < NR - 1; ++r)
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          aa[r][i][s] = abase[i * mstride + ((int64_t)(r < G ? r : 0) * KS + s) * 64];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int s = 0; s < KS; ++s) aa[NR - 1][i][s] = aa[0][i][s];
  };
  a_prologue(abase1);

  // ResBlock2: raw x at c1's columns (time n0 - h2 + col) initialises c1's accumulators; requested
  // first so it is the oldest load in flight
  const int co_blk = wm * 32 * MB;
  const int wcol = wn * (32 * NB) + (lane & 31);  // this lane's column of n-block 0
  uint4 rres[MB][NB][2];
  if (RB2) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = n0 - h2 + wcol + 32 * j;
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int i = 0; i < 2; ++i) rres[mi][j][i] = plane_load16(rsx, (t * C + co_blk + 32 * mi + 16 * i + 8 * half) * 2);
    }
  }

  // ---- 1. stage lrelu(x) ---------------------------------------------------------------------
  // (the tile is allocated in whole passes of RPP rows: no row guard on the stores; rows outside the utterance arrive as
  // zeros from the buffer loads, no bounds test per piece)
  {
    const int useg = tid % SEG, urow = tid / SEG;
    const int voff = ((tx0 + urow) * C + useg * 8) * 2;
    uint4 st[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (i * RPP < W1) st[i] = plane_load16(rsx, voff + i * (RPP * C * 2));
    unsigned char* dst = smem_r + (size_t)urow * RS + useg * 16;
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (i * RPP < W1) {
        uint4 v = st[i];
        v.x = lrelu_pk2<F16>(v.x, slope2); v.y = lrelu_pk2<F16>(v.y, slope2);
        v.z = lrelu_pk2<F16>(v.z, slope2); v.w = lrelu_pk2<F16>(v.w, slope2);
        *reinterpret_cast<uint4*>(dst + (size_t)i * (RPP * RS)) = v;
      }
  }
  __syncthreads();

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;

  const unsigned char* bcol = smem_r + (size_t)wcol * RS + half * 16;
  if (RB2) {
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w4[4] = {rres[mi][j][i].x, rres[mi][j][i].y, rres[mi][j][i].z, rres[mi][j][i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[mi][j][8 * i + 2 * e] = lo16<F16>(w4[e]);
            acc[mi][j][8 * i + 2 * e + 1] = hi16<F16>(w4[e]);
          }
        }
  }

  // one conv over the LDS tile: groups g = chunk*ktaps + tap, A fragments from the register ring.
  // The prefetch of group g+NR-1 is issued UNCONDITIONALLY (index clamped to the last group):
  // a conditional load makes the compiler's s_waitcnt insertion assume it may not be pending, and
  // the vmcnt it then emits also waits for the load just issued -- i.e. a full L2 round trip per
  // group.  Straight-line issue gives exact counts (vmcnt((NR-1)*KS*MB) ... ).
  // B fragments software-pipelined one k-step ahead (pinned by sched_barriers): the four ds_read_b128 of
  // step s+1 are issued in front of the MFMAs of step s, across group boundaries too (`nxt` = the next
  // group's tile position), so a wave hides its own LDS latency instead of relying on its SIMD neighbour
  uint4 bq[2][NB];
  auto b_load = [&](uint4 (&dst)[NB], const unsigned char* bb, int s) {
#pragma unroll
    for (int j = 0; j < NB; ++j) dst[j] = *reinterpret_cast<const uint4*>(bb + (size_t)(32 * j) * RS + s * 32);
  };
  auto mma_group2 = [&](const uint4 (&av)[MB][KS], const unsigned char* cur, const unsigned char* nxt) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) b_load(bq[(s + 1) & 1], cur, s + 1);
      else b_load(bq[(s + 1) & 1], nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) acc[mi][j] = mfma16<F16>(av[mi][s], bq[s & 1][j], acc[mi][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto conv_loop = [&](const uint4* abase, int dil) {
    int chunk = 0, tap = 0, g = 0;
    auto bpos = [&](int tp, int ch) { return bcol + (size_t)(tp * dil) * RS + ch * (CKB * 2); };
    auto advance = [&](int& tp, int& ch) { if (++tp == p.ktaps) { tp = 0; ++ch; } };
    b_load(bq[0], bpos(0, 0), 0);
    for (; g + NR <= G; g += NR) {
#pragma unroll
      for (int par = 0; par < NR; ++par) {
        int gn = g + par + NR - 1;
        gn = gn < G ? gn : G - 1;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
          for (int s = 0; s < KS; ++s)
            aa[(par + NR - 1) % NR][mi][s] = abase[mi * mstride + ((int64_t)gn * KS + s) * 64];
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top of its group
        {
          int ntap = tap, nchunk = chunk;
          advance(ntap, nchunk);
          if (nchunk >= NCH) { ntap = tap; nchunk = chunk; }  // last group: a harmless re-read of its own tile
          mma_group2(aa[par], bpos(tap, chunk), bpos(ntap, nchunk));
        }
        advance(tap, chunk);
      }
    }
#pragma unroll
    for (int par = 0; par < NR - 1; ++par) {  // tail: fewer than NR groups left, all in the ring
      if (g + par < G) {
        {
          int ntap = tap, nchunk = chunk;
          advance(ntap, nchunk);
          if (nchunk >= NCH) { ntap = tap; nchunk = chunk; }
          mma_group2(aa[par], bpos(tap, chunk), bpos(ntap, nchunk));
        }
        advance(tap, chunk);
      }
    }
  };

  // ---- 2. c1 ---------------------------------------------------------------------------------
  conv_loop(abase1, p.dil);

  // c2's first A groups are requested now; they land during step 3.  (MB = 1 also requests the raw residual
  // here; with two m-blocks per wave its 64 registers on top of the 128 accumulators spill, so MB = 2 loads it
  // in step 4, when c1's accumulators are dead -- an L2 hit: the same rows were staged a moment ago.)
  if (!RB2) a_prologue(abase2);  // (RB2: after step 3 -- t1 keeps all accumulators live through it, the ring would spill)
  if (MB == 1) {
    if (!RB2) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {  // (columns from NTO on are never stored: whatever row they load is harmless)
        const int t = n0 + wcol + 32 * j;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
          for (int i = 0; i < 2; ++i) rres[mi][j][i] = plane_load16(rsx, (t * C + co_blk + 32 * mi + 16 * i + 8 * half) * 2);
      }
    }
  }

  // ---- 3. ft = lrelu(round16(c1 + b1)) over the x tile ----------------------------------------
  {
    f32x2v bia[MB][8];
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int cc = co_blk + 32 * mi + 16 * (r >> 2) + 8 * half + 2 * (r & 3);
        bia[mi][r] = f32x2v{p.bias1[cc], p.bias1[cc + 1]};
      }
    __syncthreads();  // every wave has finished reading lrelu(x)
    // columns whose time lies outside the utterance (zero padding of c2's input) exist only in its first / last tiles:
    // a block-uniform choice, not a select per element
    const bool edge = n0 - h2 < 0 || n0 - h2 + NTC > p.T;
    auto mid = [&](auto edge_c) {
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = wcol + 32 * j;
        const int t = n0 - h2 + col;
        const bool inside = t >= 0 && t < p.T;
#pragma unroll
        for (int mi = 0; mi < MB; ++mi)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (RB2) {
                // the rounded t1 is c2's residual: it stays in the accumulators.  (Element by element here: with the 64
                // accumulators live through this phase, the aligned register pairs of the packed form spill at C >= 64.)
                const unsigned r16 = pk2<F16>(acc[mi][j][8 * i + 2 * e] + bia[mi][4 * i + e].x,
                                              acc[mi][j][8 * i + 2 * e + 1] + bia[mi][4 * i + e].y);
                acc[mi][j][8 * i + 2 * e] = lo16<F16>(r16);
                acc[mi][j][8 * i + 2 * e + 1] = hi16<F16>(r16);
                w[e] = pk2<F16>(lrelu_max(acc[mi][j][8 * i + 2 * e], p.slope), lrelu_max(acc[mi][j][8 * i + 2 * e + 1], p.slope));
              } else {
                const f32x2v a2 = f32x2v{acc[mi][j][8 * i + 2 * e], acc[mi][j][8 * i + 2 * e + 1]} + bia[mi][4 * i + e];
                const f32x2v l2 = lrelu2v(unpack2<F16>(pk2<F16>(a2.x, a2.y)), slope2);
                w[e] = pk2<F16>(l2.x, l2.y);
              }
              if (decltype(edge_c)::value) w[e] = inside ? w[e] : 0u;
            }
            *reinterpret_cast<uint4*>(smem_r + (size_t)(col + (RB2 ? h2 : 0)) * RS +
                                      (co_blk + 32 * mi + 16 * i + 8 * half) * 2) = make_uint4(w[0], w[1], w[2], w[3]);
          }
      }
    };
    if (edge) mid(std::true_type{});
    else mid(std::false_type{});
    if (RB2) a_prologue(abase2);
    __syncthreads();
  }

  // ---- 4. c2, accumulator = residual (+ running sum) -------------------------------------------
  unsigned short* ob = p.out + (int64_t)b * p.T * C;
#pragma unroll
  for (int mi = 0; mi < MB; ++mi) {
    if (MB == 2 && !RB2) {  // one m-block's residual at a time: 32 registers in flight, not 64
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int t = n0 + wcol + 32 * j;
#pragma unroll
        for (int i = 0; i < 2; ++i) rres[mi][j][i] = plane_load16(rsx, (t * C + co_blk + 32 * mi + 16 * i + 8 * half) * 2);
      }
    }
    // the running MRF sum of this m-block's outputs, requested in batches (JB column blocks at a time: all four, or two
    // where t1 occupies the accumulators as well -- RB2 -- and a batch of four would spill) from addresses clamped into
    // the utterance (columns outside the stored range get some valid row's values; they are never stored), so the
    // loads go out back to back instead of one load / wait / add round trip per piece (tools/isa_scan.py)
    constexpr int JB = RB2 ? 2 : NB;
#pragma unroll
    for (int j0 = 0; j0 < NB; j0 += JB) {
      uint4 osum[JB][2];
      if (p.accum) {
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) {
          int t = (RB2 ? n0 - h2 : n0) + wcol + 32 * (j0 + jj);
          t = t < 0 ? 0 : (t >= p.T ? p.T - 1 : t);
#pragma unroll
          for (int i = 0; i < 2; ++i)
            osum[jj][i] = *reinterpret_cast<const uint4*>(ob + (int64_t)t * C + co_blk + 32 * mi + 16 * i + 8 * half);
        }
      }
#pragma unroll
      for (int jj = 0; jj < JB; ++jj) {
        const int j = j0 + jj;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned w4[4] = {rres[mi][j][i].x, rres[mi][j][i].y, rres[mi][j][i].z, rres[mi][j][i].w};
          f32x2v v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = RB2 ? f32x2v{acc[mi][j][8 * i + 2 * e], acc[mi][j][8 * i + 2 * e + 1]} : unpack2<F16>(w4[e]);
          if (p.accum) {
            const unsigned o4[4] = {osum[jj][i].x, osum[jj][i].y, osum[jj][i].z, osum[jj][i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += unpack2<F16>(o4[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            acc[mi][j][8 * i + 2 * e] = v[e].x;
            acc[mi][j][8 * i + 2 * e + 1] = v[e].y;
          }
        }
      }
    }
    if (MB == 2) __builtin_amdgcn_sched_barrier(0);
  }
  conv_loop(abase2, dil2);

  // ---- 5. epilogue -----------------------------------------------------------------------------
  f32x2v bia[MB][8];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int cc = co_blk + 32 * mi + 16 * (r >> 2) + 8 * half + 2 * (r & 3);
      bia[mi][r] = f32x2v{p.bias2[cc], p.bias2[cc + 1]};
    }
  auto store_all = [&](auto fin) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol + 32 * j;
      const int t = RB2 ? n0 - h2 + col : n0 + col;
      if (RB2 ? (col < h2 || col >= NTC - h2 || t < 0) : col >= NTO) continue;
      if (t >= p.T) continue;
#pragma unroll
      for (int mi = 0; mi < MB; ++mi)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          unsigned w[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2v v = fin(f32x2v{acc[mi][j][8 * i + 2 * e], acc[mi][j][8 * i + 2 * e + 1]} + bia[mi][4 * i + e]);
            w[e] = pk2<F16>(v.x, v.y);
          }
          *reinterpret_cast<uint4*>(ob + (int64_t)t * C + co_blk + 32 * mi + 16 * i + 8 * half) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
  };
  const float dv = p.out_div, dinv = 1.f / p.out_div;  // the MRF mean (common.h: mrf_div), uniform choice
  if (dv == 1.f) {
    store_all([](f32x2v v) { return v; });
  } else if (mrf_div_fast(dv)) {
    const f32x2v c2 = {dinv, dinv}, nd2 = {-dv, -dv};
    store_all([=](f32x2v v) { return div_small_const2(v, nd2, c2); });
  } else {
    store_all([=](f32x2v v) { return f32x2v{v.x / dv, v.y / dv}; });
  }
}

template <int C, int NR, int OCC, bool RB2, int MB>
static int32_t launch_pair(const ResPairParams& p0, bool f16, hipStream_t stream) {
  constexpr int WM = C / (32 * MB), WN = MB == 2 ? (C == 128 ? 1 : 2) : pair16_waves(RB2) / (C / 32);
  constexpr int NTC = 128 * WN, RS = C * 2 + 16, NTH = 64 * WM * WN;
  ResPairParams p = p0;
  const int h2 = (p.ktaps - 1) / 2 * (RB2 ? p.dil2 : 1), h1 = (p.ktaps - 1) / 2 * p.dil;
  const int NTO = NTC - 2 * h2;
  WETTS_REQUIRE(NTO > 0, "second conv's halo exceeds the tile");
  p.ntiles = cdiv(p.T, NTO);
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock grid too large");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  WETTS_REQUIRE((int64_t)p.T * C * 2 < (int64_t)INT32_MAX, "utterance plane too large for the pair kernel's 32-bit offsets");
  constexpr int RPP = NTH / (C / 8);  // the tile in whole staging passes (the kernel's stores carry no row guard)
  const size_t lds = (size_t)((NTC + 2 * (h1 > h2 ? h1 : h2) + RPP - 1) / RPP * RPP) * RS;
  WETTS_REQUIRE(lds <= 160 * 1024, "pair tile exceeds the LDS");
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per device and instantiation
    static signed char st16[2][64] = {};
    if (f16) WETTS_REQUIRE(lds_opt_in((const void*)resblock_pair16_kernel<C, true, NR, OCC, RB2, MB>, st16[1]), "LDS opt-in refused");
    else WETTS_REQUIRE(lds_opt_in((const void*)resblock_pair16_kernel<C, false, NR, OCC, RB2, MB>, st16[0]), "LDS opt-in refused");
  }
  if (f16)
    hipLaunchKernelGGL((resblock_pair16_kernel<C, true, NR, OCC, RB2, MB>), dim3(grid), dim3(NTH), lds, stream, p);
  else
    hipLaunchKernelGGL((resblock_pair16_kernel<C, false, NR, OCC, RB2, MB>), dim3(grid), dim3(NTH), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// output columns a block of the pair kernel computes per conv (the decoder's tile arithmetic, model.hip)
int resblock_pair16_ntc(int C, bool rb2) {
  return 128 * (pair16_waves(rb2) / (C / 32));
}

bool resblock_pair16_supported(const PackedConvB& c1, const PackedConvB& c2) {
  const int C = c1.Cin;
  if (!(C == 32 || C == 64 || C == 128)) return false;
  if (c1.Cout != C || c2.Cin != C || c2.Cout != C || c1.up || c2.up) return false;
  if (c1.ktaps != c2.ktaps || (c1.ktaps & 1) == 0 || c2.dil != 1 || c1.f16 != c2.f16) return false;
  if (c1.pad != (c1.ktaps - 1) / 2 * c1.dil || c2.pad != (c2.ktaps - 1) / 2) return false;
  return (c1.ktaps - 1) * c1.dil <= RESPAIR_MAX_SPAN;
}

int32_t launch_resblock_pair16(const PackedConvB& c1, const PackedConvB& c2, ResPairParams p,
                               hipStream_t stream) {
  WETTS_REQUIRE(resblock_pair16_supported(c1, c2), "resblock pair shape not supported by the fused kernel");
  WETTS_REQUIRE(c1.wpk && c2.wpk, "16-bit conv weight not packed");
  p.wpk1 = c1.wpk; p.bias1 = c1.bias;
  p.wpk2 = c2.wpk; p.bias2 = c2.bias;
  p.ktaps = c1.ktaps;
  p.dil = c1.dil;
  const bool h = c1.f16 != 0;
  // ring depth 2 at 3 waves/SIMD measured best (profiles/r01_conv16_fused_pair.txt: deeper rings
  // cost occupancy or issue slots and lose 5-15 %)
  p.dil2 = 1;
  switch (c1.Cin) {
    case 32: return launch_pair<32, 2, 3, false, 1>(p, h, stream);
    case 64: return launch_pair<64, 2, 3, false, 1>(p, h, stream);
    default: return launch_pair<128, 2, 3, false, 1>(p, h, stream);
  }
}

bool resblock2_chain16_supported(const PackedConvB& c1, const PackedConvB& c2, int max_waste_pct) {
  const int C = c1.Cin;
  if (!(C == 32 || C == 64 || C == 128)) return false;
  if (c1.Cout != C || c2.Cin != C || c2.Cout != C || c1.up || c2.up) return false;
  if (c1.ktaps != c2.ktaps || (c1.ktaps & 1) == 0 || c1.f16 != c2.f16) return false;
  if (c1.pad != (c1.ktaps - 1) / 2 * c1.dil || c2.pad != (c2.ktaps - 1) / 2 * c2.dil) return false;
  if ((c1.ktaps - 1) * (c1.dil > c2.dil ? c1.dil : c2.dil) > (C == 32 ? RESPAIR2_MAX_SPAN32 : RESPAIR_MAX_SPAN))
    return false;
  const int NTC = resblock_pair16_ntc(C, true);
  return (c2.ktaps - 1) * c2.dil * 100 <= max_waste_pct * NTC;
}

int32_t launch_resblock2_chain16(const PackedConvB& c1, const PackedConvB& c2, ResPairParams p,
                                 hipStream_t stream) {
  WETTS_REQUIRE(resblock2_chain16_supported(c1, c2, 100), "ResBlock2 shape not supported by the fused kernel");
  WETTS_REQUIRE(c1.wpk && c2.wpk, "16-bit conv weight not packed");
  p.wpk1 = c1.wpk; p.bias1 = c1.bias;
  p.wpk2 = c2.wpk; p.bias2 = c2.bias;
  p.ktaps = c1.ktaps;
  p.dil = c1.dil;
  p.dil2 = c2.dil;
  const bool h = c1.f16 != 0;
  switch (c1.Cin) {
    case 32: return launch_pair<32, 2, 2, true, 1>(p, h, stream);
    case 64: return launch_pair<64, 2, 2, true, 1>(p, h, stream);
    default: return launch_pair<128, 2, 2, true, 1>(p, h, stream);
  }
}

}  // namespace wetts
