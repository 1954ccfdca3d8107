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
#include <type_traits>

#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

// Round 6: what the SQ counters of this kernel said (profiles/r06_sq_counters_mrf16.txt: matrix pipe 26 % busy at C = 32,
// 15 VALU + 6.6 SALU instructions per MFMA, 190 uniform branches in the epilogues) and what was done about it:
//  * SHARED staging (C = 32): lrelu(x) is the same tile for every chain of the stage -- only the halo differs -- so it is
//    staged ONCE with the widest c1 halo into its own LDS tile, and lrelu(t) of each chain goes to a second tile.  The
//    per-chain re-staging (global load, unpack, leaky-relu, pack, ds_write: ~28 VALU per 16-byte piece, three times over)
//    was a third of the kernel's vector work.  The block is eight waves wide (768 columns at C = 32, 384 at C = 64: see
//    stage16_nb below), one per CU.
//  * the raw x of a lane's own outputs (c1's accumulator init) is loaded once and kept in registers across the chains;
//  * x is addressed through a buffer descriptor of the utterance's plane: rows before / behind the utterance read as
//    zeros (lrelu(0) = 0: the convs' zero padding) without a bounds test, exec mask or branch per piece;
//  * the `inside` select of lrelu(t) and the choice of the quotient are uniform per block and hoisted out of the
//    per-element loops; bias adds, slope products and the quotient run on pairs (v_pk_add / mul / fma_f32).
// Operation order and rounding points per output element are unchanged: bit-identical to the chain-by-chain launches.
template <int C, bool F16, int NR, int OCC, int NB, bool SHARED, int NW = 4>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(OCC, OCC)))
void rb2_stage16_kernel(const ResStage2Params p) {
  constexpr int WM = C / 32, WN = NW / WM;
  constexpr int NTH = 64 * NW;
  constexpr int NTC = 32 * NB * WN;
  constexpr int CKB = C >= 64 ? 64 : 32;
  constexpr int NCH = C / CKB, KS = CKB / 16;
  constexpr int SEG = C / 8;      // 16-byte pieces per row
  constexpr int RPP = NTH / SEG;  // rows per staging pass
  constexpr int RS = C * 2 + 16;
  constexpr int MAXU = (NTC + RESPAIR2_MAX_SPAN32 + RPP - 1) / RPP;
  static_assert(NTH % SEG == 0, "piece index must not depend on the pass");

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
  // the utterance's [T][C] plane as a raw buffer: offsets outside it (rows before 0 / from T on) load zeros
  const __amdgpu_buffer_rsrc_t rsx = plane_rsrc(xb, p.T * C * 2);
  const int co_blk = wm * 32;
  const int wcol = wn * (32 * NB) + (lane & 31);
  unsigned char* const xt = smem_r;                                       // lrelu(x)
  unsigned char* const tt = SHARED ? smem_r + (size_t)p.xrows * RS : xt;  // lrelu(t) of the current chain
  const f32x2v slope2 = {p.slope, p.slope};
  // columns whose time lies outside the utterance exist only in its first / last tiles (block-uniform)
  const bool edge = n0 - o < 0 || n0 - o + NTC > p.T;

  auto load16 = [&](int byte_off) { return plane_load16(rsx, byte_off); };

  // raw x at this lane's columns: c1's accumulator init in every chain (requested first: the oldest loads in flight).
  // SHARED (C = 32) keeps it in registers across the chains; the C = 64 tile has no room (it spills) and reloads it per
  // chain, an L2 hit
  uint4 rres[NB][2];
  auto load_res = [&]() {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = n0 - o + wcol + 32 * j;
#pragma unroll
      for (int i = 0; i < 2; ++i) rres[j][i] = load16((t * C + co_blk + 16 * i + 8 * half) * 2);
    }
  };
  if (SHARED) load_res();

  // ---- stage lrelu(x): rows r <-> time tx0 + r --------------------------------------------------------------------
  auto stage_x = [&](int tx0, int rows) {
    const int useg = tid % SEG, urow = tid / SEG;
    const int voff = ((tx0 + urow) * C + useg * 8) * 2;
    uint4 st[MAXU];
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (i * RPP < rows) st[i] = load16(voff + i * (RPP * C * 2));
    unsigned char* dst = xt + (size_t)urow * RS + useg * 16;
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (i * RPP < rows) {  // (the tile is allocated in whole passes: no guard on the last pass's rows)
        uint4 v = st[i];
        v.x = lrelu_pk2<F16>(v.x, slope2); v.y = lrelu_pk2<F16>(v.y, slope2);
        v.z = lrelu_pk2<F16>(v.z, slope2); v.w = lrelu_pk2<F16>(v.w, slope2);
        *reinterpret_cast<uint4*>(dst + (size_t)i * (RPP * RS)) = v;
      }
  };
  if (SHARED) {
    stage_x(n0 - o - p.h1max, NTC + 2 * p.h1max);
    // rows of the lrelu(t) tile no chain writes (the halo on both sides of its NTC columns) feed discarded columns
    // only; zeroed once so that nothing uninitialised enters an MFMA
    for (int r = tid; r < 2 * o * (RS / 16); r += NTH) {
      const int row = r / (RS / 16), pc = r % (RS / 16);
      *reinterpret_cast<uint4*>(tt + (size_t)(row < o ? row : NTC + row) * RS + pc * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
  }

  uint4 sum[NB][2];  // the running MRF sum of this lane's outputs, packed 16 bit (rounded after every chain)
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) sum[j][i] = make_uint4(0u, 0u, 0u, 0u);

  for (int ch = 0; ch < p.nchain; ++ch) {
    const int ktaps = p.ktaps[ch], dil1 = p.dil1[ch], dil2 = p.dil2[ch];
    const int h1 = (ktaps - 1) / 2 * dil1, h2 = (ktaps - 1) / 2 * dil2;
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
    if (!SHARED) load_res();

    // ---- 1. lrelu(x) in LDS (SHARED: staged once above; else per chain with this chain's halo, L2 hits) ----------
    if (!SHARED) {
      if (ch > 0) __syncthreads();  // the previous chain's c2 has finished reading the tile
      stage_x(n0 - o - h1, NTC + 2 * (h1 > h2 ? h1 : h2));
    }
    if (!SHARED || ch == 0) __syncthreads();

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

    // one conv over an LDS tile (the loop of resblock16.hip at MB = 1: unconditional A prefetch NR - 1 groups
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
    auto conv_loop = [&](const uint4* abase, int dil, const unsigned char* bcol) {
      int chunk = 0, tap = 0, g = 0;
      auto bpos = [&](int tp, int cq) { return bcol + (size_t)(tp * dil) * RS + cq * (CKB * 2); };
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

    // ---- 2. c1: column c <-> time n0 - o + c reads the x tile's rows c + tap * dil1 (+ the halo this chain leaves unused)
    conv_loop(abase1, dil1, xt + (size_t)(wcol + (SHARED ? p.h1max - h1 : 0)) * RS + half * 16);
    a_prologue(abase2);

    // ---- 3. t = round16(c1 + b1) stays in the accumulators; lrelu(t) goes to the t tile, toff rows down --------------
    const int toff = SHARED ? o : h2;
    {
      f32x2v bia[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int cc = co_blk + 16 * (r >> 2) + 8 * half + 2 * (r & 3);
        bia[r] = f32x2v{p.bias1[ch][cc], p.bias1[ch][cc + 1]};
      }
      // the tile's readers are done: this chain's c1 (single tile) / the previous chain's c2 (shared staging)
      if (!SHARED || ch > 0) __syncthreads();
      auto mid = [&](auto edge_c) {
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
              const f32x2v a2 = f32x2v{acc[j][8 * i + 2 * e], acc[j][8 * i + 2 * e + 1]} + bia[4 * i + e];
              const f32x2v t2 = unpack2<F16>(pk2<F16>(a2.x, a2.y));
              acc[j][8 * i + 2 * e] = t2.x;
              acc[j][8 * i + 2 * e + 1] = t2.y;
              // lrelu_pk(r16) on the values just unpacked (same operations, two unpacks less)
              const f32x2v l2 = lrelu2v(t2, slope2);
              w[e] = pk2<F16>(l2.x, l2.y);
              if (decltype(edge_c)::value) w[e] = inside ? w[e] : 0u;
            }
            *reinterpret_cast<uint4*>(tt + (size_t)(col + toff) * RS + (co_blk + 16 * i + 8 * half) * 2) =
                make_uint4(w[0], w[1], w[2], w[3]);
          }
        }
      };
      if (edge) mid(std::true_type{});
      else mid(std::false_type{});
      __syncthreads();
    }

    // ---- 4. c2: accumulator = t (+ the running sum), rows c + tap * dil2 of the t tile ------------------------------
    if (ch > 0) {
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned o4[4] = {sum[j][i].x, sum[j][i].y, sum[j][i].z, sum[j][i].w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const f32x2v s2 = f32x2v{acc[j][8 * i + 2 * e], acc[j][8 * i + 2 * e + 1]} + unpack2<F16>(o4[e]);
            acc[j][8 * i + 2 * e] = s2.x;
            acc[j][8 * i + 2 * e + 1] = s2.y;
          }
        }
    }
    conv_loop(abase2, dil2, tt + (size_t)(wcol + toff - h2) * RS + half * 16);

    // ---- 5. + b2 (/ n on the last chain), round: the new running sum ---------------------------------------
    {
      f32x2v bia[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int cc = co_blk + 16 * (r >> 2) + 8 * half + 2 * (r & 3);
        bia[r] = f32x2v{p.bias2[ch][cc], p.bias2[ch][cc + 1]};
      }
      auto fin_all = [&](auto fin) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            unsigned w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const f32x2v v = fin(f32x2v{acc[j][8 * i + 2 * e], acc[j][8 * i + 2 * e + 1]} + bia[4 * i + e]);
              w[e] = pk2<F16>(v.x, v.y);
            }
            sum[j][i] = make_uint4(w[0], w[1], w[2], w[3]);
          }
      };
      const float dv = p.out_div;
      if (ch != p.nchain - 1 || dv == 1.f) {
        fin_all([](f32x2v v) { return v; });
      } else if (mrf_div_fast(dv)) {  // (num_kernels of every recipe: 3) common.h: div_small_const on a pair
        const float dinv = 1.f / dv;
        const f32x2v c2 = {dinv, dinv}, nd2 = {-dv, -dv};
        fin_all([=](f32x2v v) { return div_small_const2(v, nd2, c2); });
      } else {
        fin_all([=](f32x2v v) { return f32x2v{v.x / dv, v.y / dv}; });
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

// The tile shapes (round 6, measured on configs[2], same box): EIGHT waves per block, three accumulator blocks per wave,
// lrelu(x) staged once for all chains into its own LDS tile -- C = 32: 768 columns (139 KB of LDS), C = 64: 384 columns
// (130 KB): one block of eight waves per CU instead of two of four.  The halo the common origin costs drops from 72 of 384
// (512 before the shared staging) columns to 72 of 768 at C = 32 and from 72 of 256 to 72 of 384 at C = 64, and the step went
// 8.61 -> 8.19 (C = 32 on eight waves) -> 7.69 (C = 64 too, one restaged tile of 512 columns) -> 7.55 ms (C = 64 with the shared
// staging); the kernels' own durations moved less (C = 32 1.55 -> 1.56 ms, C = 64 1.37 -> 1.26) than the step -- a CU's one
// big block leaves the neighbouring call's encoder stages (bench.py pipelines calls) more room than two did.
static constexpr int stage16_nb(int C) { return 3; }
static constexpr bool stage16_shared(int C) { return true; }
static constexpr int stage16_nw(int C) { return 8; }  // waves per block
static int stage16_ntc(int C) { return 32 * stage16_nb(C) * (stage16_nw(C) / (C / 32)); }

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
  const int NTC = stage16_ntc(C);
  if (2 * o >= NTC || 2 * o * 100 > max_waste_pct * NTC) return 0;
  return NTC - 2 * o;
}

int32_t launch_resblock2_stage16(const PackedConvB* const* c1, const PackedConvB* const* c2, int nchain,
                                 ResStage2Params p, hipStream_t stream) {
  const int nto = resblock2_stage16_nto(c1, c2, nchain, 100);
  WETTS_REQUIRE(nto > 0, "ResBlock2 stage not supported by the fused kernel");
  const int C = c1[0]->Cin;
  const int NTC = stage16_ntc(C), RS = C * 2 + 16;
  const int RPP = 64 * stage16_nw(C) / (C / 8);
  p.nchain = nchain;
  int wmax = 0, h1max = 0;
  for (int j = 0; j < nchain; ++j) {
    p.wpk1[j] = c1[j]->wpk; p.bias1[j] = c1[j]->bias;
    p.wpk2[j] = c2[j]->wpk; p.bias2[j] = c2[j]->bias;
    p.ktaps[j] = c1[j]->ktaps;
    p.dil1[j] = c1[j]->dil;
    p.dil2[j] = c2[j]->dil;
    const int h1 = (p.ktaps[j] - 1) / 2 * p.dil1[j], h2 = (p.ktaps[j] - 1) / 2 * p.dil2[j];
    if (h1 > wmax) wmax = h1;
    if (h2 > wmax) wmax = h2;
    if (h1 > h1max) h1max = h1;
  }
  p.origin = (NTC - nto) / 2;
  p.h1max = h1max;
  p.ntiles = cdiv(p.T, nto);
  const int64_t nb = (int64_t)p.ntiles * p.B;
  if (nb <= 0) return WETTS_OK;
  WETTS_REQUIRE(nb < (1ll << 30), "resblock stage grid too large");
  WETTS_REQUIRE((int64_t)p.T * C * 2 < (int64_t)INT32_MAX, "utterance plane too large for the stage kernel's 32-bit offsets");
  p.nblocks = (int)nb;
  const unsigned grid = (unsigned)(((nb + 7) / 8) * 8);
  // LDS: the lrelu(x) tile in whole staging passes of RPP rows (its stores carry no row guard); shared staging adds the
  // lrelu(t) tile behind it: NTC columns + the widest c2 halo (= origin) on both sides
  const bool shared = stage16_shared(C);
  const int xneed = shared ? NTC + 2 * h1max : NTC + 2 * wmax;
  p.xrows = (xneed + RPP - 1) / RPP * RPP;
  const size_t lds = (size_t)(p.xrows + (shared ? NTC + 2 * p.origin : 0)) * RS;
  WETTS_REQUIRE(lds <= 160 * 1024, "ResBlock2 stage tile exceeds the LDS");
  const bool f16 = c1[0]->f16 != 0;
  auto go = [&](auto kern) -> int32_t {
    if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: opt in once per device
      static signed char state[4][64] = {};
      int dev = 0;
      (void)hipGetDevice(&dev);
      const int slot = (C == 32 ? 0 : 2) + (f16 ? 1 : 0);
      if (dev >= 0 && dev < 64 && !state[slot][dev]) {
        WETTS_HIP_CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        state[slot][dev] = 1;
      }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * stage16_nw(C)), lds, stream, p);
    return WETTS_OK;
  };
  if (C == 32) {
    if (f16) WETTS_TRY(go(rb2_stage16_kernel<32, true, 2, 2, stage16_nb(32), stage16_shared(32), stage16_nw(32)>));
    else WETTS_TRY(go(rb2_stage16_kernel<32, false, 2, 2, stage16_nb(32), stage16_shared(32), stage16_nw(32)>));
  } else {
    if (f16) WETTS_TRY(go(rb2_stage16_kernel<64, true, 2, 2, stage16_nb(64), stage16_shared(64), stage16_nw(64)>));
    else WETTS_TRY(go(rb2_stage16_kernel<64, false, 2, 2, stage16_nb(64), stage16_shared(64), stage16_nw(64)>));
  }
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
