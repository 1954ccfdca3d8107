// Dense Conv1d / ConvTranspose1d for launches that cannot fill the chip with 64x64 tiles: B = 1 streaming
// windows (vits_model.cc:128-153, inference_onnx.py:37-76), the text encoder / duration predictor / flow
// at short texts.  Same GEMM view, packed weights and results as conv_mfma_kernel (conv_mfma.hip), other
// schedule.
//
// What bounds such a launch (profiles/r02_b1_anatomy.txt: 26-44 us per conv, launch gaps 3-10 %): with
// 64x64 tiles a 192 -> 768 conv over 64 frames is THREE blocks, and a block's matrix work runs at the
// rate of one CU (0.6 TFLOP/s): 64 x 64 x K MACs take 13 ns per unit of K, 30 us at K = 2304, while 250
// CUs idle.  K-splitting inside a 64x64 block does not help (the four SIMDs of the CU are the limit;
// measured) and splitting K across blocks costs two device-scope fences per launch (L2 write-back /
// invalidate between XCDs; measured slower than not splitting).  So: the smallest tile the matrix core
// offers and the K range split over the block's waves --
//  * tile 32 x 32, one block = KG waves (4, 8 or 16) = K-groups: wave w runs stages w, w + KG, ... of the
//    reduction (a stage = one 16-channel chunk with all its taps; four chunks for 1x1 convs) with its own
//    LDS double buffer, staged by the wave itself -- no block barrier in the K loop.  4x as many blocks
//    as 64x64 tiles, each wave's chain K / (2 KG) MFMAs instead of K / 2.  8 / 16 waves serve long
//    reductions with short taps: a k = 3 stage is 0.6 us of matrix work, less than the round trip of its
//    staging loads, so a wave's time is (stages per wave) x (load latency);
//  * the KG partial tiles meet in LDS; wave w sums accumulator registers w 16/KG .. of all waves in wave
//    order (deterministic) and runs the epilogue for them;
//  * a register ring of R weight fragments per wave (R = all groups of a stage for k <= 5, half of them
//    for k = 7 / 11): the prefetch runs a stage ahead.  The tap count is a template parameter so that
//    ring slots, loop structure and therefore the s_waitcnt counts are static; every load is
//    unconditional (indices clamped, past-the-end stages multiply the weights' zero tail with a zeroed
//    tile) for the same reason ("code-generation trap").
#include <stdlib.h>

#include "common.h"

namespace wetts {

typedef float f32x16s __attribute__((ext_vector_type(16)));

template <int KT, int SCH, int KG>
__global__ __launch_bounds__(64 * KG) void conv_small_kernel(const ConvParams p) {
  constexpr int CK = kConvCK;           // KG: K-groups = waves per block (4, 8 or 16)
  constexpr int NT = 32, MT = 32;
  constexpr int ROWS = CK * SCH;        // staged rows per stage
  constexpr int RP = ROWS / 2;          // row pairs: lanes 0-31 stage row 2i, lanes 32-63 row 2i+1
  constexpr int MAXCS = KT == 1 ? 1 : 5; // 32-column slots per row: 32 + span <= 160 columns (span = 0 for 1x1)
  constexpr int GPS = KT * 2 * SCH;     // groups (4 k-steps each) per stage
  constexpr int R = GPS <= 10 ? GPS : GPS / 2;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int lane = threadIdx.x & 63;
  const int kg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l32 = lane & 31;

  const int ntiles = (p.N + NT - 1) / NT;
  const int mtiles = (p.M + MT - 1) / MT;
  int bid = blockIdx.x;
  const int ntile = bid % ntiles;
  bid /= ntiles;
  const int mtile = bid % mtiles;
  const int b = bid / mtiles;
  const int n0 = ntile * NT;
  const int W = NT + p.span;
  const int CS = (W + 31) >> 5;  // 32-column slots per row (uniform)
  float* buf0 = smem + kg * (2 * ROWS * W);
  float* buf1 = buf0 + ROWS * W;

  const float* xb = p.x + (int64_t)b * p.x_bs;
  const float* mrow = p.in_mask ? p.in_mask + (int64_t)b * p.in_mask_stride : nullptr;

  // per-column info (stage independent): clamped time + validity / mask factor
  int tcl[MAXCS];
  float mcol[MAXCS];
#pragma unroll
  for (int i = 0; i < MAXCS; ++i) {
    const int col = l32 + 32 * i;
    const int t = n0 + p.off_lo + col;
    const bool ok = (col < W) && (t >= 0) && (t < p.Tin);
    tcl[i] = ok ? t : 0;
    mcol[i] = ok ? (mrow ? mrow[t] : 1.f) : 0.f;
  }
  const int NS = p.nchunks / SCH;  // stages (the launcher guarantees divisibility)

  // One stage through registers: CS is uniform but not a compile-time constant, so the slots a conv does
  // not have are skipped with uniform branches around whole load / store groups (no loads inside
  // divergent control flow, and the same branch pattern every stage).
  float stage[RP][MAXCS];
  auto load_stage = [&](int st) {  // unconditional: a stage index past the end re-reads stage 0
    st = st < NS ? st : 0;
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      int ci = st * ROWS + 2 * r + half;
      ci = ci < p.Cin ? ci : 0;  // (channels past Cin meet zero weights)
      const int ch = p.in_rev_base >= 0 ? (p.in_rev_base - ci) : ci;
      const float* xr = xb + (int64_t)ch * p.x_cs;
#pragma unroll
      for (int i = 0; i < MAXCS; ++i)
        if (i < CS) stage[r][i] = xr[tcl[i]];
    }
  };
  auto store_stage = [&](float* buf, float keep) {  // keep = 0: a stage past the end stores zeros
#pragma unroll
    for (int r = 0; r < RP; ++r) {
      float* row = buf + (2 * r + half) * W;
#pragma unroll
      for (int i = 0; i < MAXCS; ++i) {
        const int col = l32 + 32 * i;
        if (i < CS && col < W) {
          float v = stage[r][i];
          if (p.in_act == IN_LRELU) v = v > 0.f ? v : v * p.in_slope;
          row[col] = v * (mcol[i] * keep);
        }
      }
    }
  };

  f32x16s acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // packed A stream of this tile's 32 rows: [mt32][G][64 lanes][4]
  const int G = p.nchunks * KT * 2;
  const float4* abase = reinterpret_cast<const float4*>(p.wpk) + ((int64_t)mtile * G) * 64 + lane;
  auto gclamp = [&](int st) { const int g = st * GPS; return g < G ? g : G; };  // group G: the zero tail

  float4 ring[R];
  {
    const int g0 = gclamp(kg);
#pragma unroll
    for (int j = 0; j < R; ++j) ring[j] = abase[(int64_t)(g0 + j) * 64];
  }
  load_stage(kg);
  store_stage(buf0, kg < NS ? 1.f : 0.f);
  __builtin_amdgcn_wave_barrier();

  const int bcol0 = l32 - p.pad - p.off_lo;
  const int iters = (NS + KG - 1) / KG;
  for (int it = 0; it < iters; ++it) {
    const int st = it * KG + kg;
    const float* cur = (it & 1) ? buf1 : buf0;
    const int gnext = gclamp(st + KG), gcur = gclamp(st);
    load_stage(st + KG);
#pragma unroll
    for (int gi = 0; gi < GPS; ++gi) {
      const float4 a = ring[gi % R];
      const int chunk_local = gi / (2 * KT), tap = (gi % (2 * KT)) / 2, hp = gi & 1;
      const float* brow0 = cur + (chunk_local * CK + hp * 8 + half) * W + bcol0 + tap * p.dil;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float bv = brow0[s * 2 * W];
        const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // refill the slot (behind its last use, so the load lands in the same registers) with the group R
      // ahead: later in this stage, or in this wave's next stage
      ring[gi % R] = (gi + R < GPS) ? abase[(int64_t)(gcur + gi + R) * 64]
                                    : abase[(int64_t)(gnext + gi + R - GPS) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    store_stage((it & 1) ? buf0 : buf1, st + KG < NS ? 1.f : 0.f);
    __builtin_amdgcn_wave_barrier();  // the wave's own LDS traffic is processed in order
  }

  // the KG partial tiles meet in LDS ([wave][r][lane]); wave w finishes accumulator registers
  // w*16/KG .. (w+1)*16/KG - 1, adding the partials in wave order
  constexpr int FPW = 16 / KG;
  __syncthreads();  // every wave is done with its staging buffers
  {
    float* mine = smem + kg * 1024 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
  }
  __syncthreads();
  float fin[FPW];
#pragma unroll
  for (int q = 0; q < FPW; ++q) {
    const float* src = smem + (FPW * kg + q) * 64 + lane;
    float v = src[0];
#pragma unroll
    for (int w2 = 1; w2 < KG; ++w2) v += src[w2 * 1024];
    fin[q] = v;
  }

  // ---- epilogue (generic: bias, per-utterance bias, activation, mask, residual, running sum, mean,
  //      polyphase ConvTranspose1d store) ------------------------------------------------------------
  const int64_t ob = (int64_t)b * p.o_bs;
  const int64_t rb = (int64_t)b * p.r_bs;
  const float* bb = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_stride : nullptr;
  const float* omask = p.out_mask ? p.out_mask + (int64_t)b * p.out_mask_stride : nullptr;
  const int col = n0 + l32;
  if (col >= p.N) return;
#pragma unroll
  for (int q = 0; q < FPW; ++q) {
    const int r = FPW * kg + q;  // accumulator register r holds row (r & 3) + 8 (r >> 2) + 4 half
    const int row = mtile * MT + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (row >= p.M) continue;
    int co = row, t = col;
    if (p.up > 0) {
      co = row / p.up;
      t = col * p.up + (row - co * p.up) - p.up_pad;
      if (t < 0 || t >= p.Tout) continue;
    }
    float v = fin[q];
    if (p.out_act == OUT_GATE) {  // (tanh, sigmoid) rows of output row row / 2 in registers (r, r + 1): FPW >= 2
      if (q & 1) continue;
      const int gi = row >> 1, gH = p.M >> 1;
      float sa = fin[(q + 1) % FPW];
      if (p.bias) { v += p.bias[gi]; sa += p.bias[gH + gi]; }
      if (bb) { v += bb[gi]; sa += bb[gH + gi]; }
      p.out[ob + (int64_t)gi * p.o_cs + t] = wn_gate(v, sa);
      continue;
    }
    if (p.bias) v += p.bias[co];
    if (bb) v += bb[co];
    if (p.wn_skip) {  // WaveNet residual / skip update instead of a store (common.h)
      const int64_t hb = (int64_t)b * p.wn_H * p.Tout + t;
      if (!p.wn_last && row < p.wn_H) {
        float* hp = p.wn_h + hb + (int64_t)row * p.Tout;
        *hp = (*hp + v) * p.wn_mask[(int64_t)b * p.wn_mask_stride + t];
      } else {
        float* sp = p.wn_skip + hb + (int64_t)(p.wn_last ? row : row - p.wn_H) * p.Tout;
        *sp = p.wn_first ? v : *sp + v;
      }
      continue;
    }
    if (p.out_act == OUT_RELU) v = v > 0.f ? v : 0.f;
    if (p.out_act == OUT_GELU) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (omask) v *= omask[t];
    float* dst = p.out + ob + (int64_t)co * p.o_cs + t;
    if (p.res) v += p.res[rb + (int64_t)co * p.r_cs + t];
    if (p.accum) v += *dst;
    if (p.out_div != 1.f) v = mrf_div(v, p.out_div, 1.f / p.out_div, mrf_div_fast(p.out_div));
    *dst = v;
  }
}

// launches of at most this many 64x64 tiles (four times as many 32x32 blocks) take this kernel (0: never)
// (per calling thread, set for the duration of a C-ABI call from the model's setting: no process-wide state)
static thread_local int tls_small_max_tiles = 256;
SmallConvScope::SmallConvScope(int max_tiles) : prev(tls_small_max_tiles) { tls_small_max_tiles = max_tiles; }
SmallConvScope::~SmallConvScope() { tls_small_max_tiles = prev; }

static size_t small_lds_bytes(int sch, int span, int kg) {
  const size_t stage = (size_t)kg * 2 * kConvCK * sch * (32 + span) * sizeof(float);
  const size_t red = (size_t)kg * 1024 * sizeof(float);
  return stage > red ? stage : red;
}

template <int KT, int SCH, int KG>
static int32_t launch_small_kg(const ConvParams& p, hipStream_t stream) {
  const int64_t blocks = (int64_t)cdiv(p.M, 32) * cdiv(p.N, 32) * p.B;
  const size_t lds = small_lds_bytes(SCH, p.span, KG);
  static bool attr_done[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (lds > 64 * 1024 && dev >= 0 && dev < 64 && !attr_done[dev]) {
    WETTS_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_small_kernel<KT, SCH, KG>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev] = true;
  }
  hipLaunchKernelGGL((conv_small_kernel<KT, SCH, KG>), dim3((unsigned)blocks), dim3(64 * KG), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// Waves per block: a stage of a short-tap conv is well under a microsecond of matrix work, less than the
// round trip of its staging loads, so a wave's time is (stages per wave) x (load latency): long
// reductions get 8 or 16 waves (2 / 4 per SIMD) as long as the blocks do not fill the chip anyway and
// LDS / registers allow (MAXKG: 8 for k >= 7).
template <int KT, int SCH, int MAXKG>
static int32_t launch_small_cfg(const ConvParams& p, hipStream_t stream, int force_kg = 0) {
  const int64_t blocks = (int64_t)cdiv(p.M, 32) * cdiv(p.N, 32) * p.B;
  const int NS = p.nchunks / SCH;
  // stages from which 8 / 16 waves are used.  8 waves from 8 stages (round 5; was 16): the flow's in_layers (12 stages)
  // and the C = 128 window convs get the 8-wave split -- B = 1 encoder call 1.491 -> 1.455 ms, first window 1.054 ->
  // 1.025, first chunk 2.54 -> 2.48 ms same box (profiles/r05_small_launch_knobs.txt; the sweep's switches are gone)
  constexpr int ns8 = 8, ns16 = 32;
  int kg = 4;
  if (NS >= ns8 && blocks * 2 <= 512) kg = 8;
  if (NS >= ns16 && blocks * 4 <= 512) kg = 16;
  if (force_kg) kg = force_kg;
  if (kg > MAXKG) kg = MAXKG;
  if (p.out_act == OUT_GATE && kg > 8) kg = 8;  // the gate pairs two accumulator registers of one wave's share
  while (kg > 4 && small_lds_bytes(SCH, p.span, kg) > 128 * 1024) kg >>= 1;
  if constexpr (MAXKG >= 16)
    if (kg == 16) return launch_small_kg<KT, SCH, 16>(p, stream);
  if constexpr (MAXKG >= 8)
    if (kg == 8) return launch_small_kg<KT, SCH, 8>(p, stream);
  return launch_small_kg<KT, SCH, 4>(p, stream);
}

static int64_t blocks_of(const ConvParams& p) { return (int64_t)cdiv(p.M, 32) * cdiv(p.N, 32) * p.B; }

// p: geometry filled by launch_conv.  *taken = false when the shape is not one this kernel handles
// (the caller then runs conv_mfma_kernel).
int32_t launch_conv_small(const ConvParams& p, hipStream_t stream, bool* taken) {
  *taken = false;
  const int64_t tiles64 = (int64_t)cdiv(p.M, 64) * cdiv(p.N, 64) * p.B;
  if (tls_small_max_tiles <= 0 || tiles64 <= 0 || tiles64 > tls_small_max_tiles) return WETTS_OK;
  if (p.span > 128 || p.nchunks < 2) return WETTS_OK;
  const bool sch4 = p.ktaps == 1 && (p.nchunks % 4) == 0;
  if (small_lds_bytes(sch4 ? 4 : 1, p.span, 4) > 160 * 1024) return WETTS_OK;
  *taken = true;
  switch (p.ktaps) {
    case 1: {
      // 1x1 convs of launches with at most 256 blocks: one-chunk stages over 8 waves instead of four-chunk stages
      // over 4 (of which one idles at 12 chunks): a wave's chain is 1-2 short stages instead of one long one --
      // B = 1 encoder call 1.64 -> 1.51 ms (profiles/r03_small_1x1_ab.txt: 8 waves beat 4 and 16).
      if (blocks_of(p) * 4 <= 1024) return launch_small_cfg<1, 1, 16>(p, stream, 8);
      return sch4 ? launch_small_cfg<1, 4, 8>(p, stream) : launch_small_cfg<1, 1, 16>(p, stream);
    }
    case 2: return launch_small_cfg<2, 1, 16>(p, stream);
    case 3: return launch_small_cfg<3, 1, 16>(p, stream);
    case 5: return launch_small_cfg<5, 1, 16>(p, stream);
    case 7: return launch_small_cfg<7, 1, 8>(p, stream);
    case 11: return launch_small_cfg<11, 1, 8>(p, stream);
    default: break;
  }
  *taken = false;
  return WETTS_OK;
}

}  // namespace wetts
