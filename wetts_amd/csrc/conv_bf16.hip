// bf16 HiFi-GAN decoder path (BASELINE.json configs[2] / [4] precision): implicit-GEMM Conv1d /
// polyphase ConvTranspose1d on v_mfma_f32_32x32x16_bf16 with f32 accumulation.
//
// Layout: activations are CHANNEL-LAST bf16  X[b][t][C]  inside the decoder.  The MFMA B operand
// wants 8 consecutive K elements per lane; with K ordered (tap, ci) those are 8 consecutive input
// channels of one time step, i.e. one aligned 16-byte piece of a channel-last row -- so staging is
// plain 16-byte copies HBM -> registers -> LDS (leaky-relu applied on the way), B fragments are
// single ds_read_b128, and the epilogue writes 4 consecutive channels (8 bytes) per lane.
// LDS rows are padded by 16 bytes so the 16-lane groups of ds_read_b128 hit 16 distinct slots.
//
// At bf16 the per-conv arithmetic intensity is C*k/2 flop/B: k=3 launches are HBM-bound, k=11 at
// C>=128 are MFMA-bound (ridge ~400 flop/B) -- see DESIGN.md.
//
// Replaces, at reduced precision, the same reference code as conv_mfma.hip:
// decoders.py:63-82,157-170,205-214 (ups / ResBlock convs / conv_post).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

// ------------------------------------------------------------------------------------------
// weight packing: [mt32][g = chunk*ktaps + tap][ks][lane][8 bf16]
//   lane -> row = mt32*32 + (lane&31);  k elements = ci = chunk*CKB + ks*16 + 8*(lane>>5) + e
// transposed: rows are (phase, co) with co fastest; tap kk = phase + tap*up
// ------------------------------------------------------------------------------------------
__global__ void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ out,
                                 int M, int Cin, int Cout, int k, int ktaps, int up, int transposed,
                                 int CKB, int f16, int gate_h, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int KS = CKB / 16;
  int e = (int)(idx & 7);
  int lane = (int)((idx >> 3) & 63);
  int64_t rest = idx >> 9;
  int ks = (int)(rest % KS);
  rest /= KS;
  const int nchunks = (Cin + CKB - 1) / CKB;
  const int G = nchunks * ktaps;
  int g = (int)(rest % G);
  int mt32 = (int)(rest / G);
  int chunk = g / ktaps, tap = g % ktaps;
  int ci = chunk * CKB + ks * 16 + 8 * (lane >> 5) + e;
  // row permutation: MFMA row rho = 8q+4h+e carries channel 16*(q>>1) + 8h + 4*(q&1) + e of the m-block, so
  // one lane's 16 accumulator rows are two runs of 8 consecutive channels (16-byte stores)
  const int rho = lane & 31, q = rho >> 3, h = (rho >> 2) & 1;
  int row = mt32 * 32 + 16 * (q >> 1) + 8 * h + 4 * (q & 1) + (rho & 3);
  if (gate_h > 0) {
    // fused WN gate: m-block mt32 holds tanh channels 16 mt32 .. +15 (first run of a lane) and their
    // sigmoid partners gate_h + the same (second run)
    const int local = row - mt32 * 32, c = mt32 * 16 + (local & 15);
    row = c < gate_h ? (local >> 4) * gate_h + c : M;
  }
  float v = 0.f;
  if (row < M && ci < Cin) {
    if (!transposed) {
      v = w[((int64_t)row * Cin + ci) * k + tap];
    } else {
      int ph = row / Cout, co = row % Cout;
      int kk = ph + tap * up;
      if (kk < k) v = w[((int64_t)ci * Cout + co) * k + kk];
    }
  }
  out[idx] = f16 ? cv_out<true>(v) : cv_out<false>(v);
}

int32_t pack_conv_weight_bf16(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k,
                              int dil, int pad, int transposed, int up, int f16, hipStream_t stream,
                              PackedConvB* pc, int gate_h) {
  pc->f16 = f16;
  pc->gate_h = gate_h;
  pc->Cin = Cin;
  pc->Cout = Cout;
  pc->bias = bias_dev;
  pc->CKB = Cin >= 64 ? 64 : 32;
  if (!transposed) {
    pc->M = Cout; pc->ktaps = k; pc->dil = dil; pc->pad = pad; pc->up = 0; pc->up_pad = 0;
  } else {
    pc->M = Cout * up; pc->ktaps = cdiv(k, up); pc->dil = -1; pc->pad = 0; pc->up = up;
    pc->up_pad = pad;
  }
  int o0 = -pc->pad, o1 = (pc->ktaps - 1) * pc->dil - pc->pad;
  pc->off_lo = o0 < o1 ? o0 : o1;
  pc->span = (o0 < o1 ? o1 : o0) - pc->off_lo;
  pc->nchunks = cdiv(Cin, pc->CKB);
  const int G = pc->nchunks * pc->ktaps, KS = pc->CKB / 16;
  const int mt32 = cdiv(pc->M, 128) * 4;
  int64_t total = (int64_t)mt32 * G * KS * 64 * 8;
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->wpk, total * sizeof(unsigned short)));
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream,
                     w_dev, pc->wpk, pc->M, Cin, Cout, k, pc->ktaps, up > 0 ? up : 1, transposed,
                     pc->CKB, f16, gate_h, total);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

void free_packed_bf16(PackedConvB* pc) {
  if (pc->wpk) (void)hipFree(pc->wpk);
  pc->wpk = nullptr;
}

// ------------------------------------------------------------------------------------------
// the conv kernel: 4 waves (WM x WN), each wave one 32-row m-block x NB 32-column n-blocks
// ------------------------------------------------------------------------------------------
// tanh(a) * sigmoid(b) on the hardware exp / rcp (v_exp_f32, v_rcp_f32; ~1e-6 relative, far inside the
// 16-bit rounding that follows): in a conv epilogue the vector ALU work is paid in matrix-pipe time, and
// libm's tanhf / expf made the fused in_layer conv twice as long as the plain one.
//   tanh(a) = 2 / (1 + e^-2a) - 1,  sigmoid(b) = 1 / (1 + e^-b); both saturate correctly at +-inf.
__device__ __forceinline__ float gate_fast(float a, float b) {
  const float ea = __expf(-2.f * a), eb = __expf(-b);
  const float th = 2.f * __builtin_amdgcn_rcpf(1.f + ea) - 1.f;
  return th * __builtin_amdgcn_rcpf(1.f + eb);
}

// EPIM: 0 = plain epilogue, 1 / 2 = the fused WaveNet epilogues of the 16-bit flow (ConvBParams::epi_mode)
// MRFTAG = name tag of the decoder's ResBlock launches (no code difference): rocprofv3 then separates the MRF class
// from the upsamplers / flow convs that share the instantiation (as conv_mfma_kernel's MRF flag does at f32)
template <int NB, int WM, int WN, int CKB, bool F16, int EPIM = 0, bool MRFTAG = false>
// three waves per SIMD where the register allocation reaches it without heavy spilling (the compiler
// otherwise spreads over VGPRs + AGPRs and settles at two)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((CKB == 32 || WM == 4) ? 3 : 1)))
void conv_bf16_kernel(const ConvBParams p) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int MT = 32 * WM;
  constexpr int NT = 32 * NB * WN;
  constexpr int SEG = CKB / 8;          // 16-byte pieces per staged row
  constexpr int KS = CKB / 16;          // MFMA k-steps per (chunk, tap) group
  constexpr int RS = CKB * 2 + 16;      // LDS row stride in bytes (padded)
  constexpr int MAXU = (NT + 128 + 256 / SEG - 1) / (256 / SEG);  // staging passes of 256 / SEG rows

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5;

  const int ntiles = (p.N + NT - 1) / NT;
  const int mtiles = (p.M + MT - 1) / MT;
  int bid = blockIdx.x;
  const int ntile = bid % ntiles;
  bid /= ntiles;
  const int mtile = bid % mtiles;
  const int b = bid / mtiles;
  const int n0 = ntile * NT;
  const int W = NT + p.span;
  unsigned char* buf0 = smem_b;
  unsigned char* buf1 = smem_b + (size_t)p.lds_rows * RS;

  const unsigned short* xb = p.x + (int64_t)b * p.x_bs;

  // staging (round 6): unit (row, piece) = (tid / SEG + RPP * i, tid % SEG), i = 0 .. passes - 1.  The item's [Tin][Cin]
  // plane is addressed through a buffer descriptor (conv16_dev.h: plane_rsrc): rows before / behind it and -- offset
  // forced negative -- pieces of a channel chunk's tail behind Cin load zeros without a bounds test or exec mask per piece;
  // the LDS buffers are allocated in whole passes (p.lds_rows), so the stores carry no row guard either.  Before, every
  // piece was `branch, load, s_waitcnt vmcnt(0), convert, ds_write`: one memory round trip per piece.
  constexpr int RPP = 256 / SEG;
  const __amdgpu_buffer_rsrc_t rsx = plane_rsrc(xb, p.Tin * p.Cin * 2);
  const int useg = tid % SEG, urow = tid / SEG;  // 256 % SEG == 0, so the piece index does not depend on the pass
  const int vrow = ((n0 + p.off_lo + urow) * p.Cin + useg * 8) * 2;
  const int vpass = RPP * p.Cin * 2;
  uint4 st0[MAXU];
  auto load_chunk = [&](int c, uint4* st) __attribute__((always_inline)) {
    // (c beyond the last chunk: every offset negative -> zeros, no memory traffic; keeps the issue pattern uniform)
    const bool cok = c * CKB + useg * 8 < p.Cin && c < p.nchunks;
    const int v0 = cok ? vrow + c * (CKB * 2) : -(1 << 30);
#pragma unroll
    for (int i = 0; i < MAXU; ++i)
      if (i * RPP < W) st[i] = plane_load16(rsx, cok ? v0 + i * vpass : v0);
  };
  const bool lrelu = p.in_act == IN_LRELU;
  const f32x2v slope2 = {p.in_slope, p.in_slope};
  auto store_chunk = [&](unsigned char* buf, const uint4* st) __attribute__((always_inline)) {
    unsigned char* dst = buf + (size_t)urow * RS + useg * 16;
    auto go = [&](auto lr) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < MAXU; ++i)
        if (i * RPP < W) {
          uint4 v = st[i];
          if (decltype(lr)::value) {
            v.x = lrelu_pk2<F16>(v.x, slope2); v.y = lrelu_pk2<F16>(v.y, slope2);
            v.z = lrelu_pk2<F16>(v.z, slope2); v.w = lrelu_pk2<F16>(v.w, slope2);
          }
          *reinterpret_cast<uint4*>(dst + (size_t)i * (RPP * RS)) = v;
        }
    };
    if (lrelu) go(std::true_type{});
    else go(std::false_type{});
  };

  f32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- output geometry of this wave ---------------------------------------------------------
  const int mrow_blk = mtile * MT + wm * 32;  // first row of this wave's m-block (uniform)
  int ph = 0, co_blk = mrow_blk;              // transposed: rows are (phase, co)
  if (p.up > 0) {
    ph = mrow_blk / p.cout;
    co_blk = mrow_blk - ph * p.cout;
  }
  const int wcol0 = n0 + wn * (32 * NB);
  const bool rows_ok = mrow_blk + 32 <= p.M;

  // residual / running sum folded into the accumulator init (plain convs only).  Three phases -- all residual loads, all
  // running-sum loads, then the conversions and adds: as one loop (load, load, add per 16-byte piece) every piece was a
  // serialised memory round trip before the first MFMA (the f32 kernel's same defect cost the lone last-c2 launches 40 %
  // of their matrix time, profiles/r05_sq_counters_mrf.txt).  Same values, same additions: bit-identical.
  if (p.up == 0 && (p.res || p.accum) && rows_ok) {
    uint4 rr[NB][2], oo[NB][2];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wcol0 + 32 * j + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = co_blk + 16 * i + 8 * half;
        rr[j][i] = make_uint4(0u, 0u, 0u, 0u);
        if (p.res && t < p.N)
          rr[j][i] = *reinterpret_cast<const uint4*>(p.res + (int64_t)b * p.r_bs + (int64_t)t * p.cout + c);
      }
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wcol0 + 32 * j + (lane & 31);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = co_blk + 16 * i + 8 * half;
        oo[j][i] = make_uint4(0u, 0u, 0u, 0u);
        if (p.accum && t < p.N)
          oo[j][i] = *reinterpret_cast<const uint4*>(p.out + (int64_t)b * p.o_bs + (int64_t)t * p.cout + c);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wcol0 + 32 * j + (lane & 31);
      if (t < p.N) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = 0.f;
          if (p.res) {
            const unsigned w4[4] = {rr[j][i].x, rr[j][i].y, rr[j][i].z, rr[j][i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] = cv_in<F16>((unsigned short)(w4[e] & 0xffffu));
              v[2 * e + 1] = cv_in<F16>((unsigned short)(w4[e] >> 16));
            }
          }
          if (p.accum) {
            const unsigned w4[4] = {oo[j][i].x, oo[j][i].y, oo[j][i].z, oo[j][i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] += cv_in<F16>((unsigned short)(w4[e] & 0xffffu));
              v[2 * e + 1] += cv_in<F16>((unsigned short)(w4[e] >> 16));
            }
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[j][8 * i + e] = v[e];
        }
      }
    }
  }

  // ---- A stream ------------------------------------------------------------------------------
  const int G = p.nchunks * p.ktaps;
  const int mt32 = mtile * WM + wm;
  const uint4* abase = reinterpret_cast<const uint4*>(p.wpk) + ((int64_t)mt32 * G * KS) * 64 + lane;
  uint4 aa[2][KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) aa[0][s] = abase[s * 64];
#pragma unroll
  for (int s = 0; s < KS; ++s) aa[1][s] = aa[0][s];

  load_chunk(0, st0);
  store_chunk(buf0, st0);
  __syncthreads();

  // Main loop (round 6; the structure of resblock16.hip's conv loop).  Before, every MFMA was `ds_read_b128, s_waitcnt
  // lgkmcnt(0), v_mfma` -- the LDS latency exposed once per MFMA -- and the A prefetch sat behind a uniform branch, which
  // makes the compiler wait vmcnt(0) on the load it has just issued (a full L2 round trip per group).  Now: B fragments
  // one k-step ahead in a second register set (pinned by sched_barriers), the A prefetch of group g + 1 unconditional
  // with a clamped index, and the next chunk's staging loads issued at every chunk start (offsets forced out of range
  // behind the last chunk: no traffic) instead of behind a per-group test.
  const unsigned char* const bcol0 = buf0 + (size_t)(wn * (32 * NB) + (lane & 31) - p.pad - p.off_lo) * RS + half * 16;
  const size_t bufd = (size_t)p.lds_rows * RS;  // buf1 - buf0
  // ROLL (the 128-row tile with 64-channel chunks: acc 64 + A ring 32 + staging 32 registers leave no room for a second
  // B set at three waves per SIMD): ONE B set, each fragment re-requested right behind the MFMA that consumed it -- it is
  // needed again NB MFMAs later, which covers the LDS latency.  Otherwise: two B sets, a whole k-step ahead.
  constexpr bool ROLL = CKB == 64 && WM == 4;
  uint4 bq[ROLL ? 1 : 2][NB];
  auto b_load = [&](uint4 (&dst)[NB], const unsigned char* bb, int s) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < NB; ++j) dst[j] = *reinterpret_cast<const uint4*>(bb + (size_t)(32 * j) * RS + s * 32);
  };
  auto mma_group = [&](const uint4 (&av)[KS], const unsigned char* cur, const unsigned char* nxt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (ROLL) {
        const unsigned char* nb = s + 1 < KS ? cur + (s + 1) * 32 : nxt;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          acc[j] = mfma16<F16>(av[s], bq[0][j], acc[j]);
          bq[0][j] = *reinterpret_cast<const uint4*>(nb + (size_t)(32 * j) * RS);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        if (s + 1 < KS) b_load(bq[(s + 1) & 1], cur, s + 1);
        else b_load(bq[(s + 1) & 1], nxt, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j] = mfma16<F16>(av[s], bq[s & 1][j], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  int chunk = 0, tap = 0;
  b_load(bq[0], bcol0, 0);
  load_chunk(1, st0);
  auto group = [&](int gg, const uint4 (&av)[KS], uint4 (&an)[KS]) __attribute__((always_inline)) {
    int gn = gg + 1;
    gn = gn < G ? gn : G - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) an[s] = abase[((int64_t)gn * KS + s) * 64];
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch at the top of its group
    const unsigned char* cur = bcol0 + ((chunk & 1) ? bufd : 0) + (size_t)(tap * p.dil) * RS;
    const bool last_tap = tap + 1 == p.ktaps;
    // (the chunk's last group: the next tile is the other buffer, complete only behind the barrier -- a harmless
    // re-read of this group's own position, and the real fragment is fetched after the barrier)
    mma_group(av, cur, last_tap ? cur : cur + (size_t)p.dil * RS);
    if (last_tap) {
      tap = 0;
      if (chunk + 1 < p.nchunks) {
        store_chunk((chunk & 1) ? buf0 : buf1, st0);
        __syncthreads();
        ++chunk;
        b_load(bq[0], bcol0 + ((chunk & 1) ? bufd : 0), 0);
        load_chunk(chunk + 1, st0);  // in flight behind this chunk's MFMAs
      }
    } else {
      ++tap;
    }
  };
  int g = 0;
  for (; g + 2 <= G; g += 2) {  // ping-pong A register sets, statically indexed
    group(g, aa[0], aa[1]);
    group(g + 1, aa[1], aa[0]);
  }
  if (g < G) group(g, aa[0], aa[1]);

  // ---- epilogue: + bias, / div, round to 16 bit, channel-last stores ---------------------------
  if (!rows_ok) return;  // M is a multiple of 32 in every decoder conv; guard only
  if (EPIM == 1) {
    // WN gate: a lane's first run of 8 rows are tanh channels, the second their sigmoid partners
    // (pack_bf16_kernel, gate_h).  Both conv outputs are rounded to 16 bit first, as the unfused path
    // stores them; the gate itself uses the hardware exp / rcp (gate_fast), k_gate_cl16 libm.
    const int H = p.wn_H, mtb = mrow_blk >> 5, c0 = mtb * 16 + 8 * half;
    if (c0 >= H) return;
    float bt[8], bs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bt[e] = p.bias ? p.bias[c0 + e] : 0.f;
      bs[e] = p.bias ? p.bias[H + c0 + e] : 0.f;
    }
    if (p.bias_b) {
      const float* bb = p.bias_b + (int64_t)b * p.bias_b_stride;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        bt[e] += bb[c0 + e];
        bs[e] += bb[H + c0 + e];
      }
    }
    unsigned short* ob = p.out + (int64_t)b * p.o_bs;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wcol0 + 32 * j + (lane & 31);
      if (t >= p.N) continue;
      unsigned o[4];
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) {
        const unsigned ta = pk2<F16>(acc[j][2 * e2] + bt[2 * e2], acc[j][2 * e2 + 1] + bt[2 * e2 + 1]);
        const unsigned sa = pk2<F16>(acc[j][8 + 2 * e2] + bs[2 * e2], acc[j][9 + 2 * e2] + bs[2 * e2 + 1]);
        o[e2] = pk2<F16>(gate_fast(lo16<F16>(ta), lo16<F16>(sa)), gate_fast(hi16<F16>(ta), hi16<F16>(sa)));
      }
      *reinterpret_cast<uint4*>(ob + (int64_t)t * H + c0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
    return;
  }
  if (EPIM == 3) {
    // conv_post as a 32-row conv whose only non-zero weight row is row 0 (k_conv_post_mfma16): tanh -> f32
    if (half == 0) {
      float* of = p.out_f32 + (int64_t)b * p.Tout;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int t = wcol0 + 32 * j + (lane & 31);
        if (t < p.N) of[t] = tanhf(acc[j][0]);
      }
    }
    return;
  }
  if (EPIM == 2) {
    // WN residual / skip update on the rounded conv output (equals k_wn_update_cl16 on the stored tensor)
    const int H = p.wn_H;
    float bia[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bia[r] = p.bias ? p.bias[co_blk + 16 * (r >> 3) + 8 * half + (r & 7)] : 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int t = wcol0 + 32 * j + (lane & 31);
      if (t >= p.N) continue;
      const int64_t rowi = (int64_t)b * p.Tout + t;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = co_blk + 16 * i + 8 * half;
        unsigned w[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)
          w[e2] = pk2<F16>(acc[j][8 * i + 2 * e2] + bia[8 * i + 2 * e2],
                           acc[j][8 * i + 2 * e2 + 1] + bia[8 * i + 2 * e2 + 1]);
        if (!p.wn_last && c < H) {
          unsigned short* hp = p.wn_h + rowi * H + c;
          const uint4 hv = *reinterpret_cast<const uint4*>(hp);
          const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
          const float mk = p.wn_mask[rowi];
          unsigned o[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2)
            o[e2] = pk2<F16>((lo16<F16>(hw[e2]) + lo16<F16>(w[e2])) * mk,
                             (hi16<F16>(hw[e2]) + hi16<F16>(w[e2])) * mk);
          *reinterpret_cast<uint4*>(hp) = make_uint4(o[0], o[1], o[2], o[3]);
        } else {
          float4* sp = reinterpret_cast<float4*>(p.wn_skip + rowi * H + (p.wn_last ? c : c - H));
          float4 a = make_float4(lo16<F16>(w[0]), hi16<F16>(w[0]), lo16<F16>(w[1]), hi16<F16>(w[1]));
          float4 bq = make_float4(lo16<F16>(w[2]), hi16<F16>(w[2]), lo16<F16>(w[3]), hi16<F16>(w[3]));
          if (!p.wn_first) {
            const float4 pa = sp[0], pb = sp[1];
            a.x += pa.x; a.y += pa.y; a.z += pa.z; a.w += pa.w;
            bq.x += pb.x; bq.y += pb.y; bq.z += pb.z; bq.w += pb.w;
          }
          sp[0] = a;
          sp[1] = bq;
        }
      }
    }
    return;
  }
  const bool dodiv = p.out_div != 1.f;
  unsigned short* ob = p.out + (int64_t)b * p.o_bs;
  // lane rows = two runs of 8 consecutive channels (see pack_bf16_kernel) -> 16-byte stores
  float bia[16];
#pragma unroll
  for (int r = 0; r < 16; ++r)
    bia[r] = p.bias ? p.bias[co_blk + 16 * (r >> 3) + 8 * half + (r & 7)] : 0.f;
  if (p.bias_b) {
    const float* bb = p.bias_b + (int64_t)b * p.bias_b_stride;
#pragma unroll
    for (int r = 0; r < 16; ++r) bia[r] += bb[co_blk + 16 * (r >> 3) + 8 * half + (r & 7)];
  }
  auto store_all = [&](auto fin) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = wcol0 + 32 * j + (lane & 31);
      if (col >= p.N) continue;
      int t = col;
      if (p.up > 0) {
        t = col * p.up + ph - p.up_pad;
        if (t < 0 || t >= p.Tout) continue;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fin(acc[j][8 * i + e] + bia[8 * i + e]);
        uint4 o;
        o.x = pk2<F16>(v[0], v[1]); o.y = pk2<F16>(v[2], v[3]);
        o.z = pk2<F16>(v[4], v[5]); o.w = pk2<F16>(v[6], v[7]);
        *reinterpret_cast<uint4*>(ob + (int64_t)t * p.cout + co_blk + 16 * i + 8 * half) = o;
      }
    }
  };
  const float dv = p.out_div, dinv = 1.f / p.out_div;  // the MRF mean (common.h: mrf_div), uniform choice
  if (!dodiv) store_all([](float v) { return v; });
  else if (mrf_div_fast(dv)) store_all([=](float v) { return div_small_const(v, dv, dinv); });
  else store_all([=](float v) { return v / dv; });
}

template <int NB, int WM, int WN, int CKB>
static int32_t launch_b(const ConvBParams& p, hipStream_t stream, bool f16) {
  constexpr int MT = 32 * WM, NT = 32 * NB * WN, RS = CKB * 2 + 16;
  int64_t blocks = (int64_t)cdiv(p.N, NT) * cdiv(p.M, MT) * p.B;
  if (blocks <= 0) return WETTS_OK;
  WETTS_REQUIRE(blocks < (1ll << 31), "conv grid too large");
  // one staging buffer is enough when the whole reduction is a single channel chunk
  const int nbuf = p.nchunks > 1 ? 2 : 1;
  constexpr int RPP = 256 / (CKB / 8);  // a buffer = whole staging passes (the kernel's stores carry no row guard)
  ConvBParams pl = p;
  pl.lds_rows = (NT + p.span + RPP - 1) / RPP * RPP;
  WETTS_REQUIRE((int64_t)p.Tin * p.Cin * 2 < (int64_t)INT32_MAX, "input plane too large for the 16-bit conv's 32-bit offsets");
  size_t lds = (size_t)nbuf * pl.lds_rows * RS;
  WETTS_REQUIRE(lds <= 64 * 1024, "16-bit conv tile exceeds the default dynamic LDS");
  const dim3 grid((unsigned)blocks), blk(256);
  if constexpr (NB == 4 && WM == 4 && WN == 1 && CKB == 64) {  // the shape of the flow's WN convs
    if (p.epi_mode == 1) {
      if (f16) hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, true, 1>), grid, blk, lds, stream, pl);
      else hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, false, 1>), grid, blk, lds, stream, pl);
      WETTS_LAUNCH_CHECK();
      return WETTS_OK;
    }
    if (p.epi_mode == 2) {
      if (f16) hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, true, 2>), grid, blk, lds, stream, pl);
      else hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, false, 2>), grid, blk, lds, stream, pl);
      WETTS_LAUNCH_CHECK();
      return WETTS_OK;
    }
  }
  if constexpr (NB == 4 && WM == 1 && WN == 4 && CKB == 32) {  // conv_post (k_conv_post_mfma16)
    if (p.epi_mode == 3) {
      if (f16) hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, true, 3>), grid, blk, lds, stream, pl);
      else hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, false, 3>), grid, blk, lds, stream, pl);
      WETTS_LAUNCH_CHECK();
      return WETTS_OK;
    }
  }
  WETTS_REQUIRE(p.epi_mode == 0, "fused epilogue requested for a tile shape it is not instantiated for");
  if (p.tag) {
    if (f16)
      hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, true, 0, true>), grid, blk, lds, stream, pl);
    else
      hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, false, 0, true>), grid, blk, lds, stream, pl);
    WETTS_LAUNCH_CHECK();
    return WETTS_OK;
  }
  if (f16)
    hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, true>), grid, blk, lds, stream, pl);
  else
    hipLaunchKernelGGL((conv_bf16_kernel<NB, WM, WN, CKB, false>), grid, blk, lds, stream, pl);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

int32_t launch_conv_bf16(const PackedConvB& pc, ConvBParams p, hipStream_t stream) {
  p.wpk = pc.wpk;
  p.bias = pc.bias;
  p.M = pc.M;
  p.Cin = pc.Cin;
  p.cout = pc.Cout;
  p.ktaps = pc.ktaps;
  p.dil = pc.dil;
  p.pad = pc.pad;
  p.off_lo = pc.off_lo;
  p.span = pc.span;
  p.nchunks = pc.nchunks;
  p.up = pc.up;
  p.up_pad = pc.up_pad;
  WETTS_REQUIRE(pc.wpk != nullptr, "bf16 conv weight not packed");
  WETTS_REQUIRE((pc.gate_h > 0) == (p.epi_mode == 1), "gate-packed weights and the gate epilogue go together");
  WETTS_REQUIRE(p.span <= 128, "conv receptive field too wide");
  WETTS_REQUIRE(pc.Cout % 32 == 0 && pc.Cin % 8 == 0, "bf16 path needs Cout %% 32 == 0, Cin %% 8 == 0");
  p.N = p.up > 0 ? p.Tin + p.ktaps - 1 : p.Tout;
  if (pc.CKB == 64) {
    if (p.M >= 128) return launch_b<4, 4, 1, 64>(p, stream, pc.f16 != 0);
    return launch_b<4, 2, 2, 64>(p, stream, pc.f16 != 0);
  }
  if (p.M >= 128) return launch_b<4, 4, 1, 32>(p, stream, pc.f16 != 0);
  if (p.M >= 64) return launch_b<4, 2, 2, 32>(p, stream, pc.f16 != 0);
  return launch_b<4, 1, 4, 32>(p, stream, pc.f16 != 0);
}

// ------------------------------------------------------------------------------------------
// layout / precision boundaries of the bf16 decoder
// ------------------------------------------------------------------------------------------
// f32 channel-first [B,C,T] (batch / channel strides x_bs / x_cs, optional row mask [B][>= T]) -> 16-bit channel-last
// [B,T,C]: (x * mask) rounded once
template <bool F16>
__global__ void cf32_to_cl16_kernel(const float* __restrict__ x, int64_t x_bs, int64_t x_cs, const float* __restrict__ mask,
                                    int64_t mask_stride, unsigned short* __restrict__ out, int B, int C, int T) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, c8, t) with t fastest
  const int C8 = C / 8;
  int64_t total = (int64_t)B * C8 * T;
  if (idx >= total) return;
  int t = (int)(idx % T);
  int c8 = (int)((idx / T) % C8);
  int b = (int)(idx / ((int64_t)T * C8));
  const float* xp = x + (int64_t)b * x_bs + (int64_t)(c8 * 8) * x_cs + t;
  const float mk = mask ? mask[(int64_t)b * mask_stride + t] : 1.f;
  uint4 o;
  o.x = pk2<F16>(xp[0] * mk, xp[x_cs] * mk);
  o.y = pk2<F16>(xp[2 * x_cs] * mk, xp[3 * x_cs] * mk);
  o.z = pk2<F16>(xp[4 * x_cs] * mk, xp[5 * x_cs] * mk);
  o.w = pk2<F16>(xp[6 * x_cs] * mk, xp[7 * x_cs] * mk);
  *reinterpret_cast<uint4*>(out + ((int64_t)b * T + t) * C + c8 * 8) = o;
}

int32_t k_cf32_to_cl16_strided(const float* x, int64_t x_bs, int64_t x_cs, const float* mask, int64_t mask_stride,
                               unsigned short* out, int B, int C, int T, int f16, hipStream_t s) {
  WETTS_REQUIRE(C % 8 == 0, "channel count must be a multiple of 8");
  int64_t n = (int64_t)B * (C / 8) * T;
  if (n == 0) return WETTS_OK;
  if (f16)
    hipLaunchKernelGGL(cf32_to_cl16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       x, x_bs, x_cs, mask, mask_stride, out, B, C, T);
  else
    hipLaunchKernelGGL(cf32_to_cl16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       s, x, x_bs, x_cs, mask, mask_stride, out, B, C, T);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

int32_t k_cf32_to_cl16(const float* x, unsigned short* out, int B, int C, int T, int f16,
                       hipStream_t s) {
  return k_cf32_to_cl16_strided(x, (int64_t)C * T, T, nullptr, 0, out, B, C, T, f16, s);
}

// conv_post on channel-last bf16: lrelu(0.01) -> Conv1d(C,1,k) -> tanh -> f32 audio [B,T]
template <bool F16>
__global__ __launch_bounds__(256) void conv_post_bf16_kernel(const unsigned short* __restrict__ x,
                                                             const float* __restrict__ w, int k,
                                                             int B, int C, int T,
                                                             float* __restrict__ out) {
  extern __shared__ float wsh[];  // [j][c]
  for (int i = threadIdx.x; i < C * k; i += blockDim.x) {
    int c = i / k, j = i % k;
    wsh[j * C + c] = w[i];
  }
  __syncthreads();
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)B * T) return;
  const int b = (int)(idx / T), t = (int)(idx % T);
  const int pad = (k - 1) / 2;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const int tt = t + j - pad;
    if (tt < 0 || tt >= T) continue;
    const unsigned short* row = x + ((int64_t)b * T + tt) * C;
    for (int c = 0; c < C; c += 8) {
      const uint4 v = *reinterpret_cast<const uint4*>(row + c);
      const unsigned wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo = cv_in<F16>((unsigned short)(wv[e] & 0xffffu)), hi = cv_in<F16>((unsigned short)(wv[e] >> 16));
        lo = lo > 0.f ? lo : lo * 0.01f;
        hi = hi > 0.f ? hi : hi * 0.01f;
        acc += wsh[j * C + c + 2 * e] * lo + wsh[j * C + c + 2 * e + 1] * hi;
      }
    }
  }
  out[idx] = tanhf(acc);
}

// conv_post on the matrix cores: lrelu(0.01) (rounded to 16 bit, as every conv input of this mode) ->
// Conv1d(32, 1, 7) with 16-bit weights, f32 accumulation -> tanh -> f32.  `pc` packs a 32-row weight
// whose rows 1..31 are zero; the vector-ALU kernel below spends ~1100 lane operations per sample on
// conversions, leaky-relu and broadcast weight reads (0.6 ms per step at B = 64, v3), this one is
// bound by reading the activations once.
int32_t k_conv_post_mfma16(const PackedConvB& pc, const unsigned short* x, int B, int C, int T,
                           float* out, hipStream_t s) {
  WETTS_REQUIRE(pc.wpk && pc.Cin == C && pc.Cout == 32 && pc.CKB == 32, "conv_post: weights not packed for this shape");
  ConvBParams p;
  memset(&p, 0, sizeof(p));
  p.x = x; p.x_bs = (int64_t)C * T; p.Cin = C; p.Tin = T;
  p.in_act = IN_LRELU; p.in_slope = 0.01f;  // F.leaky_relu default slope, decoders.py:78
  p.Tout = T; p.out_div = 1.f; p.B = B;
  p.epi_mode = 3;
  p.out_f32 = out;
  return launch_conv_bf16(pc, p, s);
}

int32_t k_conv_post_bf16(const unsigned short* x, const float* w, int k, int B, int C, int T,
                         float* out, int f16, hipStream_t s) {
  WETTS_REQUIRE(C % 8 == 0 && k <= 15, "conv_post bf16: unsupported shape");
  int64_t n = (int64_t)B * T;
  if (n == 0) return WETTS_OK;
  if (f16)
    hipLaunchKernelGGL(conv_post_bf16_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256),
                       (size_t)C * k * 4, s, x, w, k, B, C, T, out);
  else
    hipLaunchKernelGGL(conv_post_bf16_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256),
                       (size_t)C * k * 4, s, x, w, k, B, C, T, out);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
