// Implicit-GEMM Conv1d / ConvTranspose1d on the gfx950 f32 matrix cores.
//
// Replaces every dense Conv1d / ConvTranspose1d the reference runs through ATen on the
// infer() path: HiFi-GAN conv_pre / ups / ResBlock convs (decoders.py:63-82,157-170,205-214),
// WN in/res_skip layers (modules.py:66-85), coupling pre/post (flows.py:496-498), the encoder's
// 1x1 q/k/v/o, FFN and proj convs (attentions.py:225-233,403-411; encoders.py:54) and the
// duration predictors' convs (duration_predictors.py:50-54,297-310).
//
// Why f32 MFMA: the Baker config is quoted at fp32 with a 1e-3 waveform tolerance; per-conv
// arithmetic intensity is 28-113 flop/B, above the f32 ridge (157 TF / 8 TB/s ~ 20 flop/B), so the
// stack is compute-bound and v_mfma_f32_32x32x2_f32 (exact f32 fma chain, 157 TF peak, ~2.4x a
// VALU kernel) is the binding unit.  Data movement follows the HBM rules: each activation tile
// is read once per (m-tile) into LDS with the pre-activation (leaky-relu / mask) applied while
// staging, weights stream from L2 in pre-packed fragment order (one dwordx4 per lane = 4 k-steps).
//
// Tile: 256 threads = 4 waves arranged WM x WN; each wave owns MB x NB 32x32 accumulators.
// K loop: chunks of kConvCK=16 input channels; per chunk all taps re-use the same LDS tile
// (columns shifted by tap*dil), so a k-tap conv reads its input once, not k times.
#include <stdlib.h>

#include "common.h"

namespace wetts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kBufRsrcDword3 = 0x00020000;  // gfx9 family raw buffer: 32-bit data format, no swizzle

// max(x, slope*x) == (x > 0 ? x : slope*x) for 0 <= slope <= 1: two VALU ops instead of three
// (v_max_f32 through asm: no canonicalising v_max x, x in front of it)
__device__ __forceinline__ float lrelu2(float x, float slope) {
  const float m = x * slope;
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(m));
  return r;
}

// ------------------------------------------------------------------------------------------
// weight packing
// layout: [mt32][g][lane][4]  with kstep = g*4+s, chunk = kstep / (ktaps*8),
//         tap = (kstep % (ktaps*8)) / 8, pair = kstep % 8,
//         lane -> row = mt32*32 + (lane&31), ci = chunk*16 + pair*2 + (lane>>5)
// ------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out,
                                        int M, int Cin, int Cout, int k, int ktaps, int up,
                                        int transposed, int rev_in, int gate_H, int G, int64_t total) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  int s = (int)(idx & 3);
  int lane = (int)((idx >> 2) & 63);
  int64_t rest = idx >> 8;
  int g = (int)(rest % G);
  int mt32 = (int)(rest / G);
  int kstep = g * 4 + s;
  int per_chunk = ktaps * 8;
  int chunk = kstep / per_chunk;
  int within = kstep % per_chunk;
  int tap = within / 8;
  int pair = within % 8;
  int ci = chunk * kConvCK + pair * 2 + (lane >> 5);
  int row = mt32 * 32 + (lane & 31);
  float v = 0.f;
  if (row < M && ci < Cin) {
    if (!transposed) {
      const int src = gate_H > 0 ? (row & 1) * gate_H + (row >> 1) : row;
      v = w[((int64_t)src * Cin + (rev_in ? Cin - 1 - ci : ci)) * k + tap];
    } else {
      int co = row / up, ph = row % up;
      int kk = ph + tap * up;
      if (kk < k) v = w[((int64_t)ci * Cout + co) * k + kk];
    }
  }
  out[idx] = v;
}

int32_t pack_conv_weight(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k,
                         int dil, int pad, int transposed, int up, hipStream_t stream,
                         PackedConv* pc, int rev_in, int gate_H) {
  WETTS_REQUIRE(gate_H == 0 || (!transposed && Cout == 2 * gate_H), "gate packing needs a Conv1d with 2 * gate_H rows");
  pc->gate_H = gate_H;
  pc->Cin = Cin;
  pc->Cout = Cout;
  pc->k_orig = k;
  pc->bias = bias_dev;
  if (!transposed) {
    pc->M = Cout;
    pc->ktaps = k;
    pc->dil = dil;
    pc->pad = pad;
    pc->up = 0;
    pc->up_pad = 0;
  } else {
    pc->M = Cout * up;
    pc->ktaps = cdiv(k, up);
    pc->dil = -1;
    pc->pad = 0;
    pc->up = up;
    pc->up_pad = pad;
  }
  int o0 = -pc->pad, o1 = (pc->ktaps - 1) * pc->dil - pc->pad;
  pc->off_lo = o0 < o1 ? o0 : o1;
  pc->span = (o0 < o1 ? o1 : o0) - pc->off_lo;
  pc->nchunks = cdiv(Cin, kConvCK);
  int G = pc->nchunks * pc->ktaps * 2;
  int mt32 = cdiv(pc->M, 128) * 4;
  int64_t total = (int64_t)mt32 * G * 256;
  // + one zero tile: the tap-specialised kernel prefetches group G (one past the end) blindly
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->wpk, (total + (int64_t)G * 256) * sizeof(float)));
  WETTS_HIP_CHECK(hipMemsetAsync(pc->wpk + total, 0, (size_t)G * 256 * sizeof(float), stream));
  int threads = 256;
  int64_t blocks = (total + threads - 1) / threads;
  hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream,
                     w_dev, pc->wpk, pc->M, Cin, Cout, k, pc->ktaps, up > 0 ? up : 1, transposed,
                     rev_in, gate_H, G, total);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

void free_packed(PackedConv* pc) {
  if (pc->wpk) (void)hipFree(pc->wpk);
  pc->wpk = nullptr;
}

// ------------------------------------------------------------------------------------------
// the conv kernel
// ------------------------------------------------------------------------------------------
// EPI: 0 = generic epilogue (runtime activation / masks / late residual), 1 = plain
// (acc + bias [+ per-utterance bias]), 2 = plain followed by the MRF mean division, 3 = polyphase
// ConvTranspose1d store, 4 = plain times the output mask (the flow's pre / post convs), 5 = the WaveNet gate.  MRF = name tag of the ResBlock launches (no code difference): rocprofv3
// --stats then separates bench.py's dominant-kernel class from the flow / encoder convs that share
// the tile shape.
// Four 32x32 accumulators per wave (64 AGPRs) plus ~105 VGPRs sat one allocation granule above the
// three-waves-per-SIMD budget (168 registers): asking for three waves makes the compiler fit, and
// the extra resident wave hides the staging / A-fragment waits of the chunked loop.
// (The ablation / experiment switches this kernel carried in round 1 are gone from the product; their
// measurements are kept under profiles/r01_conv_ablation*.txt.)
// FAST = 16-byte staging for plain convs (no mask, no channel flip, whole 16-channel chunks, rows that
// start on 16-byte boundaries and hold a multiple of 4 samples): the f32 matrix pipe and the vector ALU
// are ONE resource on gfx950 -- every VALU instruction of any wave on the SIMD costs the MFMA stream
// ~4 cycles (profiles/r02_mfma_valu_coissue.txt) -- and the general staging path spends ~10 VALU per
// element on addresses, bounds and selects.  FAST: buffer addressing (uniform descriptor + uniform row
// offset + one lane offset + immediates: no vector address arithmetic), aligned 16-byte loads with one
// in-range predicate per piece, leaky-relu as mul + max, ds_write_b128.  The staged window starts at
// the 16-byte boundary at or below its first column; `sh` shifts the B-operand columns to match.
// LDS row stride of the FAST kernels with tiles of at most 256 columns: a compile-time constant (room for the widest
// window the host admits, span <= 125, plus the 16-byte alignment slack), so that the row part of every B-fragment
// address in the tap loop is an instruction immediate instead of vector address arithmetic -- the f32 matrix pipe
// pays ~3 cycles for every VALU instruction a wave issues (profiles/r02_mfma_valu_coissue.txt) and the loop spent 20
// of them per 32 MFMAs on row bases (12 now: the compiler still merges the reads into ds_read2_b32, whose 8-bit offsets
// cannot span rows, and adds a literal per row; ds_read_b32 with 16-bit offsets through inline asm brings it to 1 and
// measured no further gain).  Single convs at C >= 128: +2 ... 5 % (profiles/r04_conv_fixed_stride.txt).  512-column
// tiles keep the run-time stride (a fixed one would halve their blocks per CU).
constexpr int conv_fixed_stride(int nt) { return ((nt + 132 - 32 + 63) / 64) * 64 + 32; }
constexpr bool conv_has_fixed_stride(int nt) { return nt <= 256; }

template <int MB, int NB, int WM, int WN, int EPI, bool FAST>
__device__ __forceinline__ void conv_mfma_body(const ConvParams& p, int bid) {
  static_assert(WM * WN == 4, "4 waves per block");
  constexpr int CK = kConvCK;
  constexpr int MT = 32 * MB * WM;
  constexpr int NT = 32 * NB * WN;
  constexpr int MAXCI = (NT + 128 + 63) / 64;  // span <= 128 enforced on the host
  constexpr int RPW = CK / 4;                  // staged rows per wave

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // block -> (b, mtile, ntile); ntile fastest so neighbouring blocks share halos in L2
  const int ntiles = (p.N + NT - 1) / NT;
  const int mtiles = (p.M + MT - 1) / MT;
  const int ntile = bid % ntiles;
  bid /= ntiles;
  const int mtile = bid % mtiles;
  const int b = bid / mtiles;

  const int n0 = ntile * NT;
  // this utterance's own extent (ragged batches), else the dense geometry
  int Tin = p.Tin, Tout = p.Tout, N = p.N;
  if (p.lens) {
    Tin = min((int)p.lens[b] * p.len_mul, p.Tin);
    Tout = p.up > 0 ? Tin * p.up : Tin;
    N = p.up > 0 ? Tin + p.ktaps - 1 : Tout;
    if (n0 >= N) return;  // uniform per block
  }
  constexpr bool FIXW = FAST && conv_has_fixed_stride(NT);
  const int W = FIXW ? conv_fixed_stride(NT) : FAST ? ((NT + p.span + 3 + 3) & ~3) : NT + p.span;  // LDS row stride
  float* buf0 = smem;
  float* buf1 = smem + CK * W;

  const float* xb = p.x + (int64_t)(FAST ? __builtin_amdgcn_readfirstlane(b) : b) * p.x_bs;
  const float* mrow = p.in_mask ? p.in_mask + (int64_t)b * p.in_mask_stride : nullptr;

  // ---- FAST staging state -----------------------------------------------------------------------
  constexpr int Q4 = FAST ? (NT + 125 + 3 + 255) / 256 : 1;  // 16-byte pieces per lane per row
  const int tstart = (n0 + p.off_lo) & ~3;                   // first staged time (may be < 0)
  const int sh = FAST ? ((n0 + p.off_lo) & 3) : 0;
  const int W4 = (NT + p.span + sh + 3) >> 2;                 // pieces needed per row (<= W / 4)
  const int t0c = tstart < 0 ? 0 : tstart;                    // the row offset stays non-negative
  const int vbase = lane * 16 + (tstart - t0c) * 4;           // lane byte offset from (row, t0c)
  bool okq[Q4];
#pragma unroll
  for (int q = 0; q < Q4; ++q) {
    const int piece = lane + 64 * q, tp = tstart + 4 * piece;
    okq[q] = piece < W4 && tp >= 0 && tp + 4 <= Tin;  // rows hold a multiple of 4 samples
  }
  const bool wrq_last = lane + 64 * (Q4 - 1) < W4;  // LDS: only the last piece column can be partial
  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xb), 0,
      FAST ? __builtin_amdgcn_readfirstlane(((p.Cin - 1) * (int)p.x_cs + p.Tin) * 4) : 0, kBufRsrcDword3);
  f32x4v stage4[RPW][Q4];
  auto load_chunk4 = [&](int c) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const int soff = __builtin_amdgcn_readfirstlane(((c * CK + wave + 4 * r) * (int)p.x_cs + t0c) * 4);
#pragma unroll
      for (int q = 0; q < Q4; ++q) {
        stage4[r][q] = f32x4v{0.f, 0.f, 0.f, 0.f};
        if (okq[q]) stage4[r][q] = __builtin_amdgcn_raw_buffer_load_b128(rsx, vbase + q * 1024, soff, 0);
      }
    }
  };
  auto store_chunk4 = [&](float* buf) {
    const bool lr = p.in_act == IN_LRELU;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float* row = buf + (wave + 4 * r) * W + 4 * lane;
#pragma unroll
      for (int q = 0; q < Q4; ++q)
        if (q < Q4 - 1 || wrq_last) {
          f32x4v v = stage4[r][q];
          if (lr) {
            v.x = lrelu2(v.x, p.in_slope);
            v.y = lrelu2(v.y, p.in_slope);
            v.z = lrelu2(v.z, p.in_slope);
            v.w = lrelu2(v.w, p.in_slope);
          }
          *reinterpret_cast<f32x4v*>(row + 256 * q) = v;
        }
    }
  };

  // ---- general staging: per-column info is chunk independent ----------------------------------------
  int tcol[MAXCI];
  float mcol[MAXCI];
  if (!FAST) {
#pragma unroll
    for (int i = 0; i < MAXCI; ++i) {
      int col = lane + 64 * i;
      int t = n0 + p.off_lo + col;
      bool ok = (col < W) && (t >= 0) && (t < Tin);
      tcol[i] = ok ? t : -1;
      mcol[i] = (ok && mrow) ? mrow[t] : 1.f;
    }
  }

  float stage[RPW][MAXCI];
  auto load_chunk = [&](int c) {
    if (FAST) return load_chunk4(c);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      int ci = c * CK + wave + 4 * r;
      bool cok = ci < p.Cin;
      int ch = p.in_rev_base >= 0 ? (p.in_rev_base - ci) : ci;
      const float* xr = xb + (int64_t)ch * p.x_cs;
#pragma unroll
      for (int i = 0; i < MAXCI; ++i) {
        float v = 0.f;
        if (cok && tcol[i] >= 0) v = xr[tcol[i]];
        stage[r][i] = v;  // raw: no use of the value here, so the loads stay in flight
      }
    }
  };
  // activation + mask are applied on the way into LDS, i.e. AFTER the chunk's MFMA work, so the
  // HBM latency of the staging loads is hidden behind the matrix cores instead of stalling here
  auto store_chunk = [&](float* buf) {
    if (FAST) return store_chunk4(buf);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      float* row = buf + (wave + 4 * r) * W;
#pragma unroll
      for (int i = 0; i < MAXCI; ++i) {
        int col = lane + 64 * i;
        if (col < W) {
          float v = stage[r][i];
          if (p.in_act == IN_LRELU) v = v > 0.f ? v : v * p.in_slope;
          row[col] = v * mcol[i];
        }
      }
    }
  };

  f32x16 acc[MB][NB];
#pragma unroll
  for (int i = 0; i < MB; ++i)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // Residual / running-sum operands are folded into the accumulator INIT (C input of the first
  // MFMA) instead of being read in the epilogue: their HBM latency then overlaps the staging of
  // chunk 0 and the epilogue becomes store-only.  (Sum order changes by one f32 rounding.)
  const bool pre_res = (p.res != nullptr || p.accum) && p.up == 0 && p.out_act == OUT_NONE &&
                       p.out_mask == nullptr;
  // fast path: the whole tile lies inside the output and this is a plain (non-transposed) conv;
  // then every address is  uniform 64-bit base + 32-bit (row*stride + lane) offset  and there are
  // no per-element bounds checks -- the prologue/epilogue shrink from ~25 to ~4 instructions per
  // element, which matters most for the C=32/64 stages whose MFMA loop is only a few k-steps.
  const bool full_tile = (p.up == 0) && (n0 + NT <= N) && (mtile * MT + MT <= p.M);
  const int wrow0 = mtile * MT + wm * MB * 32;  // first row of this wave (uniform)
  const int wcol0 = n0 + wn * (32 * NB);        // first column of this wave (uniform)
  if (pre_res && full_tile) {
    const char* rbase = p.res ? reinterpret_cast<const char*>(
                                    p.res + (int64_t)b * p.r_bs + (int64_t)wrow0 * p.r_cs + wcol0)
                              : nullptr;
    const char* abase_o = reinterpret_cast<const char*>(
        p.out + (int64_t)b * p.o_bs + (int64_t)wrow0 * p.o_cs + wcol0);
    const unsigned rcs4 = (unsigned)p.r_cs * 4u, ocs4 = (unsigned)p.o_cs * 4u;
    const unsigned rlane = (unsigned)(4 * (lane >> 5)) * rcs4 + (unsigned)(lane & 31) * 4u;
    const unsigned olane = (unsigned)(4 * (lane >> 5)) * ocs4 + (unsigned)(lane & 31) * 4u;
    // Two phases behind block-level branches.  (Round 5: as one loop -- `v = res; if (accum) v += out` per element -- the
    // compiler put a uniform branch, the two loads, a vmcnt(0) and the add behind each other for EVERY element: 64
    // serialised memory round trips per wave before its first MFMA.  The launches that have both operands -- the last
    // c2 of each ResBlock chain, 8 ms of the headline step -- ran with the matrix pipe 60-83 % busy for it,
    // profiles/r05_sq_counters_mrf.txt.)  Same values, same additions: bit-identical.
    if (rbase) {  // the residual: every load in flight at once, first use is the first MFMA
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned rr = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2));
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j][r] = *reinterpret_cast<const float*>(rbase + (rr * rcs4 + rlane + 128u * j));
        }
    }
    if (p.accum) {  // + the running sum: two column blocks' worth of temporaries at a time, one wait, then the adds
#pragma unroll
      for (int jh = 0; jh < NB; jh += 2) {
        float pv[MB][16][2];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned rr = (unsigned)(i * 32 + (r & 3) + 8 * (r >> 2));
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (jh + u < NB)
                pv[i][r][u] = *reinterpret_cast<const float*>(abase_o + (rr * ocs4 + olane + 128u * (jh + u)));
          }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int u = 0; u < 2; ++u)
              if (jh + u < NB) acc[i][jh + u][r] += pv[i][r][u];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else if (pre_res) {
    const int64_t ob0 = (int64_t)b * p.o_bs;
    const int64_t rb0 = (int64_t)b * p.r_bs;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int mrow0 = mtile * MT + (wm * MB + i) * 32 + 4 * (lane >> 5);
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int col = n0 + wn * (32 * NB) + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mrow0 + (r & 3) + 8 * (r >> 2);
          float v = 0.f;
          if (col < N && row < p.M) {
            if (p.res) v = p.res[rb0 + (int64_t)row * p.r_cs + col];
            if (p.accum) v += p.out[ob0 + (int64_t)row * p.o_cs + col];
          }
          acc[i][j][r] = v;
        }
      }
    }
  }

  // packed A stream for this wave's m-blocks
  const int G = p.nchunks * p.ktaps * 2;  // groups of 4 k-steps
  const float4* abase[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    int mt32 = mtile * (MB * WM) + wm * MB + i;
    abase[i] = reinterpret_cast<const float4*>(p.wpk) + ((int64_t)mt32 * G) * 64 + lane;
  }

  // A fragments ping-pong between two register sets: a_nxt holds the even groups, a_alt the odd ones
  float4 a_nxt[MB], a_alt[MB];
#pragma unroll
  for (int i = 0; i < MB; ++i) a_nxt[i] = abase[i][0];
#pragma unroll
  for (int i = 0; i < MB; ++i) a_alt[i] = a_nxt[i];

  load_chunk(0);
  store_chunk(buf0);
  __syncthreads();

  const int half = lane >> 5;
  const int bcol0 = wn * (32 * NB) + (lane & 31) - p.pad - p.off_lo + sh;
  int g = 0;
  for (int c = 0; c < p.nchunks; ++c) {
    const float* cur = (c & 1) ? buf1 : buf0;
    const bool more = (c + 1) < p.nchunks;
    if (more) load_chunk(c + 1);
    for (int tap = 0; tap < p.ktaps; ++tap) {
      const int coff = bcol0 + tap * p.dil;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        float4 a_cur[MB];
        ++g;
        // the prefetch is unconditional (group G exists: next m-block / zero tail): a conditional load
        // makes the compiler's waitcnt insertion assume it may not be pending and emit vmcnt(0)
        if (hp == 0) {
#pragma unroll
          for (int i = 0; i < MB; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
          for (int i = 0; i < MB; ++i) a_alt[i] = abase[i][(int64_t)g * 64];
        } else {
#pragma unroll
          for (int i = 0; i < MB; ++i) a_cur[i] = a_alt[i];
#pragma unroll
          for (int i = 0; i < MB; ++i) a_nxt[i] = abase[i][(int64_t)g * 64];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch a whole group ahead of its use
        const float* brow0 = cur + (hp * 8 + half) * W + coff;
        float bv[2][NB];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
          for (int j = 0; j < NB; ++j) bv[s & 1][j] = brow0[s * 2 * W + 32 * j];
#pragma unroll
          for (int i = 0; i < MB; ++i) {
            const float av = s == 0 ? a_cur[i].x : s == 1 ? a_cur[i].y : s == 2 ? a_cur[i].z
                                                                                 : a_cur[i].w;
#pragma unroll
            for (int j = 0; j < NB; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[s & 1][j], acc[i][j], 0, 0, 0);
          }
        }
      }
    }
    if (more) store_chunk((c & 1) ? buf0 : buf1);
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------
  const int64_t ob = (int64_t)b * p.o_bs;
  const int64_t rb = (int64_t)b * p.r_bs;
  const float* bb = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_stride : nullptr;
  const float* omask = p.out_mask ? p.out_mask + (int64_t)b * p.out_mask_stride : nullptr;
  if (EPI == 5) {
    // WaveNet gate (OUT_GATE, common.h): packed rows (row, row + 1) = (tanh row, sigmoid row) of output row row / 2 sit
    // in accumulator registers (r, r + 1) of one lane.  Its own EPI value: this code in the shared generic tail below
    // cost every instantiation ~40 registers (the MRF kernels spilled).
    const int gH = p.M >> 1;
#pragma unroll
    for (int i = 0; i < MB; ++i) {
      const int mrow0 = mtile * MT + (wm * MB + i) * 32 + 4 * half;
      // every bias value of this wave's 8 row pairs first (rows clamped: the loads are unconditional), then the stores:
      // loaded inside the store loop each row pair was its own load / wait / use round trip at the end of every tile
      float bt[8], bs[8], ut[8], us[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = 2 * q;
        const int gi = min(mrow0 + (r & 3) + 8 * (r >> 2), p.M - 2) >> 1;
        bt[q] = p.bias ? p.bias[gi] : 0.f;
        bs[q] = p.bias ? p.bias[gH + gi] : 0.f;
        ut[q] = bb ? bb[gi] : 0.f;
        us[q] = bb ? bb[gH + gi] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int row = mrow0 + (r & 3) + 8 * (r >> 2);
        if (row >= p.M) continue;
        const int gi = row >> 1, q = r >> 1;
        // the sums of the launch this one replaces, to the bit: whole tiles of the plain epilogue add acc + (bias +
        // bias_b), edge tiles (the generic tail) (acc + bias) + bias_b
        float b0 = bt[q], b1 = bs[q], u0 = ut[q], u1 = us[q];
        if (full_tile) { b0 += u0; b1 += u1; u0 = u1 = 0.f; }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int col = n0 + wn * (32 * NB) + j * 32 + (lane & 31);
          if (col < N)
            p.out[ob + (int64_t)gi * p.o_cs + col] = wn_gate((acc[i][j][r] + b0) + u0, (acc[i][j][r + 1] + b1) + u1);
        }
      }
    }
    return;
  }
  if (EPI != 0 && full_tile) {
    char* obase = reinterpret_cast<char*>(p.out + ob + (int64_t)wrow0 * p.o_cs + wcol0);
    const unsigned ocs4 = (unsigned)p.o_cs * 4u;
    const unsigned olane = (unsigned)(4 * half) * ocs4 + (unsigned)(lane & 31) * 4u;
    // all bias loads first (back to back), then the store stream
    float bia[MB][16];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) bia[i][r] = 0.f;
    if (p.bias) {
      const float* bp = p.bias + wrow0 + 4 * half;
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bia[i][r] = bp[i * 32 + (r & 3) + 8 * (r >> 2)];
    }
    if (bb) {
      const float* bp = bb + wrow0 + 4 * half;
#pragma unroll
      for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bia[i][r] += bp[i * 32 + (r & 3) + 8 * (r >> 2)];
    }
    float om[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) om[j] = EPI == 4 ? omask[wcol0 + 32 * j + (lane & 31)] : 1.f;
    auto store_all = [&](auto fin) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rloc = i * 32 + (r & 3) + 8 * (r >> 2);
          const unsigned roff = (unsigned)rloc * ocs4 + olane;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            float v = fin(acc[i][j][r] + bia[i][r]);
            if (EPI == 4) v *= om[j];
            *reinterpret_cast<float*>(obase + (roff + 128u * j)) = v;
          }
        }
      }
    };
    if (EPI == 2) {  // the MRF mean (common.h: mrf_div): uniform choice of the quotient form, straight-line code each
      const float dv = p.out_div, dinv = 1.f / p.out_div;
      if (mrf_div_fast(dv)) store_all([=](float v) { return div_small_const(v, dv, dinv); });
      else store_all([=](float v) { return v / dv; });
    } else {
      store_all([](float v) { return v; });
    }
    return;
  }
  if (EPI == 3) {
    // polyphase ConvTranspose1d store (stride a power of two): row = (co, phase),
    // t = col*up + phase - pad; shifts instead of divisions, 32-bit byte offsets
    const int sh = p.up_shift, um = p.up - 1;
    char* obase = reinterpret_cast<char*>(p.out + ob);
    const unsigned ocs4 = (unsigned)p.o_cs * 4u;
    float bia[MB][16];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        bia[i][r] = (p.bias && row < p.M) ? p.bias[row >> sh] : 0.f;
      }
    // Stride >= 4 with the padding and the output length multiples of 4 (k = 16 / stride 8: every first and second stage):
    // a lane's four consecutive accumulator rows are four consecutive PHASES of one output channel, i.e. four consecutive
    // samples t0 .. t0 + 3 with t0 a multiple of 4 -- one aligned 16-byte store instead of four scalar ones, valid or
    // invalid as a whole (round 6: 650 -> 633 us and 596 -> 530 us for the two k = 16 stages of the headline).
    if (sh >= 2 && (p.up_pad & 3) == 0 && (Tout & 3) == 0 && (p.M & 3) == 0) {
#pragma unroll
      for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = wrow0 + i * 32 + 8 * q + 4 * half;  // rows row .. row + 3: phases ph .. ph + 3 of channel co
          const int co = row >> sh, ph = row & um;
          const unsigned rowoff = (unsigned)co * ocs4;
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const int col = wcol0 + 32 * j + (lane & 31);
            const int t = (col << sh) + ph - p.up_pad;
            if (row < p.M && col < N && t >= 0 && t < Tout) {
              f32x4v v;
              v.x = acc[i][j][4 * q] + bia[i][4 * q];
              v.y = acc[i][j][4 * q + 1] + bia[i][4 * q + 1];
              v.z = acc[i][j][4 * q + 2] + bia[i][4 * q + 2];
              v.w = acc[i][j][4 * q + 3] + bia[i][4 * q + 3];
              *reinterpret_cast<f32x4v*>(obase + (rowoff + (unsigned)t * 4u)) = v;
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wrow0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int co = row >> sh, ph = row & um;
        const unsigned rowoff = (unsigned)co * ocs4;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int col = wcol0 + 32 * j + (lane & 31);
          const int t = (col << sh) + ph - p.up_pad;
          if (row < p.M && col < N && t >= 0 && t < Tout)
            *reinterpret_cast<float*>(obase + (rowoff + (unsigned)t * 4u)) = acc[i][j][r] + bia[i][r];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < MB; ++i) {
    const int mrow0 = mtile * MT + (wm * MB + i) * 32 + 4 * half;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int col = n0 + wn * (32 * NB) + j * 32 + (lane & 31);
      if (col >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mrow0 + (r & 3) + 8 * (r >> 2);
        if (row >= p.M) continue;
        int co = row, t = col;
        if (p.up > 0) {
          co = row / p.up;
          t = col * p.up + (row - co * p.up) - p.up_pad;
          if (t < 0 || t >= Tout) continue;
        }
        float v = acc[i][j][r];
        if (p.bias) v += p.bias[co];
        if (bb) v += bb[co];
        if (p.wn_skip) {  // WaveNet residual / skip update instead of a store (common.h)
          const int64_t hb = (int64_t)b * p.wn_H * p.Tout + t;  // (dense row geometry)
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
        if (!pre_res) {
          if (p.res) v += p.res[rb + (int64_t)co * p.r_cs + t];
          if (p.accum) v += *dst;
        }
        if (p.out_div != 1.f) v = mrf_div(v, p.out_div, 1.f / p.out_div, mrf_div_fast(p.out_div));
        *dst = v;
      }
    }
  }
}

template <int MB, int NB, int WM, int WN, int EPI = 0, bool MRF = false, bool FAST = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MB * NB >= 4 && WN < 4) ? 3 : 1)))
void conv_mfma_kernel(const ConvParams p) {
  conv_mfma_body<MB, NB, WM, WN, EPI, FAST>(p, blockIdx.x);
}

// Several INDEPENDENT convs of one shape class in one launch (the c1 -- or the c2 -- convs of the k = 3 / 7 / 11
// ResBlocks of a stage: same input length, same channel count, different weights, taps and dilation).  A launch of
// the C = 256 stage is 1728 blocks on 1024 block slots -- 1.7 rounds, the second one 69 % full -- and the next
// launch cannot start before it has drained; three of them in one grid (longest taps first) leave one such tail
// per group instead of one per conv.  Each block runs conv_mfma_body on its own member: same code, same results.
constexpr int kConvGroupMax = 3;
struct ConvGroupParams {
  ConvParams p[kConvGroupMax];
  int first[kConvGroupMax + 1];  // first block of member j; first[n] = grid size
  int n;
};
template <int MB, int NB, int WM, int WN, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((MB * NB >= 4 && WN < 4) ? 3 : 1)))
void conv_mfma_group_kernel(const ConvGroupParams gp) {
  const int bid = blockIdx.x;
  int j = 0;
  if (gp.n > 1 && bid >= gp.first[1]) j = 1;
  if (gp.n > 2 && bid >= gp.first[2]) j = 2;
  conv_mfma_body<MB, NB, WM, WN, EPI, true>(gp.p[j], bid - gp.first[j]);
}



static bool conv_fast_ok(const ConvParams& p);

static int g_conv_variant = -1;  // -1: read WETTS_CONV_VARIANT once; 0 single-role, 1 wave-specialised

int conv_variant() {
  if (g_conv_variant < 0) {
    const char* e = getenv("WETTS_CONV_VARIANT");
    g_conv_variant = e ? atoi(e) : 0;
  }
  return g_conv_variant;
}
void set_conv_variant(int v) { g_conv_variant = v; }

template <int MB, int NB, int WM, int WN>
static int32_t launch_cfg(const ConvParams& p, hipStream_t stream) {
  constexpr int MT = 32 * MB * WM, NT = 32 * NB * WN;
  int ntiles = cdiv(p.N, NT), mtiles = cdiv(p.M, MT);
  int64_t blocks = (int64_t)ntiles * mtiles * p.B;
  if (blocks <= 0) return WETTS_OK;
  WETTS_REQUIRE(blocks < (1ll << 31), "conv grid too large");
  // FAST staging (see the kernel): plain input, 16-byte aligned rows holding a multiple of 4 samples
  const bool fast = conv_fast_ok(p);
  const size_t lds = (size_t)2 * kConvCK *
                     (fast ? (conv_has_fixed_stride(NT) ? conv_fixed_stride(NT) : ((NT + p.span + 3 + 3) & ~3)) : NT + p.span) *
                     sizeof(float);
  const dim3 grid((unsigned)blocks), blk(256);
  // epilogue specialisation: residual / running sum are folded into the accumulator init
  // whenever there is no output activation or mask, which leaves "acc + bias [/ div]"
  const bool plain = p.up == 0 && p.out_act == OUT_NONE && p.out_mask == nullptr && p.wn_skip == nullptr;
  if (p.out_act == OUT_GATE) {  // (prepare_conv has checked that nothing else is asked of the epilogue)
    if (fast)
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 5, false, true>), grid, blk, lds, stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 5, false>), grid, blk, lds, stream, p);
  } else if (p.up > 0 && p.up_shift >= 0 && p.out_act == OUT_NONE && !p.out_mask && !p.res && !p.accum &&
      !p.bias_b && p.out_div == 1.f) {
    if (fast)
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 3, false, true>), grid, blk, lds, stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 3, false>), grid, blk, lds, stream, p);
  } else if (p.up == 0 && p.out_act == OUT_NONE && p.out_mask && !p.res && !p.accum && p.out_div == 1.f) {
    if (fast)
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 4, false, true>), grid, blk, lds, stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 4, false>), grid, blk, lds, stream, p);
  } else if (plain && p.tag && fast) {  // MRF ResBlock launches: own kernel symbol, FAST staging
    if (p.out_div == 1.f)
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 1, true, true>), grid, blk, lds, stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 2, true, true>), grid, blk, lds, stream, p);
  } else if (plain && p.tag) {
    if (p.out_div == 1.f)
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 1, true>), grid, blk, lds, stream, p);
    else
      hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 2, true>), grid, blk, lds, stream, p);
  } else if (plain && p.out_div == 1.f && fast) {  // (the flow's in_layers, FFN convs on 16-byte rows)
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 1, false, true>), grid, blk, lds, stream, p);
  } else if (plain && p.out_div == 1.f) {
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 1, false>), grid, blk, lds, stream, p);
  } else if (plain) {
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 2, false>), grid, blk, lds, stream, p);
  } else if (fast) {  // generic epilogue (activation + mask: FFN conv_1) on 16-byte staging
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 0, false, true>), grid, blk, lds, stream, p);
  } else {
    hipLaunchKernelGGL((conv_mfma_kernel<MB, NB, WM, WN, 0, false>), grid, blk, lds, stream, p);
  }
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// fills the geometry fields of `p` from the packed descriptor (shared by launch_conv and launch_conv_group)
static int32_t prepare_conv(const PackedConv& pc, ConvParams& p) {
  p.wpk = pc.wpk;
  p.bias = pc.bias;
  p.M = pc.M;
  p.Cin = pc.Cin;
  p.ktaps = pc.ktaps;
  p.dil = pc.dil;
  p.pad = pc.pad;
  p.off_lo = pc.off_lo;
  p.span = pc.span;
  p.nchunks = pc.nchunks;
  p.up = pc.up;
  p.up_pad = pc.up_pad;
  p.up_shift = -1;
  for (int q = 0; q < 16; ++q)
    if (pc.up == (1 << q)) p.up_shift = q;
  WETTS_REQUIRE(pc.wpk != nullptr, "conv weight not packed");
  WETTS_REQUIRE((p.out_act == OUT_GATE) == (pc.gate_H > 0), "gate epilogue and interleaved packing go together");
  WETTS_REQUIRE(p.out_act != OUT_GATE || (!p.res && !p.accum && !p.out_mask && p.out_div == 1.f && !p.wn_skip && pc.up == 0),
                "gate epilogue: plain store only");
  // a 1x1 conv whose output is multiplied by the SAME 0/1 mask does not need it on the input: columns with
  // mask 0 come out 0 either way, the others are untouched (and the plain input can use FAST staging)
  if (pc.ktaps == 1 && pc.up == 0 && p.in_mask && p.in_mask == p.out_mask &&
      p.in_mask_stride == p.out_mask_stride && p.out_act == OUT_NONE)
    p.in_mask = nullptr;
  WETTS_REQUIRE(p.span <= 128, "conv receptive field too wide (span %d > 128)", p.span);
  if (p.up > 0) {
    p.N = p.Tin + p.ktaps - 1;
  } else {
    p.N = p.Tout;
  }
  return WETTS_OK;
}

static bool conv_fast_ok(const ConvParams& p) {
  return p.in_mask == nullptr && p.in_rev_base < 0 && (p.Cin % kConvCK) == 0 && (p.x_cs & 3) == 0 &&
         (p.x_bs & 3) == 0 && (p.Tin & 3) == 0 && p.span <= 125 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0 &&
         ((int64_t)p.Cin * p.x_cs) < (1ll << 29) && (p.lens == nullptr || (p.len_mul & 3) == 0);
}

template <int MB, int NB, int WM, int WN>
static int32_t launch_group_cfg(ConvGroupParams& gp, int max_span, hipStream_t stream) {
  constexpr int MT = 32 * MB * WM, NT = 32 * NB * WN;
  int64_t total = 0;
  for (int j = 0; j < gp.n; ++j) {
    gp.first[j] = (int)total;
    total += (int64_t)cdiv(gp.p[j].N, NT) * cdiv(gp.p[j].M, MT) * gp.p[j].B;
  }
  gp.first[gp.n] = (int)total;
  WETTS_REQUIRE(total > 0 && total < (1ll << 31), "conv group grid size");
  const size_t lds = (size_t)2 * kConvCK * (conv_has_fixed_stride(NT) ? conv_fixed_stride(NT) : ((NT + max_span + 3 + 3) & ~3)) *
                     sizeof(float);
  hipLaunchKernelGGL((conv_mfma_group_kernel<MB, NB, WM, WN, 1>), dim3((unsigned)total), dim3(256), lds, stream, gp);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// The members must be plain MRF convs of one shape class (same M, N, B, Cin; FAST staging; residual / running sum
// in the accumulator init; no mean division): the c1 or non-final c2 convs of a stage's ResBlocks.  Anything else
// -- and a group of one -- runs as separate launches, which is always equivalent.
int32_t launch_conv_group(const PackedConv* const* pcs, const ConvParams* ps, int n, hipStream_t stream,
                          int* launches) {
  if (launches) *launches = n;
  if (n <= 0) return WETTS_OK;
  ConvGroupParams gp;
  memset(&gp, 0, sizeof(gp));
  bool ok = n >= 2 && n <= kConvGroupMax && conv_variant() == 0;
  int order[kConvGroupMax] = {0, 1, 2};
  if (ok) {  // longest reduction first: the short members fill the tail of the long ones
    for (int a = 0; a < n; ++a)
      for (int b = a + 1; b < n; ++b)
        if (pcs[order[b]]->ktaps > pcs[order[a]]->ktaps) { int t = order[a]; order[a] = order[b]; order[b] = t; }
  }
  int max_span = 0;
  for (int j = 0; j < n && ok; ++j) {
    gp.p[j] = ps[order[j]];
    WETTS_TRY(prepare_conv(*pcs[order[j]], gp.p[j]));
    const ConvParams& q = gp.p[j];
    ok = q.up == 0 && q.out_act == OUT_NONE && q.out_mask == nullptr && q.out_div == 1.f && q.tag && conv_fast_ok(q) &&
         q.M == gp.p[0].M && q.N == gp.p[0].N && q.B == gp.p[0].B && q.Cin == gp.p[0].Cin && q.lens == gp.p[0].lens;
    if (q.span > max_span) max_span = q.span;
  }
  gp.n = n;
  if (ok) {
    const ConvParams& q = gp.p[0];
    const int64_t cols = (int64_t)q.N * q.B;
    auto nblk = [&](int mt, int nt) { return (int64_t)cdiv(q.M, mt) * cdiv(q.N, nt) * q.B; };
    // the tile choice of launch_conv for these shapes (big tiles only: small grids are not what this is for)
    if (q.M >= 128 && cols >= 4096 && nblk(128, 128) >= 384) {
      if (launches) *launches = 1;
      return launch_group_cfg<1, 4, 4, 1>(gp, max_span, stream);
    }
    if (q.M > 32 && q.M < 128 && cols >= 8192 && nblk(64, 256) >= 384) {
      if (launches) *launches = 1;
      return launch_group_cfg<1, 4, 2, 2>(gp, max_span, stream);
    }
  }
  for (int j = 0; j < n; ++j) WETTS_TRY(launch_conv(*pcs[j], ps[j], stream));
  return WETTS_OK;
}

int32_t launch_conv(const PackedConv& pc, ConvParams p, hipStream_t stream) {
  WETTS_TRY(prepare_conv(pc, p));
  // launches of at most ~one small tile per CU are latency-, not throughput-bound: own schedule
  if (conv_variant() == 0 && p.lens == nullptr) {
    bool taken = false;
    WETTS_TRY(launch_conv_small(p, stream, &taken));
    if (taken) return WETTS_OK;
  }
  // 1x1 convs with a plain input: the LDS-DMA GEMM with strip scheduling (gemm_pw.hip)
  {
    const int v = conv_variant();
    if ((v == 0 || (v >= 7 && v <= 9)) && pw_gemm_eligible(pc, p)) return launch_pw_gemm(pc, p, stream, v);
  }
  // tile selection: fill the chip first, then maximise per-wave register reuse
  const int64_t cols = (int64_t)p.N * p.B;
  // 1x4 wave tiles (one A fragment feeds four B fragments) measured 2-3 % faster than 2x2 at
  // every MRF shape (profiles/r01_conv_tileshape_ab.txt): half the A-stream loads per MFMA.
  switch (conv_variant()) {  // microbenchmark override (tools/bench_conv.py); 0 in production
    case 2: return launch_cfg<1, 2, 4, 1>(p, stream);
    case 3: return launch_cfg<1, 4, 2, 2>(p, stream);
    case 4: return launch_cfg<1, 2, 2, 2>(p, stream);
    case 5: return launch_cfg<1, 1, 2, 2>(p, stream);
    case 6: return launch_cfg<1, 4, 4, 1>(p, stream);
    default: break;
  }
  // ... but a grid of fewer than 1.5 big tiles per CU leaves CUs idle or unevenly loaded: the flow /
  // encoder convs (192 -> 384 over B*Ty = 13 k columns: 336 big tiles) run 28-50 % faster on
  // 64x64 tiles (profiles/r01_conv_tile_fill.txt); the MRF shapes have >= 1600 big tiles.
  auto nblk = [&](int mt, int nt) { return (int64_t)cdiv(p.M, mt) * cdiv(p.N, nt) * p.B; };
  constexpr int64_t kFill = 384;
  if (p.M >= 128 && cols >= 4096) {
    // Grids of many rounds (the MRF shapes: >= 1500 blocks of 128 x 128): the big tile.  Mid-size grids -- the flow's
    // in_layers, the encoders' FFN convs, conv_pre: a few hundred big tiles on 256 CUs -- by a cost model fitted to
    // tools/bench_conv.py at B = 16 and B = 64 (profiles/r04_inlayer_tiles.txt, r05_ffn_tiles_b64.txt): rounds of blocks
    // per CU x tile area / relative tile efficiency (64 x 64: 1, 64 x 128: 1.07, 128 x 128: 1.18).  It reproduces the
    // measured order in every case -- B = 16 in_layers: 64 x 64 (96 vs 77 / 71 TF/s); B = 64 FFN 192 -> 768 over 128
    // columns per item: 64 x 128 (101 vs 84 on the big tile, 95 on 64 x 64); B = 64 in_layers: 64 x 128 / 128 x 128 (125 /
    // 124 vs 117) -- where the old rule (big tile from 384 blocks up) left 17-21 % on the table at B = 64.
    const int64_t n128 = nblk(128, 128);
    const int kCUs = device_cus();  // (256 on MI355X; read from the device, so another part's count is modelled too)
    if (n128 >= 4 * kCUs) return launch_cfg<1, 4, 4, 1>(p, stream);
    // (a grid of fewer blocks than CUs leaves CUs idle and one wave per SIMD: never a candidate while another shape fills the
    // chip -- the FFN's 768 -> 192 at B = 64 is 192 tiles of 64 x 128: 67 TF/s against 70 on 384 tiles of 64 x 64)
    auto cost = [&](int64_t blocks, int area, double eff) {
      return (blocks < kCUs ? 4.0 : 1.0) * (double)((blocks + kCUs - 1) / kCUs) * area / eff;
    };
    const double c128 = cost(n128, 128 * 128, 1.18), c64w = cost(nblk(64, 128), 64 * 128, 1.07),
                 c64 = cost(nblk(64, 64), 64 * 64, 1.0);
    if (c128 <= c64w && c128 <= c64) return launch_cfg<1, 4, 4, 1>(p, stream);
    if (c64w <= c64) return launch_cfg<1, 2, 2, 2>(p, stream);
    return launch_cfg<1, 1, 2, 2>(p, stream);
  }
  if (p.M > 32 && cols >= 8192) {
    if (nblk(64, 256) >= kFill) return launch_cfg<1, 4, 2, 2>(p, stream);
    return launch_cfg<1, 1, 2, 2>(p, stream);
  }
  if (p.M <= 32 && cols >= 8192 && p.ktaps >= 5) return launch_cfg<1, 4, 1, 4>(p, stream);
  if (p.M <= 32 && cols >= 8192) return launch_cfg<1, 2, 1, 4>(p, stream);
  return launch_cfg<1, 1, 2, 2>(p, stream);
}

}  // namespace wetts
