// Plain stride-1 Conv1d of the 16-bit decoder for the stages too wide for the fused pair kernel (C >= 256:
// ResBlock1 convs of the first upsampling stage, decoders.py:157-170) -- the round-3 loop of resblock16.hip's
// MB = 2 variant as a single conv:
//   * a wave owns 64 rows x 128 columns (two m-blocks x four n-blocks): one ds_read_b128 feeds TWO MFMAs, one
//     A fragment four (the 1 x 4 wave tile of conv_bf16_kernel reads LDS once per MFMA, which is what paces it:
//     profiles/r01_conv16_ablation.txt has the loop at 53 % of the bf16 peak with everything but MFMAs and LDS
//     reads removed);
//   * B fragments are software-pipelined one k-step ahead, A fragments one (chunk, tap) group ahead, every
//     prefetch unconditional (index clamped) so that the waits carry exact counts;
//   * the staging loads of the next 64-channel chunk are issued in two halves (taps 0 and 1 of the current
//     chunk), so only half a chunk's registers are live beside the 128 accumulators;
//   * block = 2 waves = 128 rows x 128 columns: M = 256 gives two m-tiles per time tile and twice as many blocks
//     as a 256-row tile would -- this stage has few columns (8 samples per frame), so the grid is what fills
//     the chip.
// Same packed weights, same K order per accumulator as conv_bf16_kernel: results are bit-identical to it.
#include "common.h"
#include "conv_bf16.h"
#include "conv16_dev.h"

namespace wetts {

template <bool F16>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv16_mb2_kernel(const ConvBParams p) {
  constexpr int MB = 2, NB = 4, WM = 2, CKB = 64, KS = 4, SEG = 8, NTH = 128;
  constexpr int MT = 32 * MB * WM, NT = 32 * NB;  // 128 x 128
  constexpr int RS = CKB * 2 + 16;
  constexpr int MAXSPAN = 64;
  constexpr int MAXU = ((NT + MAXSPAN) * SEG + NTH - 1) / NTH;  // 16-byte pieces per thread per chunk
  constexpr int HU = (MAXU + 1) / 2;                            // ... per staging half

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_m[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;

  const int ntiles = (p.N + NT - 1) / NT;
  const int mtiles = p.M / MT;
  int bid = blockIdx.x;
  const int ntile = bid % ntiles;
  bid /= ntiles;
  const int mtile = bid % mtiles;
  const int b = bid / mtiles;
  const int n0 = ntile * NT;
  const int W = NT + p.span;

  const unsigned short* xb = p.x + (int64_t)b * p.x_bs;
  const int useg = tid % SEG;
  const bool lrelu = p.in_act == IN_LRELU;
  const float slope = p.in_slope;

  // staging unit u = tid + NTH * i -> (row = u / SEG = lrow + 16 i, piece = tid % SEG); rows outside [0, Tin)
  // are zeros.  Address = wave-uniform part (scalar registers) + ONE per-lane 32-bit offset, so the loop carries
  // no per-piece address registers.
  constexpr int RPI = NTH / SEG;  // rows covered by one unit index (16)
  const int lrow = tid / SEG;
  const unsigned lane_off = (unsigned)(lrow * p.Cin + useg * 8) * 2u;  // bytes
  const int t_lane = n0 + p.off_lo + lrow;
  auto load_half = [&](int c, int h, uint4 (&st)[HU]) {
#pragma unroll
    for (int i = 0; i < HU; ++i) {
      const int r0 = RPI * (h * HU + i);  // uniform
      const unsigned char* base = reinterpret_cast<const unsigned char*>(
          xb + (int64_t)(n0 + p.off_lo + r0) * p.Cin + c * CKB);  // uniform (may point in front of the row 0: never read there)
      const int t = t_lane + r0;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (lrow + r0 < W && t >= 0 && t < p.Tin) v = *reinterpret_cast<const uint4*>(base + lane_off);
      st[i] = v;
    }
  };
  auto store_half = [&](int which, int h, const uint4 (&st)[HU]) {  // which: LDS buffer 0 / 1
    const unsigned drow = (unsigned)which * (unsigned)(W * RS) + (unsigned)(lrow * RS + useg * 16);
#pragma unroll
    for (int i = 0; i < HU; ++i) {
      const int r0 = RPI * (h * HU + i);
      if (lrow + r0 < W) {
        uint4 v = st[i];
        if (lrelu) {
          v.x = lrelu_pk<F16>(v.x, slope); v.y = lrelu_pk<F16>(v.y, slope);
          v.z = lrelu_pk<F16>(v.z, slope); v.w = lrelu_pk<F16>(v.w, slope);
        }
        *reinterpret_cast<uint4*>(smem_m + drow + (unsigned)(r0 * RS)) = v;
      }
    }
  };

  const int co_blk = mtile * MT + wm * 32 * MB;  // first output channel of this wave
  const int wcol = lane & 31;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;

  // residual / running sum folded into the accumulator init, in conv_bf16_kernel's order (res, then out)
  if (p.res || p.accum) {
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int t = n0 + 32 * j + wcol;
        if (t < p.N) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int c = co_blk + 32 * mi + 16 * i + 8 * half;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            if (p.res) {
              const uint4 rr = *reinterpret_cast<const uint4*>(p.res + (int64_t)b * p.r_bs + (int64_t)t * p.cout + c);
              const unsigned w4[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] = lo16<F16>(w4[e]);
                v[2 * e + 1] = hi16<F16>(w4[e]);
              }
            }
            if (p.accum) {
              const uint4 oo = *reinterpret_cast<const uint4*>(p.out + (int64_t)b * p.o_bs + (int64_t)t * p.cout + c);
              const unsigned w4[4] = {oo.x, oo.y, oo.z, oo.w};
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                v[2 * e] += lo16<F16>(w4[e]);
                v[2 * e + 1] += hi16<F16>(w4[e]);
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[mi][j][8 * i + e] = v[e];
          }
        }
      }
  }

  // ---- A stream: groups g = chunk * ktaps + tap, KS k-steps each, one group ahead -----------------------
  // Every global load of the loop is UNCONDITIONAL (indices clamped), so every wait carries an exact count.  Loads
  // retire in order: the staging loads issued at the top of groups 0 and 1 of a chunk are older than the A
  // fragments requested inside those groups, so the first A wait of the following group also waits for them --
  // by then they have had that whole group's 32 MFMAs to land.
  const int G = p.nchunks * p.ktaps;
  const int64_t mstride = (int64_t)G * KS * 64;
  const uint4* abase = reinterpret_cast<const uint4*>(p.wpk) + ((int64_t)(mtile * (MT / 32) + wm * MB) * G * KS) * 64 + lane;
  uint4 ac[MB][KS];  // the A fragments of the current group; slot s is refilled for the next group right after its use
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int s = 0; s < KS; ++s) ac[mi][s] = abase[mi * mstride + (int64_t)s * 64];

  // chunk 0
  uint4 st[HU];
  load_half(0, 0, st);
  store_half(0, 0, st);
  load_half(0, 1, st);
  store_half(0, 1, st);
  __syncthreads();

  const unsigned char* bcol0 = smem_m + (size_t)(wcol - p.pad - p.off_lo) * RS + half * 16;
  uint4 bq[2][NB];
  auto b_load = [&](uint4 (&dst)[NB], const unsigned char* bb, int s) {
#pragma unroll
    for (int j = 0; j < NB; ++j) dst[j] = *reinterpret_cast<const uint4*>(bb + (size_t)(32 * j) * RS + s * 32);
  };
  auto bpos = [&](int tp, int ch) { return bcol0 + (size_t)(ch & 1) * W * RS + (size_t)(tp * p.dil) * RS; };
  int g = 0;
  // the KS k-steps of one group; B fragments one k-step ahead (`nxt` = where the next group's first ones live),
  // A fragments of group g + 1 (the last group re-reads itself) into the slots this group has finished with
  auto mma_steps = [&](const unsigned char* cur, const unsigned char* nxt) {
    const int gn = g + 1 < G ? g + 1 : g;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) b_load(bq[(s + 1) & 1], cur, s + 1);
      else b_load(bq[(s + 1) & 1], nxt, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int mi = 0; mi < MB; ++mi) acc[mi][j] = mfma16<F16>(ac[mi][s], bq[s & 1][j], acc[mi][j]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mi = 0; mi < MB; ++mi) ac[mi][s] = abase[mi * mstride + ((int64_t)gn * KS + s) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
    ++g;
  };

  b_load(bq[0], bpos(0, 0), 0);
  for (int chunk = 0; chunk < p.nchunks; ++chunk) {
    const bool more = chunk + 1 < p.nchunks;
    const int cn = more ? chunk + 1 : chunk;  // clamped: the last chunk re-loads itself and stores nothing
    const int nbuf = (chunk + 1) & 1;
    // tap 0: first half of the next chunk's input
    load_half(cn, 0, st);
    __builtin_amdgcn_sched_barrier(0);
    mma_steps(bpos(0, chunk), bpos(1, chunk));
    // tap 1: first half into LDS, second half requested (ktaps >= 3 on this path)
    if (more) store_half(nbuf, 0, st);
    load_half(cn, 1, st);
    __builtin_amdgcn_sched_barrier(0);
    mma_steps(bpos(1, chunk), bpos(2, chunk));
    for (int tap = 2; tap < p.ktaps; ++tap) {
      // the last tap re-reads its own position: the other buffer is complete only behind the barrier below
      mma_steps(bpos(tap, chunk), tap + 1 < p.ktaps ? bpos(tap + 1, chunk) : bpos(tap, chunk));
    }
    if (more) {
      store_half(nbuf, 1, st);
      __syncthreads();
      b_load(bq[0], bpos(0, chunk + 1), 0);
    }
  }

  // ---- epilogue: + bias, / div, round to 16 bit, 16-byte channel-last stores ------------------------------
  const bool dodiv = p.out_div != 1.f;
  unsigned short* ob = p.out + (int64_t)b * p.o_bs;
  float bia[MB][16];
#pragma unroll
  for (int mi = 0; mi < MB; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      bia[mi][r] = p.bias ? p.bias[co_blk + 32 * mi + 16 * (r >> 3) + 8 * half + (r & 7)] : 0.f;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int t = n0 + 32 * j + wcol;
    if (t >= p.N) continue;
#pragma unroll
    for (int mi = 0; mi < MB; ++mi)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = acc[mi][j][8 * i + e] + bia[mi][8 * i + e];
          if (dodiv) v[e] = v[e] / p.out_div;
        }
        uint4 o;
        o.x = pk2<F16>(v[0], v[1]); o.y = pk2<F16>(v[2], v[3]);
        o.z = pk2<F16>(v[4], v[5]); o.w = pk2<F16>(v[6], v[7]);
        *reinterpret_cast<uint4*>(ob + (int64_t)t * p.cout + co_blk + 32 * mi + 16 * i + 8 * half) = o;
      }
  }
}

bool conv16_mb2_supported(const PackedConvB& pc, const ConvBParams& p) {
  return pc.up == 0 && p.epi_mode == 0 && pc.CKB == 64 && pc.M >= 256 && pc.M % 128 == 0 && pc.Cin % 64 == 0 &&
         pc.ktaps >= 3 && pc.span <= 64 && p.bias_b == nullptr && pc.Cout == pc.M;
}

// `p` must already carry the geometry fields (launch_conv_bf16 fills them)
int32_t launch_conv16_mb2(const ConvBParams& p, bool f16, hipStream_t stream) {
  constexpr int NT = 128, MT = 128, RS = 64 * 2 + 16;
  const int64_t blocks = (int64_t)cdiv(p.N, NT) * (p.M / MT) * p.B;
  if (blocks <= 0) return WETTS_OK;
  WETTS_REQUIRE(blocks < (1ll << 31), "conv grid too large");
  const size_t lds = (size_t)2 * (NT + p.span) * RS;
  if (f16)
    hipLaunchKernelGGL(conv16_mb2_kernel<true>, dim3((unsigned)blocks), dim3(128), lds, stream, p);
  else
    hipLaunchKernelGGL(conv16_mb2_kernel<false>, dim3((unsigned)blocks), dim3(128), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
