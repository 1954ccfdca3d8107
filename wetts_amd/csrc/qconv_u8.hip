// uint8 dynamic-quantisation variant of the decoder's Conv1d layers -- what
// `export_onnx.py --quant` produces (wetts/vits/export_onnx.py:149-157: onnxruntime
// quantize_dynamic(..., weight_type=QuantType.QUInt8)) and the reference's runtimes then execute:
//   every Conv node  y = conv(x, w) + b   becomes
//     x_q, s_x, z_x = DynamicQuantizeLinear(x)        per launch, over the WHOLE activation tensor
//     acc           = ConvInteger(x_q, w_q, z_x, z_w) int32:  sum (x_q - z_x)(w_q - z_w)
//     y             = float(acc) * (s_x * s_w) + b
//   with w_q / s_w / z_w a per-tensor asymmetric uint8 quantisation of the (weight-norm folded)
//   weight.  ConvTranspose nodes are not touched by dynamic quantisation and stay float32.
// onnxruntime 1.13.1 is fetched by the reference's CMake and absent here, so parity for this variant is
// UNPINNED: oracle/vits_oracle.py restates the published operator definitions (ONNX
// DynamicQuantizeLinear / ConvInteger, onnxruntime/python/tools/quantization/quant_utils.py
// compute_scale_zp / quantize_nparray) and the tests hold this file bit-exact to that restatement.
//
// On gfx950 the integer contraction runs on v_mfma_i32_32x32x32_i8.  Its operands are SIGNED bytes, so
// both sides are shifted by 128 (xs = x_q - 128, ws = w_q - 128) and the zero points come back as
// rank-one corrections in the epilogue:
//   sum (xs + 128 - z_x)(ws + 128 - z_w)
//     = sum xs ws + (128 - z_w) sum_K xs + (128 - z_x) sum_K ws + K (128 - z_x)(128 - z_w)
// sum_K ws is a per-row constant computed at pack time; sum_K xs is a k-tap window over the per-frame
// channel sums the quantise kernel emits.  Padded positions hold xs = z_x - 128 (a real zero).
// Four launches per conv (reset, min/max, quantise, contract): a fidelity variant, HBM-bound passes
// over f32 tensors plus an integer contraction -- not a speed path on this machine.
#include "common.h"
#include "qconv_u8.h"

namespace wetts {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// order-preserving float <-> uint encoding for atomicMin / atomicMax
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
}

__device__ __forceinline__ float q_act(float v, int act, float slope) {
  return act ? (v > 0.f ? v : v * slope) : v;
}

// DynamicQuantizeLinear parameters from the tensor's (min, max), ONNX operator spec:
//   range adjusted to include 0; scale = (max - min) / 255; zp = saturate(round_half_even(-min / scale))
// `act` / `slope`: the stats hold the range of the RAW tensor, the quantiser sees act(tensor); act is monotone
// non-decreasing, so the range of act(x) is (act(min x), act(max x)) -- the very same f32 values a pass over act(x) finds
__device__ __forceinline__ void dq_params(const QuantStats* st, float* scale, int* zp, int act = 0, float slope = 0.f) {
  float mn = fminf(q_act(ord2f(st->min_ord), act, slope), 0.f), mx = fmaxf(q_act(ord2f(st->max_ord), act, slope), 0.f);
  float s = (mx - mn) / 255.f;
  if (!(s > 0.f)) s = 1.f;  // an all-zero tensor
  float z = rintf((0.f - mn) / s);
  z = fminf(fmaxf(z, 0.f), 255.f);
  *scale = s;
  *zp = (int)z;
}

__global__ void qstats_reset_kernel(QuantStats* st, int* colsum, int64_t n_colsum) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    st->min_ord = 0xffffffffu;
    st->max_ord = 0u;
  }
  if (i < n_colsum) colsum[i] = 0;
}

// min / max of act(x) over a [B][C][T] tensor (arbitrary batch / channel strides, optional [B][T] mask)
__global__ __launch_bounds__(256) void qminmax_kernel(const float* __restrict__ x, int64_t x_bs,
                                                      int64_t x_cs, const float* __restrict__ mask,
                                                      int64_t mask_stride, int B, int C, int T, int act,
                                                      float slope, QuantStats* st) {
  const int64_t total = (int64_t)B * C * T;
  float mn = INFINITY, mx = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % T);
    const int c = (int)((i / T) % C);
    const int b = (int)(i / ((int64_t)T * C));
    float v = x[b * x_bs + c * x_cs + t];
    if (mask) v *= mask[b * mask_stride + t];
    v = q_act(v, act, slope);
    mn = fminf(mn, v);
    mx = fmaxf(mx, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) {
    atomicMin(&st->min_ord, f2ord(mn));
    atomicMax(&st->max_ord, f2ord(mx));
  }
}

// the same range over a CONTIGUOUS unmasked tensor (every decoder tensor a range pass is run on: the upsamplers' outputs):
// 16-byte loads, four in flight per thread, no index arithmetic.  (The general kernel does two 64-bit divisions per
// element: 280 us for the 400 MB tensors of the last stages, 0.9 TB/s.)  act is monotone non-decreasing, so
// min / max commute with it exactly: it is applied to the two results.
__global__ __launch_bounds__(256) void qminmax_flat_kernel(const float4* __restrict__ x, int64_t n4, int act, float slope,
                                                           QuantStats* st) {
  float mn = INFINITY, mx = -INFINITY;
  auto fold = [&](const float4& v) {
    mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
    mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
  };
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    const float4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
    fold(a); fold(b); fold(c); fold(d);
  }
  for (; i < n4; i += stride) fold(x[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if ((threadIdx.x & 63) == 0 && mn <= mx) {
    atomicMin(&st->min_ord, f2ord(q_act(mn, act, slope)));
    atomicMax(&st->max_ord, f2ord(q_act(mx, act, slope)));
  }
}

// x [B][C][T] f32 -> xs [B][T][Cp] int8 (x_q - 128, channel-last, channels padded to Cp with a real
// zero) and colsum[b][t] = sum_c xs (over the Cp channels).  A thread takes one frame x 32 channels: 32 loads in
// flight (each coalesced along t across the lanes), one 32-byte store, and the frame's channel sum without an
// atomic when the tensor has no more than 32 channels (one atomic per 32-channel group otherwise; the first
// version took 8 channels per thread and paid an atomic for every 8).
__global__ __launch_bounds__(256) void qquantize_kernel(const float* __restrict__ x, int64_t x_bs,
                                                        int64_t x_cs, const float* __restrict__ mask,
                                                        int64_t mask_stride, int B, int C, int Cp, int T,
                                                        int act, float slope, const QuantStats* st, int stats_raw,
                                                        signed char* __restrict__ xs, int* __restrict__ colsum) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (b, c32, t), t fastest
  const int C32 = Cp / 32;
  if (idx >= (int64_t)B * C32 * T) return;
  const int t = (int)(idx % T);
  const int c32 = (int)((idx / T) % C32);
  const int b = (int)(idx / ((int64_t)T * C32));
  float scale;
  int zp;
  dq_params(st, &scale, &zp, stats_raw ? act : 0, slope);
  const float mk = mask ? mask[b * mask_stride + t] : 1.f;
  const float* xp = x + b * x_bs + t;
  float raw[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int c = c32 * 32 + e;
    raw[e] = xp[(int64_t)(c < C ? c : C - 1) * x_cs];  // unconditional (clamped channel), selected below
  }
  unsigned w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  int sum = 0;
#pragma unroll
  for (int e = 0; e < 32; ++e) {
    const int c = c32 * 32 + e;
    int q = zp;  // channel padding: (q - zp) = 0
    if (c < C) {
      float v = raw[e] * mk;
      v = q_act(v, act, slope);
      float r = rintf(v / scale) + (float)zp;  // round half to even, then saturate
      r = fminf(fmaxf(r, 0.f), 255.f);
      q = (int)r;
    }
    const int sv = q - 128;
    sum += sv;
    w[e >> 2] |= (unsigned)(sv & 0xff) << (8 * (e & 3));
  }
  uint4* dst = reinterpret_cast<uint4*>(xs + ((int64_t)b * T + t) * Cp + c32 * 32);
  dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
  dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
  if (C32 == 1) colsum[(int64_t)b * T + t] = sum;
  else atomicAdd(&colsum[(int64_t)b * T + t], sum);
}

// weight statistics -> (scale, zero point) per ORT's compute_scale_zp (range includes 0)
__global__ void qweight_params_kernel(const QuantStats* st, QuantWeightParams* wp) {
  if (threadIdx.x || blockIdx.x) return;
  float mn = fminf(ord2f(st->min_ord), 0.f), mx = fmaxf(ord2f(st->max_ord), 0.f);
  float s = (mx - mn) / 255.f;
  if (!(s > 0.f)) s = 1.f;
  float z = rintf(0.f - mn / s);
  z = fminf(fmaxf(z, 0.f), 255.f);
  wp->scale = s;
  wp->zp = (int)z;
}

// w [Cout][Cin][k] f32 -> packed ws = w_q - 128 in MFMA A-fragment order
//   [mt32][tap][cg (16-channel pairs: 32 channels)][lane][16 bytes]: lane -> row = mt32*32 + (lane&31),
//   channels cg*32 + 16*(lane>>5) + e;  and rowsum[row] = sum_K ws (padded channels hold a real zero)
__global__ void qpack_weight_kernel(const float* __restrict__ w, const QuantWeightParams* wp, int Cout,
                                    int Cin, int Cp, int k, signed char* __restrict__ out,
                                    int* __restrict__ rowsum, int64_t total) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one 16-byte lane record
  if (idx >= total) return;
  const int lane = (int)(idx & 63);
  int64_t rest = idx >> 6;
  const int CG = Cp / 32;
  const int cg = (int)(rest % CG);
  rest /= CG;
  const int tap = (int)(rest % k);
  const int mt32 = (int)(rest / k);
  const int row = mt32 * 32 + (lane & 31);
  const float scale = wp->scale;
  const int zp = wp->zp;
  unsigned o[4] = {0u, 0u, 0u, 0u};
  int sum = 0;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int ci = cg * 32 + 16 * (lane >> 5) + e;
    int q = zp;
    if (row < Cout && ci < Cin) {
      float r = rintf(w[((int64_t)row * Cin + ci) * k + tap] / scale) + (float)zp;
      r = fminf(fmaxf(r, 0.f), 255.f);
      q = (int)r;
    }
    const int sv = q - 128;
    sum += sv;
    o[e >> 2] |= (unsigned)(sv & 0xff) << (8 * (e & 3));
  }
  *reinterpret_cast<uint4*>(out + idx * 16) = make_uint4(o[0], o[1], o[2], o[3]);
  if (row < Cout) atomicAdd(&rowsum[row], sum);
}

// The integer contraction + dequantising epilogue.  Block = 4 waves; wave w owns m-block
// (mtile*WM + w % mw) and NB 32-column blocks of column group w / mw.
//
// Round 6 rebuild (the first version read every B operand of every tap straight from global memory, kept a three-step
// operand ring in 60 registers -- 240 in all, two waves per SIMD -- and started each column block's residual /
// running-sum loads only when the block's turn came: a wave lived 43 us, nearly all of it waiting, and the launches of
// the C <= 128 stages moved 1.75 TB/s).  Now:
//   * the block's int8 tile -- [columns + (k - 1) dilation frames][Cp channels], a contiguous piece of the
//     channel-last image -- and the per-frame channel sums of those frames are staged in LDS once; the taps are
//     ds_read_b128 at shifted rows (row stride Cp + 16 bytes: conflict-free), the zero-point window sums LDS reads;
//   * A fragments in a ring four steps deep (a step is NB MFMAs of 32 cycles, an L2 round trip is 10+);
//   * the epilogue's f32 operands (residual, running sum) of column block j + 1 are requested before block j is
//     finished, those of block 0 before the contraction starts: the loads of a wave are in flight all the time.
//   * 166 registers -- THREE waves per SIMD: the per-row constants of the epilogue are fetched per row quarter instead of
//     sixteen of each being held across the contraction, the A ring is four deep.  The waves of this kernel wait on memory
//     for 55 % of their cycles (SQ counters), and the third resident block is worth 12 % of the uint8 step (33.6 -> 29.4
//     ms, same box; a fourth wave for the convs without a residual measured nothing more).
// Integer accumulation is exact in any order, and the f32 expression per element is unchanged: bit-identical output.
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void qconv_i8_kernel(const QConvParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_q[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NT = 32 * NB;  // columns per wave
  // the 4 waves of a block take mw m-blocks x (4 / mw) column groups: mw = 4 for M >= 128 (one column group),
  // 2 for M = 64, 1 for M <= 32 (four column groups) -- no wave of a narrow conv sits idle
  const int mw = p.M > 64 ? 4 : (p.M > 32 ? 2 : 1);
  const int cgw = 4 / mw;
  const int NTB = NT * cgw;    // columns per block
  const int ntiles = (p.T + NTB - 1) / NTB;
  const int mtiles = (p.M + 32 * mw - 1) / (32 * mw);
  int bid = blockIdx.x;
  const int ntile = bid % ntiles;
  bid /= ntiles;
  const int mtile = bid % mtiles;
  const int b = bid / mtiles;
  const int mt32 = mtile * mw + wave % mw;
  const int row0 = mt32 * 32;
  const int nb0 = ntile * NTB;
  const int n0 = nb0 + (wave / mw) * NT;
  const bool active = row0 < p.M && n0 < p.T;
  const int khalf = lane >> 5, l31 = lane & 31;
  float sx;
  int zx;
  dq_params(p.stats, &sx, &zx, p.in_act, p.in_slope);
  const int padv = (zx - 128) & 0xff;
  const unsigned padw = (unsigned)padv * 0x01010101u;
  const int CG = p.Cp / 32;
  const signed char* xb = p.xs + (int64_t)b * p.T * p.Cp;
  const int* csb = p.colsum + (int64_t)b * p.T;
  const uint4* ab = reinterpret_cast<const uint4*>(p.wpk) + ((int64_t)mt32 * p.ktaps * CG) * 64 + lane;
  const int nsteps = p.ktaps * CG;  // steps = (tap, channel group) pairs, tap-major: the packed order

  // ---- A ring: requested first, lands behind the staging --------------------------------------------------------
  constexpr int RD = 4;
  uint4 a_r[RD];
#pragma unroll
  for (int u = 0; u < RD; ++u) a_r[u] = ab[(int64_t)(u < nsteps ? u : nsteps - 1) * 64];

  // ---- the tile: frames r0 .. r0 + R - 1 of the int8 image (padding frames hold the quantised zero) -------------
  const int span = (p.ktaps - 1) * p.dil;
  const int R = NTB + span;
  const int RS = p.Cp + 16;
  const int r0 = nb0 - p.pad;
  unsigned char* xt = smem_q;
  int* cs_l = reinterpret_cast<int*>(smem_q + (((size_t)R * RS + 15) & ~(size_t)15));
  {
    const int SEG = p.Cp >> 4, total = R * SEG;
    const uint4 pad4 = make_uint4(padw, padw, padw, padw);
    for (int base = 0; base < total; base += 4 * 256) {
      uint4 v[4];
      int off[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = base + u * 256 + tid;
        const int row = i / SEG, seg = i - row * SEG;
        const int t = r0 + row;
        const bool ok = i < total && t >= 0 && t < p.T;
        off[u] = i < total ? row * RS + seg * 16 : -1;
        v[u] = *reinterpret_cast<const uint4*>(xb + (int64_t)(ok ? t : 0) * p.Cp + (ok ? seg : 0) * 16);
        if (!ok) v[u] = pad4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (off[u] >= 0) *reinterpret_cast<uint4*>(xt + off[u]) = v[u];
    }
    for (int i = tid; i < R; i += 256) {
      const int t = r0 + i;
      cs_l[i] = (t >= 0 && t < p.T) ? csb[t] : p.Cp * (zx - 128);
    }
  }

  // ---- the epilogue's f32 operands: this lane's 16 rows of column block j (zeros where absent) ------------------
  float* ob = p.out + (int64_t)b * p.o_bs;
  const float* rb = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
  const bool has_ops = rb != nullptr || p.accum;
  // The epilogue walks this lane's 16 rows in four quarters of four rows, and within a row the NB column blocks back to
  // back: a wave then touches 128 * NB contiguous bytes of a row within a few instructions (one DRAM page visit) instead
  // of coming back to the row NB times, a column block apart each.  Operands of quarter q + 1 are requested before
  // quarter q is finished, those of quarter 0 before the contraction starts.
  float rv[2][4][NB], pv[2][4][NB];
  const int lane_o = 4 * khalf * (int)p.o_cs + l31, lane_r = 4 * khalf * (int)p.r_cs + l31;
  auto load_ops = [&](int q, float (&rvq)[4][NB], float (&pvq)[4][NB]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rowu = row0 + i + 8 * q;  // uniform; this lane's row is rowu + 4 * khalf
      const bool okr = active && rowu + 4 * khalf < p.M;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const bool ok = okr && n0 + 32 * j + l31 < p.T;
        rvq[i][j] = 0.f;
        pvq[i][j] = 0.f;
        if (rb && ok) rvq[i][j] = (rb + (int64_t)rowu * p.r_cs + (n0 + 32 * j))[lane_r];
        if (p.accum && ok) pvq[i][j] = (ob + (int64_t)rowu * p.o_cs + (n0 + 32 * j))[lane_o];
      }
    }
  };
  if (has_ops) load_ops(0, rv[0], pv[0]);

  __syncthreads();
  if (!active) return;

  i32x16 acc[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0;

  // ---- the contraction: B fragments from the tile, one step ahead -----------------------------------------------
  {
    const unsigned char* bbase = xt + (size_t)((n0 - nb0) + l31) * RS + 16 * khalf;
    // ONE B set, each fragment re-requested right behind the MFMA that consumed it: it is needed again NB MFMAs later,
    // which covers the LDS latency (a second set would be 16 more registers: the kernel then spills at two waves per SIMD)
    uint4 bq[NB];
    const int tap_bytes = p.dil * RS;
    int cgn = 0, boffn = 0;  // channel group / byte offset of the step whose fragments are requested next
    auto advance = [&]() __attribute__((always_inline)) {
      if (++cgn == CG) { cgn = 0; boffn += tap_bytes - (CG - 1) * 32; }
      else boffn += 32;
    };
#pragma unroll
    for (int j = 0; j < NB; ++j) bq[j] = *reinterpret_cast<const uint4*>(bbase + (size_t)(32 * j) * RS);
    advance();
    for (int step = 0; step < nsteps; step += RD) {
#pragma unroll
      for (int u = 0; u < RD; ++u) {
        if (step + u < nsteps) {
          const i32x4 a = {(int)a_r[u].x, (int)a_r[u].y, (int)a_r[u].z, (int)a_r[u].w};
          const bool more = step + u + 1 < nsteps;
          const unsigned char* nb = bbase + (more ? boffn : 0);
#pragma unroll
          for (int j = 0; j < NB; ++j) {
            const uint4 bv = bq[j];
            const i32x4 bqv = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w};
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, bqv, acc[j], 0, 0, 0);
            bq[j] = *reinterpret_cast<const uint4*>(nb + (size_t)(32 * j) * RS);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (more) advance();
        }
        const int sn = step + u + RD;
        a_r[u] = ab[(int64_t)(sn < nsteps ? sn : nsteps - 1) * 64];  // unconditional (clamped): exact wait counts
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: zero-point corrections, dequantise, bias / residual / running sum, f32 store ------
  const float sw = p.wparams->scale;
  const int zw = p.wparams->zp;
  const int cx = 128 - zx, cw = 128 - zw;
  const int K = p.Cp * p.ktaps;
  const float sprod = sx * sw;  // Mul(x_scale, w_scale) of the quantised graph, in f32
  const float* bb = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_stride : nullptr;
  const bool dodiv = p.out_div != 1.f;
  float omn = INFINITY, omx = -INFINITY;  // range of what this wave writes (for the conv that consumes it)
  int base[NB];  // zero-point terms of this lane's NB columns: window sum of the per-frame channel sums (staged; padding
  // frames hold Cp * (zx - 128))
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int* cp = cs_l + (n0 - nb0) + 32 * j + l31;
    int cs = 0;
    for (int tap = 0; tap < p.ktaps; ++tap) cs += cp[tap * p.dil];
    base[j] = cw * cs + K * cx * cw;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (has_ops && q + 1 < 4) load_ops(q + 1, rv[(q + 1) & 1], pv[(q + 1) & 1]);
    // the quarter's per-row constants (zero-point correction, bias): loaded here, four at a time -- sixteen of each
    // held across the contraction cost the third wave per SIMD
    int rcq[4];
    float rbq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + i + 8 * q + 4 * khalf;
      const bool okr = row < p.M;
      rcq[i] = okr ? cx * p.rowsum[row] : 0;
      rbq[i] = (okr && p.bias) ? p.bias[row] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = 4 * q + i;  // accumulator register: rows (r & 3) + 8 * (r >> 2) + 4 * khalf
      const int row = row0 + i + 8 * q + 4 * khalf;
      if (row >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        if (n0 + 32 * j + l31 >= p.T) continue;
        const int a = acc[j][r] + base[j] + rcq[i];
        // (the empty asm pins the rounded product: HIP contracts a * b + c -- also through __fmul_rn / __fadd_rn --
        // into v_fma_f32, one rounding instead of the graph's two)
        float v = (float)a * sprod;
        asm volatile("" : "+v"(v));
        v += rbq[i];
        if (bb) v += bb[row];  // the graph adds cond(g) to the finished conv_pre output
        if (rb) v += rv[q & 1][i][j];
        if (p.accum) v += pv[q & 1][i][j];
        if (dodiv) v = v / p.out_div;
        (ob + (int64_t)(row - 4 * khalf) * p.o_cs + (n0 + 32 * j))[lane_o] = v;
        omn = fminf(omn, v);
        omx = fmaxf(omx, v);
      }
    }
  }
  if (p.out_partial) {
    // one (min, max) record per BLOCK, no atomics: 25 k waves hitting one slot with atomicMin / atomicMax were most
    // of this kernel's time (same-address atomics serialise at ~12 ns each); qrange_reduce_kernel folds the records
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      omn = fminf(omn, __shfl_xor(omn, off, 64));
      omx = fmaxf(omx, __shfl_xor(omx, off, 64));
    }
    // (no __syncthreads: waves of a block may have returned early.)  Every wave folds its range into the BLOCK's own
    // two words: at most four contenders per address, all on this CU
    if (lane == 0 && omn <= omx) {
      atomicMin(reinterpret_cast<unsigned*>(&p.out_partial[blockIdx.x].x), f2ord(omn));
      atomicMax(reinterpret_cast<unsigned*>(&p.out_partial[blockIdx.x].y), f2ord(omx));
    }
  }
}

// partial[i] = (min_ord, max_ord) of block i (ordered-uint encoding, reset to (0xffffffff, 0)) -> the slot
__global__ __launch_bounds__(1024) void qrange_reduce_kernel(const float2* __restrict__ partial, int n,
                                                             QuantStats* __restrict__ slot) {
  __shared__ unsigned smn[16], smx[16];
  unsigned mn = 0xffffffffu, mx = 0u;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const unsigned a = __float_as_uint(partial[i].x), b = __float_as_uint(partial[i].y);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned a = __shfl_xor(mn, off, 64), b = __shfl_xor(mx, off, 64);
    mn = a < mn ? a : mn;
    mx = b > mx ? b : mx;
  }
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      mn = smn[w] < mn ? smn[w] : mn;
      mx = smx[w] > mx ? smx[w] : mx;
    }
    slot->min_ord = mn;
    slot->max_ord = mx;
  }
}

__global__ void qpartial_reset_kernel(float2* partial, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) partial[i] = make_float2(__uint_as_float(0xffffffffu), __uint_as_float(0u));
}

// ---------------------------------------------------------------------------------------------------
int32_t pack_qconv_weight(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k, int dil,
                          int pad, hipStream_t s, PackedQConv* pc) {
  pc->Cout = Cout;
  pc->Cin = Cin;
  pc->Cp = (Cin + 31) / 32 * 32;
  pc->ktaps = k;
  pc->dil = dil;
  pc->pad = pad;
  pc->bias = bias_dev;
  const int mt32 = cdiv(Cout, 128) * 4, CG = pc->Cp / 32;
  const int64_t recs = (int64_t)mt32 * k * CG * 64;
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->wpk, (size_t)recs * 16));
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->rowsum, (size_t)mt32 * 32 * sizeof(int)));
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->wparams, sizeof(QuantWeightParams)));
  WETTS_HIP_CHECK(hipMalloc((void**)&pc->wstats, sizeof(QuantStats)));
  hipLaunchKernelGGL(qstats_reset_kernel, dim3(cdiv(mt32 * 32, 256)), dim3(256), 0, s, pc->wstats,
                     pc->rowsum, (int64_t)mt32 * 32);
  WETTS_LAUNCH_CHECK();
  const int64_t nw = (int64_t)Cout * Cin * k;
  const int gridm = (int)((nw + 255) / 256 < 2048 ? (nw + 255) / 256 : 2048);
  hipLaunchKernelGGL(qminmax_kernel, dim3(gridm), dim3(256), 0, s, w_dev, (int64_t)0, (int64_t)0,
                     (const float*)nullptr, (int64_t)0, 1, 1, (int)nw, 0, 0.f, pc->wstats);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(qweight_params_kernel, dim3(1), dim3(64), 0, s, pc->wstats, pc->wparams);
  WETTS_LAUNCH_CHECK();
  hipLaunchKernelGGL(qpack_weight_kernel, dim3((unsigned)((recs + 255) / 256)), dim3(256), 0, s, w_dev,
                     pc->wparams, Cout, Cin, pc->Cp, k, pc->wpk, pc->rowsum, recs);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

void free_packed_qconv(PackedQConv* pc) {
  if (pc->wpk) (void)hipFree(pc->wpk);
  if (pc->rowsum) (void)hipFree(pc->rowsum);
  if (pc->wparams) (void)hipFree(pc->wparams);
  if (pc->wstats) (void)hipFree(pc->wstats);
  pc->wpk = nullptr;
  pc->rowsum = nullptr;
  pc->wparams = nullptr;
  pc->wstats = nullptr;
}

static int64_t qconv_partial_bytes(int B, int T) {  // upper bound: <= 8 blocks per 128 columns (Cout <= 1024)
  return align_up((int64_t)B * (T / 128 + 1) * 8 * (int64_t)sizeof(float2), 256);
}

int64_t qconv_scratch_bytes(int B, int Cin, int T) {
  const int64_t Cp = (Cin + 31) / 32 * 32;
  return align_up((int64_t)B * T * Cp, 256) + align_up((int64_t)B * T * 4, 256) + 256 + qconv_partial_bytes(B, T);
}

__global__ void qstats_reset_slots_kernel(QuantStats* st, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    st[i].min_ord = 0xffffffffu;
    st[i].max_ord = 0u;
  }
}

int32_t k_qstats_reset(QuantStats* slots, int n, hipStream_t s) {
  if (n <= 0) return WETTS_OK;
  hipLaunchKernelGGL(qstats_reset_slots_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, slots, n);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// range of act(x * mask) over [B][C][T] into a reset slot: the flat kernel for contiguous unmasked tensors
static int32_t launch_qminmax(const float* x, int64_t x_bs, int64_t x_cs, const float* mask, int64_t mask_stride, int B,
                              int C, int T, int act, float slope, QuantStats* slot, hipStream_t s) {
  const int64_t nx = (int64_t)B * C * T;
  if (nx <= 0) return WETTS_OK;
  if (!mask && x_cs == T && x_bs == (int64_t)C * T && (nx & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
    const int64_t n4 = nx >> 2;
    const int grid = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(qminmax_flat_kernel, dim3(grid), dim3(256), 0, s, reinterpret_cast<const float4*>(x), n4, act, slope,
                       slot);
  } else {
    const int gridm = (int)((nx + 255) / 256 < 4096 ? (nx + 255) / 256 : 4096);
    hipLaunchKernelGGL(qminmax_kernel, dim3(gridm), dim3(256), 0, s, x, x_bs, x_cs, mask, mask_stride, B, C, T, act, slope,
                       slot);
  }
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

int32_t k_qminmax(const float* x, int B, int C, int T, QuantStats* slot, hipStream_t s) {
  return launch_qminmax(x, (int64_t)C * T, (int64_t)T, nullptr, 0, B, C, T, 0, 0.f, slot, s);
}

int32_t launch_qconv(const PackedQConv& pc, QConvIO io, void* scratch, int64_t scratch_bytes,
                     hipStream_t s) {
  WETTS_REQUIRE(pc.wpk != nullptr, "quantised conv weight not packed");
  const int B = io.B, T = io.T;
  if ((int64_t)B * T == 0) return WETTS_OK;
  WETTS_REQUIRE(scratch_bytes >= qconv_scratch_bytes(B, pc.Cin, T), "qconv: scratch too small");
  char* sp = static_cast<char*>(scratch);
  signed char* xs = reinterpret_cast<signed char*>(sp);
  sp += align_up((int64_t)B * T * pc.Cp, 256);
  int* colsum = reinterpret_cast<int*>(sp);
  sp += align_up((int64_t)B * T * 4, 256);
  QuantStats* own = reinterpret_cast<QuantStats*>(sp);
  sp += 256;
  float2* partial = reinterpret_cast<float2*>(sp);
  const int64_t ncs = (int64_t)B * T;
  // resets the per-frame channel sums and this launch's own range slot (unused when the producer left the range)
  hipLaunchKernelGGL(qstats_reset_kernel, dim3((unsigned)((ncs + 255) / 256)), dim3(256), 0, s, own, colsum, ncs);
  WETTS_LAUNCH_CHECK();
  const bool have_range = io.in_stats != nullptr && io.mask == nullptr;
  const QuantStats* st = have_range ? io.in_stats : own;
  if (!have_range) {  // one more pass over the tensor: range of act(x * mask)
    WETTS_TRY(launch_qminmax(io.x, io.x_bs, io.x_cs, io.mask, io.mask_stride, B, pc.Cin, T, io.in_act, io.in_slope, own, s));
  }
  const int64_t nq = (int64_t)B * (pc.Cp / 32) * T;
  hipLaunchKernelGGL(qquantize_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, s, io.x, io.x_bs,
                     io.x_cs, io.mask, io.mask_stride, B, pc.Cin, pc.Cp, T, io.in_act, io.in_slope, st,
                     have_range ? 1 : 0, xs, colsum);
  WETTS_LAUNCH_CHECK();
  QConvParams p;
  memset(&p, 0, sizeof(p));
  p.xs = xs; p.colsum = colsum; p.stats = st;
  p.wpk = pc.wpk; p.rowsum = pc.rowsum; p.wparams = pc.wparams; p.bias = pc.bias;
  p.bias_b = io.bias_b; p.bias_b_stride = io.bias_b_stride;
  p.M = pc.Cout; p.Cp = pc.Cp; p.ktaps = pc.ktaps; p.dil = pc.dil; p.pad = pc.pad; p.T = T; p.B = B;
  p.out = io.out; p.o_bs = io.o_bs; p.o_cs = io.o_cs;
  p.res = io.res; p.r_bs = io.r_bs; p.r_cs = io.r_cs;
  p.accum = io.accum; p.out_div = io.out_div;
  p.in_act = have_range ? io.in_act : 0;  // the stats of the range pass already are those of act(x)
  p.in_slope = io.in_slope;
  p.out_stats = io.out_stats;
  // (NB = 2 -- 159 registers, three waves per SIMD -- measured 3 % slower: what counts is the length of the contiguous run a
  // wave moves per row, not the number of waves)
  constexpr int NB = 4;
  const int mw = pc.Cout > 64 ? 4 : (pc.Cout > 32 ? 2 : 1);  // as in the kernel
  const int64_t blocks = (int64_t)cdiv(T, 32 * NB * (4 / mw)) * cdiv(pc.Cout, 32 * mw) * B;
  WETTS_REQUIRE(blocks < (1ll << 31), "qconv grid too large");
  p.out_partial = nullptr;
  if (io.out_stats) {
    WETTS_REQUIRE(blocks * (int64_t)sizeof(float2) <= qconv_partial_bytes(B, T), "qconv: more blocks than range records");
    p.out_partial = partial;
    hipLaunchKernelGGL(qpartial_reset_kernel, dim3((unsigned)((blocks + 255) / 256)), dim3(256), 0, s, partial, (int)blocks);
    WETTS_LAUNCH_CHECK();
  }
  const int R = 32 * NB * (4 / mw) + (pc.ktaps - 1) * pc.dil;
  const size_t lds = (((size_t)R * (pc.Cp + 16) + 15) & ~(size_t)15) + (size_t)R * sizeof(int);
  WETTS_REQUIRE(lds <= 160 * 1024, "qconv: tile of %zu bytes exceeds the LDS (Cp %d, k %d, dilation %d)", lds, pc.Cp, pc.ktaps, pc.dil);
  static signed char opt_in[64] = {};
  if (lds > 64 * 1024 && !lds_opt_in(reinterpret_cast<const void*>(&qconv_i8_kernel<NB>), opt_in)) {
    set_error("qconv_i8_kernel: the device refused the %zu-byte dynamic LDS opt-in", lds);
    return WETTS_E_HIP;
  }
  hipLaunchKernelGGL((qconv_i8_kernel<NB>), dim3((unsigned)blocks), dim3(256), lds, s, p);
  WETTS_LAUNCH_CHECK();
  if (io.out_stats) {
    hipLaunchKernelGGL(qrange_reduce_kernel, dim3(1), dim3(1024), 0, s, partial, (int)blocks, io.out_stats);
    WETTS_LAUNCH_CHECK();
  }
  return WETTS_OK;
}

__global__ void tanh_inplace_kernel(float* x, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = tanhf(x[i]);
}

int32_t k_tanh_inplace(float* x, int64_t n, hipStream_t s) {
  if (n <= 0) return WETTS_OK;
  hipLaunchKernelGGL(tanh_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, n);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

}  // namespace wetts
