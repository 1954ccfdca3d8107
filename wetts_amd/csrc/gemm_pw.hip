// Pointwise (1x1) Conv1d as a GEMM on the gfx950 f32 matrix cores, both operands by LDS-DMA.
//
// What runs here: every dense 1x1 conv with a plain input (no input activation / mask / channel flip) and a
// reduction that is a whole number of 16-channel chunks -- the ConvNeXt pointwise GEMMs of the Vocos head
// (decoders.py:221-248: pw_conv1 512 -> 1536 + GELU, pw_conv2 1536 -> 512 + residual), its in / out convs
// (decoders.py:286-296), the WaveNet res_skip convs and the coupling layers' pre / post (modules.py:79-86,
// flows.py:494-513), q/k/v/o and proj of the encoders (attentions.py:225-233, encoders.py:54).
//
// Why a second kernel next to conv_mfma_kernel: that one is built around a k-tap window (a 16-channel chunk staged
// through registers, re-used by every tap).  With one tap a chunk is 32 MFMAs per wave between barriers, the staging
// goes through VGPRs and the vector ALU -- which on gfx950 is the same resource as the f32 matrix pipe
// (profiles/r02_mfma_valu_coissue.txt) -- the weight fragments stream from L2 on the same in-order counter as the
// staging loads, and a mid-size GEMM (12 k columns x 512..1536 rows) is 400-1200 tiles on 1024 block slots: the last
// partial round runs one block per CU.  Here:
//   * B (activations) AND A (packed weights) of a stage go global -> LDS by `buffer_load_dwordx4 ... lds`: no staging
//     registers, no ds_write, no vector address arithmetic (uniform row offset in an SGPR, one lane offset computed
//     once), and the inner loop is ds_read + MFMA only -- the A fragments of four k-steps are one ds_read_b128 of
//     the wave's private slice (the packed order [mt32][group][lane][4] is exactly one 1 KiB DMA per group);
//   * one barrier per stage, the next stage's DMA in flight behind the current stage's matrix work (two buffers);
//   * strip scheduling: a block owns `upb` 32-column units of one (batch item, 128-row m-tile) strip and walks them
//     in passes of 4 / 2 / 1 units, so the host can size the grid to a whole number of blocks per CU
//     (Vocos pw_conv1: 768 blocks x 192 columns = 3 per CU instead of 1152 x 128 = 4.5);
//   * block ids are permuted so that an XCD (blockIdx % 8) owns a contiguous range of strips: a batch item's
//     activations are fetched into ONE L2.
// Columns behind the end of a row are never stored; their staged values are the row's last 16-byte piece (the lane
// offset is clamped), so nothing outside the tensor is read.
#include "common.h"

namespace wetts {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kPwRsrcDword3 = 0x00020000;  // gfx9 raw buffer: 32-bit data format, no swizzle

struct PwParams {
  const float* x;
  int64_t x_bs;
  int x_cs, K, N;
  const float* wpk;
  int64_t wpk_bytes;
  int G;  // packed groups (of 4 k-steps) per 32-row block = K / 8
  const float* bias;
  const float* bias_b;
  int64_t bias_b_stride;
  int M;
  float* out;
  int64_t o_bs;
  int o_cs;
  const float* res;
  int64_t r_bs;
  int r_cs;
  const float* out_mask;
  int64_t out_mask_stride;
  int out_act;
  int B, S, upb, U, mtiles;
  // WaveNet residual / skip update instead of a store (modules.py:79-86; ConvParams.wn_*): rows < wn_H (not on the last
  // layer) -> h = (h + v) * mask, the others -> skip (+)= v.  h, skip: contiguous [B][wn_H][N].  wn_H % 32 == 0.
  float* wn_h;
  float* wn_skip;
  const float* wn_mask;
  int64_t wn_mask_stride;
  int wn_H, wn_last, wn_first;
};

// Where a wave's 32 rows come from / go to.  Plain convs: the residual and output tensors of the launch.  WaveNet
// update: the wave's rows are either h rows (read-modify-write through the accumulator init, masked) or skip rows
// (accumulated unless this is the first layer) -- uniform per wave because wn_H is a multiple of 32.
struct PwIo {
  const float* res;  // accumulator init (null: zero), rows r_row0 + ..
  float* out;
  const float* mask;  // output mask row of this batch item (null: none)
  int r_cs, r_row0, o_cs, o_row0;
};
__device__ __forceinline__ PwIo pw_io(const PwParams& p, int b, int rowu) {
  PwIo io;
  if (p.wn_skip) {
    const bool to_h = !p.wn_last && rowu < p.wn_H;
    float* base = (to_h ? p.wn_h : p.wn_skip) + (int64_t)b * p.wn_H * p.N;
    io.out = base;
    io.res = (to_h || !p.wn_first) ? base : nullptr;
    io.mask = to_h ? p.wn_mask + (int64_t)b * p.wn_mask_stride : nullptr;
    io.r_cs = io.o_cs = p.N;
    io.r_row0 = io.o_row0 = (to_h || p.wn_last) ? rowu : rowu - p.wn_H;
    return io;
  }
  io.res = p.res ? p.res + (int64_t)b * p.r_bs : nullptr;
  io.out = p.out + (int64_t)b * p.o_bs;
  io.mask = p.out_mask ? p.out_mask + (int64_t)b * p.out_mask_stride : nullptr;
  io.r_cs = p.r_cs;
  io.o_cs = p.o_cs;
  io.r_row0 = io.o_row0 = rowu;
  return io;
}

// exact-erf GELU (F.gelu default, decoders.py:243) with erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7 --
// an order below float32 round-off of the GEMM in front of it) on the hardware exp2 / rcp: 16 VALU operations where
// libm's erff is ~40 with branches, and this epilogue is paid in matrix throughput
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.f));
  float pl = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  pl = __builtin_fmaf(pl, t, 1.421413741f);
  pl = __builtin_fmaf(pl, t, -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.254829592f);
  pl *= t;
  const float e = __builtin_amdgcn_exp2f(z * z * -1.44269504088896340736f);
  const float erf_abs = __builtin_fmaf(-pl, e, 1.f);  // erf(|x| / sqrt 2)
  const float hx = 0.5f * x;
  return __builtin_fmaf(fabsf(hx), erf_abs, hx);  // 0.5 x (1 + sign(x) erf|.|) = hx + |hx| erf|.|
}

template <int CKS, int NBMAX>
struct PwLds {
  static constexpr int kB = CKS * 32 * NBMAX;    // floats: [CKS][32 * NBP] (row stride follows the pass width)
  static constexpr int kA = 4 * (CKS / 8) * 256;  // floats: [wave][group][lane][4]
  static constexpr int kBuf = kB + kA;
  static constexpr size_t kBytes = (size_t)2 * kBuf * sizeof(float);
};

// acc + bias (+ per-utterance bias) -> activation -> output mask -> store; buffer addressing (one lane offset per
// 32-column unit + a uniform row offset), bias rows by scalar loads
template <int NBP>
__device__ __forceinline__ void pw_epilogue(const PwParams& p, f32x16 (&acc)[NBP], int b, int rowu, int colj0, int half) {
  // ---- epilogue -------------------------------------------------------------------------------------------------
  // bias: rows rowu + rr (+ 4 for the upper half-wave) are uniform addresses -> scalar loads, one select per row
  const PwIo io = pw_io(p, b, rowu);
  __amdgpu_buffer_rsrc_t rso = __builtin_amdgcn_make_buffer_rsrc(io.out, 0, 0x7FFFFFFF, kPwRsrcDword3);
  int voff[NBP];
  float om[NBP];
#pragma unroll
  for (int j = 0; j < NBP; ++j) {
    const int col = colj0 + 32 * j;
    voff[j] = (4 * half * io.o_cs + col) * 4;
    om[j] = (io.mask && col < p.N) ? io.mask[col] : 1.f;
  }
  const float* bb = p.bias_b ? p.bias_b + (int64_t)b * p.bias_b_stride : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = (r & 3) + 8 * (r >> 2);
    const int rlo = min(rowu + rr, p.M - 1), rhi = min(rowu + rr + 4, p.M - 1);  // uniform
    float b0 = 0.f, b1 = 0.f;
    if (p.bias) { b0 = p.bias[rlo]; b1 = p.bias[rhi]; }
    if (bb) { b0 += bb[rlo]; b1 += bb[rhi]; }
    const float bia = half ? b1 : b0;
    if (rowu + rr + 4 * half >= p.M) continue;
    const int soff = __builtin_amdgcn_readfirstlane((io.o_row0 + rr) * io.o_cs * 4);
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
      if (colj0 + 32 * j >= p.N) continue;
      float v = acc[j][r] + bia;
      if (p.out_act == OUT_GELU) v = gelu_erf_fast(v);
      else if (p.out_act == OUT_RELU) v = fmaxf(v, 0.f);
      if (io.mask) v *= om[j];
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rso, voff[j], soff, 0);
    }
  }
}

// accumulators: zero, or the residual (its latency overlaps the first stage's DMA).  Residual columns behind the
// row's end are clamped to its last column (those accumulators are never stored); rows: see pw_gemm_eligible.
template <int NBP>
__device__ __forceinline__ void pw_acc_init(const PwParams& p, f32x16 (&acc)[NBP], int b, int rowu, int colj0, int half) {
#pragma unroll
  for (int j = 0; j < NBP; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const PwIo io = pw_io(p, b, rowu);
  if (io.res && rowu < p.M) {  // (uniform: M is a multiple of 32 whenever there is an init, see pw_gemm_eligible)
    __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(io.res), 0, 0x7FFFFFFF, kPwRsrcDword3);
    int voff[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) voff[j] = (4 * half * io.r_cs + min(colj0 + 32 * j, p.N - 1)) * 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int soff = __builtin_amdgcn_readfirstlane((io.r_row0 + (r & 3) + 8 * (r >> 2)) * io.r_cs * 4);
#pragma unroll
      for (int j = 0; j < NBP; ++j)
        acc[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsr, voff[j], soff, 0));
    }
  }
}

// One pass: NBP 32-column units starting at column n0, the whole reduction.
template <int CKS, int NBMAX, int NBP>
__device__ __forceinline__ void pw_pass(const PwParams& p, float* smem, int b, int mtile, int n0, int lane, int wave,
                                        __amdgpu_buffer_rsrc_t rsx, __amdgpu_buffer_rsrc_t rsw) {
  using L = PwLds<CKS, NBMAX>;
  constexpr int RS = 32 * NBP;             // LDS row stride of B (floats)
  constexpr int PPR = 8 * NBP;             // 16-byte pieces per row
  constexpr int RPI = 64 / PPR;            // rows per DMA instruction
  constexpr int TI = CKS / RPI;            // B instructions per stage (whole block)
  constexpr int GPS = CKS / 8;             // A groups (DMA instructions) per wave per stage
  const int half = lane >> 5;
  const int nstages = p.K / CKS;

  // ---- DMA lane offsets (stage independent) -------------------------------------------------------------------
  // B: lane -> (row within the instruction, piece); the column is clamped to the row's last whole piece
  const int prow = lane / PPR, piece = lane % PPR;
  int colb = n0 + piece * 4;
  colb = min(colb, p.x_cs - 4);
  const int vB = (prow * p.x_cs + colb) * 4;
  const int vA = lane * 16;
  const int mt32 = mtile * 4 + wave;
  const int sA0 = __builtin_amdgcn_readfirstlane(mt32 * p.G * 1024);
  auto issue = [&](int c, float* buf) {
    // B rows of stage c: instruction i covers rows i * RPI ...; the block's TI instructions are dealt to the waves
#pragma unroll
    for (int i0 = 0; i0 < (TI + 3) / 4; ++i0) {
      const int i = i0 * 4 + wave;
      if (TI % 4 == 0 || i < TI) {
        const int soff = __builtin_amdgcn_readfirstlane((c * CKS + i * RPI) * p.x_cs * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(buf + i * 256), 16, vB,
                                                 soff, 0, 0);
      }
    }
    float* abuf = buf + L::kB + wave * (GPS * 256);
#pragma unroll
    for (int g = 0; g < GPS; ++g) {
      const int soff = __builtin_amdgcn_readfirstlane(sA0 + (c * GPS + g) * 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(abuf + g * 256), 16, vA, soff,
                                               0, 0);
    }
  };

  issue(0, smem);

  // ---- accumulators: residual folded into the init (its latency overlaps the first stage's DMA) ---------------
  // Buffer addressing throughout: ONE lane offset per 32-column unit + a uniform row offset in an SGPR, so neither
  // the 64 residual loads nor the 64 stores of a pass hold an address register each.  Residual columns behind the
  // row's end are clamped to its last column (those accumulators are never stored); rows: see pw_gemm_eligible.
  f32x16 acc[NBP];
  const int rowu = mtile * 128 + wave * 32;  // uniform: first row of this wave
  const int colj0 = n0 + (lane & 31);
  pw_acc_init<NBP>(p, acc, b, rowu, colj0, half);

  for (int c = 0; c < nstages; ++c) {
    float* cur = smem + (c & 1) * L::kBuf;
    // this wave's DMA of stage c has landed; after the barrier everybody's has, and everybody is done reading
    // the other buffer (stage c - 1), which the next DMA overwrites
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0); expcnt / lgkmcnt untouched
    __syncthreads();
    if (c + 1 < nstages) issue(c + 1, smem + ((c + 1) & 1) * L::kBuf);
    const float* Bb = cur + half * RS + (lane & 31);
    const float* Ab = cur + L::kB + wave * (GPS * 256) + lane * 4;
#pragma unroll
    for (int g = 0; g < GPS; ++g) {
      const f32x4v a = *reinterpret_cast<const f32x4v*>(Ab + g * 256);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float bv[NBP];
#pragma unroll
        for (int j = 0; j < NBP; ++j) bv[j] = Bb[(g * 8 + s * 2) * RS + 32 * j];
        const float av = s == 0 ? a.x : s == 1 ? a.y : s == 2 ? a.z : a.w;
#pragma unroll
        for (int j = 0; j < NBP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
      }
    }
  }
  __syncthreads();  // the next pass's first DMA overwrites buffer 0

  pw_epilogue<NBP>(p, acc, b, rowu, colj0, half);
}

// TAG = name tag of the launches bench.py times as its dominant class (the ConvNeXt GEMMs of the Vocos models; no code
// difference): rocprofv3 then separates them from the flow / encoder 1x1 convs that share the instantiation
template <int CKS, int NBMAX, bool TAG = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void pw_gemm_kernel(const PwParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous block order: blockIdx % 8 is the XCD a block lands on
  int bid = blockIdx.x;
  const int total = gridDim.x;
  if ((total & 7) == 0) bid = (bid & 7) * (total >> 3) + (bid >> 3);
  const int part = bid % p.S;
  bid /= p.S;
  const int mtile = bid % p.mtiles;
  const int b = __builtin_amdgcn_readfirstlane(bid / p.mtiles);
  __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(p.x + (int64_t)b * p.x_bs), 0, __builtin_amdgcn_readfirstlane(p.K * p.x_cs * 4), kPwRsrcDword3);
  __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.wpk), 0, (int)p.wpk_bytes,
                                                                 kPwRsrcDword3);
  int u = part * p.upb;
  const int u1 = min(u + p.upb, p.U);
  while (u < u1) {
    const int w = u1 - u;
    if (NBMAX >= 4 && w >= 4) {
      pw_pass<CKS, NBMAX, (NBMAX >= 4 ? 4 : NBMAX)>(p, smem, b, mtile, u * 32, lane, wave, rsx, rsw);
      u += 4;
    } else if (NBMAX >= 2 && w >= 2) {
      pw_pass<CKS, NBMAX, (NBMAX >= 2 ? 2 : NBMAX)>(p, smem, b, mtile, u * 32, lane, wave, rsx, rsw);
      u += 2;
    } else {
      pw_pass<CKS, NBMAX, 1>(p, smem, b, mtile, u * 32, lane, wave, rsx, rsw);
      u += 1;
    }
  }
}

// ---- host side --------------------------------------------------------------------------------------------------
bool pw_gemm_eligible(const PackedConv& pc, const ConvParams& p) {
  if (pc.ktaps != 1 || pc.up != 0 || pc.pad != 0) return false;
  if (p.in_act != IN_NONE || p.in_mask != nullptr || p.in_rev_base >= 0 || p.lens != nullptr) return false;
  if (p.accum || p.out_div != 1.f || p.out_act == OUT_GATE) return false;
  if (p.wn_skip && ((p.wn_H % 32) != 0 || (pc.M % 32) != 0 || p.res || p.out_act != OUT_NONE || p.out_mask ||
                    p.wn_mask_stride < p.Tout))
    return false;  // WaveNet update: whole 32-row blocks on either side of wn_H, read-modify-write without row predicates
  if (p.res && (p.out_act != OUT_NONE || p.out_mask)) return false;  // the residual is folded into the accumulator init
  if (p.res && (pc.M % 32) != 0) return false;  // ... by loads without a row predicate: whole 32-row wave blocks only
  if ((int64_t)pc.M * p.o_cs * 4 >= (1ll << 31) || (p.res && (int64_t)pc.M * p.r_cs * 4 >= (1ll << 31))) return false;
  if (((pc.Cin % 16) != 0 && !p.k_rows_padded) || pc.Cin < 32) return false;
  if ((p.x_cs & 3) != 0 || (p.x_bs & 3) != 0 || (reinterpret_cast<uintptr_t>(p.x) & 15) != 0) return false;
  if (p.Tin > p.x_cs || p.x_cs < 4 || p.Tout != p.Tin) return false;
  if ((int64_t)pc.nchunks * 16 * p.x_cs * 4 >= (1ll << 31)) return false;
  if (p.k_rows_padded && p.x_bs < (int64_t)pc.nchunks * 16 * p.x_cs) return false;
  return true;
}

// Strip schedule: S blocks per (batch item, m-tile) strip of U 32-column units.  Cost model of a grid: the busiest CU
// runs ceil(blocks / CUs) blocks side by side (up to `slots`), each block `upb` units plus a per-pass constant; grids
// of fewer than two blocks per CU leave a SIMD with one wave and nothing to hide a barrier behind, so they are priced
// at reduced efficiency.  Ties go to wider passes (fewer blocks).
static void pw_schedule(int strips, int U, int K, int slots, int* S_out, int* upb_out) {
  const int cus = device_cus();
  double best = 1e30;
  int bestS = 1;
  for (int S = 1; S <= U; ++S) {
    const int upb = cdiv(U, S);
    if (S > 1 && cdiv(U, S - 1) == upb) continue;  // same block size with fewer blocks was already seen
    const int64_t blocks = (int64_t)strips * cdiv(U, upb);
    const int64_t per_cu = cdiv(blocks, cus);
    const int npass = upb / 4 + ((upb % 4) >= 2 ? 1 : 0) + ((upb % 4) & 1);
    // units of work: K per unit-column-block, plus ~K/8-equivalent of fixed cost per pass (prologue, epilogue, ramp)
    double block_cost = (double)upb * K + npass * (0.12 * K + 96.0);
    // narrow passes re-read the A fragments for fewer MFMAs: +8 % (2 units), +20 % (1 unit) on their share
    block_cost += ((upb % 4) >= 2 ? 2.0 * K * 0.08 : 0.0) + ((upb % 4) & 1 ? 1.0 * K * 0.20 : 0.0);
    const int resident = (int)(per_cu < slots ? per_cu : slots);
    const double eff = resident >= 3 ? 1.0 : resident == 2 ? 0.85 : 0.6;
    const double cost = (double)per_cu * block_cost / eff;
    if (cost < best * 0.999) {
      best = cost;
      bestS = cdiv(U, upb);
    }
  }
  *S_out = bestS;
  *upb_out = cdiv(U, bestS);
}

template <int CKS, int NBMAX>
static int32_t launch_pw(const PwParams& p, int64_t blocks, hipStream_t stream, bool tag = false) {
  const size_t lds = PwLds<CKS, NBMAX>::kBytes;
  static signed char opt_in[64] = {};
  if (lds > 64 * 1024 && !lds_opt_in(reinterpret_cast<const void*>(&pw_gemm_kernel<CKS, NBMAX>), opt_in)) {
    set_error("pw_gemm_kernel: the device refused the %zu-byte dynamic LDS opt-in", lds);
    return WETTS_E_HIP;
  }
  if (tag && lds <= 64 * 1024)
    hipLaunchKernelGGL((pw_gemm_kernel<CKS, NBMAX, true>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  else
    hipLaunchKernelGGL((pw_gemm_kernel<CKS, NBMAX>), dim3((unsigned)blocks), dim3(256), lds, stream, p);
  WETTS_LAUNCH_CHECK();
  return WETTS_OK;
}

// variant: 0 = production choice; 7 / 8 / 9 = force <16,4> / <32,4> / <32,2> (tools/bench_conv.py)
int32_t launch_pw_gemm(const PackedConv& pc, const ConvParams& cp, hipStream_t stream, int variant) {
  PwParams p;
  memset(&p, 0, sizeof(p));
  p.x = cp.x;
  p.x_bs = cp.x_bs;
  p.x_cs = (int)cp.x_cs;
  p.K = pc.nchunks * 16;  // == Cin, or Cin rounded up over the caller's padded rows (ConvParams.k_rows_padded)
  p.N = cp.Tout;
  p.wpk = pc.wpk;
  p.G = pc.nchunks * 2;
  p.mtiles = cdiv(pc.M, 128);
  p.wpk_bytes = (int64_t)p.mtiles * 4 * p.G * 1024;
  p.bias = pc.bias;
  p.bias_b = cp.bias_b;
  p.bias_b_stride = cp.bias_b_stride;
  p.M = pc.M;
  p.out = cp.out;
  p.o_bs = cp.o_bs;
  p.o_cs = (int)cp.o_cs;
  p.res = cp.res;
  p.r_bs = cp.r_bs;
  p.r_cs = (int)cp.r_cs;
  p.out_mask = cp.out_mask;
  p.out_mask_stride = cp.out_mask_stride;
  p.out_act = cp.out_act;
  p.wn_h = cp.wn_h;
  p.wn_skip = cp.wn_skip;
  p.wn_mask = cp.wn_mask;
  p.wn_mask_stride = cp.wn_mask_stride;
  p.wn_H = cp.wn_H;
  p.wn_last = cp.wn_last;
  p.wn_first = cp.wn_first;
  p.B = cp.B;
  p.U = cdiv(p.N, 32);
  if (p.B <= 0 || p.N <= 0 || p.M <= 0) return WETTS_OK;
  WETTS_REQUIRE(p.wpk_bytes < (1ll << 31), "pw_gemm: packed weight too large");
  const int strips = p.B * p.mtiles;
  const bool deep = (p.K % 32) == 0 && variant != 7;
  pw_schedule(strips, p.U, p.K, 4, &p.S, &p.upb);
  const int64_t blocks = (int64_t)strips * p.S;
  WETTS_REQUIRE(blocks < (1ll << 31), "pw_gemm grid too large");
  if (variant == 8 && deep) return launch_pw<32, 4>(p, blocks, stream);
  if (variant == 9 && deep && p.upb <= 3) return launch_pw<32, 2>(p, blocks, stream);
  return launch_pw<16, 4>(p, blocks, stream, cp.tag != 0);
}

}  // namespace wetts
