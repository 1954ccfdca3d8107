// Internal shared declarations for libwetts_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include <string.h>

#include "../../include/wetts_hip.h"

namespace wetts {

void set_error(const char* fmt, ...);

#define WETTS_HIP_CHECK(expr)                                                        \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      ::wetts::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),      \
                         __FILE__, __LINE__);                                        \
      return WETTS_E_HIP;                                                            \
    }                                                                                \
  } while (0)

// Opt-in for more than 64 KB of dynamic LDS, once per (kernel, device): `state` is the caller's function-local
// `static signed char state[64]` for ONE kernel instantiation (0 = not tried, 1 = granted, -1 = refused).  Returns
// whether the kernel may be launched with a large LDS size on the current device; a refusal is remembered and
// leaves no sticky error behind, so callers with a smaller-footprint path can fall back to it.
static inline bool lds_opt_in(const void* kernel, signed char* state) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  if (state[dev] == 0) {
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) (void)hipGetLastError();
    state[dev] = e == hipSuccess ? 1 : -1;
  }
  return state[dev] > 0;
}

#define WETTS_LAUNCH_CHECK()                                                         \
  do {                                                                               \
    hipError_t _e = hipGetLastError();                                               \
    if (_e != hipSuccess) {                                                          \
      ::wetts::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),  \
                         __FILE__, __LINE__);                                        \
      return WETTS_E_HIP;                                                            \
    }                                                                                \
  } while (0)

#define WETTS_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      ::wetts::set_error(__VA_ARGS__);  \
      return WETTS_E_INVALID;           \
    }                                   \
  } while (0)

#define WETTS_TRY(expr)          \
  do {                           \
    int32_t _r = (expr);         \
    if (_r != WETTS_OK) return _r; \
  } while (0)

// ---------------------------------------------------------------------------------------
// Implicit-GEMM Conv1d / ConvTranspose1d on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
// GEMM view (per batch item):  D[M x N] = A[M x K] * B[K x N]
//   rows  M : output channels (Conv1d) or (co, phase) pairs (ConvTranspose1d, polyphase)
//   cols  N : output time (Conv1d) or input time q (ConvTranspose1d)
//   K       : (ci, tap);  B[(ci,tap)][n] = act(x[ci][n + tap*dil - pad]) (0 outside [0,Tin))
// ---------------------------------------------------------------------------------------
constexpr int kConvCK = 16;  // input channels staged per LDS chunk (fixed by the packed layout)

enum InAct { IN_NONE = 0, IN_LRELU = 1 };
// GELU = exact erf form (F.gelu default).  GATE = the WaveNet gate (commons.py:98-105) in the epilogue of the in_layer
// conv: the weights are packed with rows interleaved (PackedConv.gate_H: packed row 2i = tanh row i, 2i + 1 = sigmoid
// row H + i, which the 32x32 MFMA leaves in neighbouring accumulator registers of one lane) and the launch writes
// tanh(a_i) * sigmoid(a_{H+i}) to the H-row tensor `out`; bias / bias_b stay in the reference's row order
enum OutAct { OUT_NONE = 0, OUT_RELU = 1, OUT_GELU = 2, OUT_GATE = 3 };
__device__ __forceinline__ float wn_gate(float ta, float sa) { return tanhf(ta) * (1.f / (1.f + expf(-sa))); }

// v / d for d in {2, 3}: q = RN(v * c), r = fma(-d, q, v) (exact), q + r * c -- the correctly rounded quotient for EVERY
// finite float (exhaustive check of all 2^32 bit patterns on the host; the one difference is the sign of a zero
// result), in 3 VALU operations where the IEEE division sequence (v_div_scale / fmas / fixup) is ~10.  c = RN(1 / d).
__device__ __forceinline__ float div_small_const(float v, float d, float c) {
  const float q = v * c;
  const float r = __builtin_fmaf(-d, q, v);
  return __builtin_fmaf(r, c, q);
}

// the MRF mean  xs / num_kernels  (decoders.py:77) in every decoder epilogue: the 3-operation form for the divisors the
// recipes use (uniform branch), the IEEE division otherwise.  EVERY path (single convs, fused pairs, chains, stage kernel,
// 16 bit) goes through this one function, so fused and unfused launches stay bit-identical to each other.
__device__ __forceinline__ bool mrf_div_fast(float d) { return d == 3.f || d == 2.f; }
__device__ __forceinline__ float mrf_div(float v, float d, float dinv, bool fast) {
  return fast ? div_small_const(v, d, dinv) : v / d;
}

struct ConvParams {
  // input
  const float* x;
  int64_t x_bs, x_cs;  // batch / channel strides (floats); time stride is 1
  int Cin, Tin;
  int vec_in;            // set by the launcher: 16-byte aligned input rows (float4 staging)
  int in_rev_base;       // >=0: physical channel = in_rev_base - ci (Flip folded into indexing)
  int in_act;            // InAct
  float in_slope;
  const float* in_mask;  // [B][>=Tin] multiply after activation, or null
  int64_t in_mask_stride;
  // weights (packed by pack_conv_weight) + biases
  const float* wpk;
  const float* bias;     // [Cout] or null
  const float* bias_b;   // [B][bias_b_stride] per-utterance bias (speaker conditioning) or null
  int64_t bias_b_stride;
  // GEMM geometry
  int M, N;
  int ktaps, dil, pad;   // x index = n + tap*dil - pad
  int off_lo, span;      // min tap offset, max-min tap offset (host precomputed)
  int nchunks;           // ceil(Cin / kConvCK)
  // output
  float* out;
  int64_t o_bs, o_cs;
  int Tout;
  int up_shift;          // log2(up) when up is a power of two, else -1
  int up, up_pad;        // up>0: ConvTranspose store  out[row/up][n*up + row%up - up_pad]
  int out_act;           // OutAct
  const float* out_mask; // [B][>=Tout] or null
  int64_t out_mask_stride;
  const float* res;      // residual, indexed like out, or null
  int64_t r_bs, r_cs;
  int accum;             // 1: add the previous contents of out
  float out_div;         // final true division (MRF mean: xs / num_kernels), 1 = none
  int B;
  int tag;               // 1: MRF ResBlock launch (separate kernel symbol for profiles)
  // 1: the caller guarantees that the input rows Cin .. ceil16(Cin) - 1 of every batch item exist and hold finite
  // values (they meet the zero tail of the packed weights): lets a reduction that is not a multiple of 16 channels
  // take the LDS-DMA GEMM (gemm_pw.hip), whose stages are whole 16-channel chunks
  int k_rows_padded;
  // ragged batch (wetts_hifigan_ragged): utterance b's input holds lens[b] * len_mul time steps -- it is
  // convolved as if it were alone (zero padding at ITS end) -- while Tin / Tout / N stay the dense row
  // geometry of the tensors; blocks whose tile starts behind an utterance's end exit at once.  null = dense.
  const int64_t* lens;
  int len_mul;
  // WaveNet residual / skip update in the epilogue of the res_skip conv (modules.py:79-86; wn_update_kernel's
  // arithmetic): rows < wn_H -> h = (h + v) * mask, rows >= wn_H -> skip (+)= v; last layer: every row is skip.
  // h, skip: contiguous [B][wn_H][Tout].  `out` is not written.  null wn_skip = off.
  float* wn_h;
  float* wn_skip;
  const float* wn_mask;
  int64_t wn_mask_stride;
  int wn_H, wn_last, wn_first;
};

// default-initialised ConvParams for a plain contiguous [B,C,T] -> [B,Cout,T] conv
inline ConvParams conv_io(const float* x, int Cin, int T, float* out, int Cout, int B) {
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.x_bs = (int64_t)Cin * T;
  p.x_cs = T;
  p.Tin = T;
  p.in_rev_base = -1;
  p.in_act = IN_NONE;
  p.in_slope = 0.f;
  p.out = out;
  p.o_bs = (int64_t)Cout * T;
  p.o_cs = T;
  p.Tout = T;
  p.out_act = OUT_NONE;
  p.out_div = 1.f;
  p.B = B;
  return p;
}

// Packed weight descriptor held by the model.
struct PackedConv {
  float* wpk = nullptr;   // device, [ceil(M/128)*4][G][64][4]
  const float* bias = nullptr;
  int M = 0, Cin = 0, ktaps = 0, dil = 1, pad = 0;
  int up = 0, up_pad = 0;  // transposed conv
  int Cout = 0;            // logical output channels (M/up for transposed)
  int k_orig = 0;
  int off_lo = 0, span = 0, nchunks = 0;
  int gate_H = 0;          // > 0: M = 2 gate_H rows packed interleaved for OUT_GATE launches (only those)
};

// Packs a natural-layout weight into MFMA fragment order on the device.
//  transposed == 0: w is Conv1d [Cout][Cin][k]
//  transposed == 1: w is ConvTranspose1d [Cin][Cout][k] with stride `up`
// rev_in: input channels in reverse order (the flow's Flip folded into the weights of `pre`)
// gate_H: rows interleaved for the gate epilogue (see OutAct)
int32_t pack_conv_weight(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k,
                         int dil, int pad, int transposed, int up, hipStream_t stream,
                         PackedConv* out, int rev_in = 0, int gate_H = 0);
void free_packed(PackedConv* pc);

// Launches the conv.  Fills geometry fields of `p` from `pc`; caller fills the I/O fields.
int32_t launch_conv(const PackedConv& pc, ConvParams p, hipStream_t stream);
// Pointwise (1x1, plain input) convs as an LDS-DMA GEMM with strip scheduling (gemm_pw.hip); launch_conv routes
// eligible launches there.  variant: 0 = production choice, 7 / 8 / 9 = forced stage shapes (micro-benchmarks)
bool pw_gemm_eligible(const PackedConv& pc, const ConvParams& p);
int32_t launch_pw_gemm(const PackedConv& pc, const ConvParams& p, hipStream_t stream, int variant);
// n independent convs of one shape class in ONE launch where that is possible (conv_mfma.hip), else n launches
// (*launches = how many kernels went out)
int32_t launch_conv_group(const PackedConv* const* pcs, const ConvParams* ps, int n, hipStream_t stream,
                          int* launches = nullptr);

int conv_variant();
void set_conv_variant(int v);
// conv_small.hip: the schedule for launches of at most ~one 64x64 tile per CU (B = 1 latency)
int32_t launch_conv_small(const ConvParams& p, hipStream_t stream, bool* taken);
struct SmallConvScope {  // RAII: launches of up to `max_tiles` 64x64 tiles use it on this thread (0: never)
  explicit SmallConvScope(int max_tiles);
  ~SmallConvScope();
  int prev;
};

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// compute units of the current device (read once; 256 on MI355X): the host-side tile / strip schedules model rounds of
// blocks per CU with it
inline int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus;
}

}  // namespace wetts
