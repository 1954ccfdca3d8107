// bf16 decoder path declarations (conv_bf16.hip).
#pragma once
#include "common.h"

namespace wetts {

struct PackedConvB {
  unsigned short* wpk = nullptr;  // device, [ceil(M/128)*4][G][KS][64][8] 16-bit, m-block rows permuted (pack_bf16_kernel)
  const float* bias = nullptr;    // f32
  int M = 0, Cin = 0, Cout = 0, ktaps = 0, dil = 1, pad = 0, up = 0, up_pad = 0;
  int off_lo = 0, span = 0, nchunks = 0, CKB = 64;
  int f16 = 0;  // 0: bfloat16 storage, 1: IEEE half
  int gate_h = 0;  // > 0: WN in_layer packed for the fused gate epilogue (rows paired tanh / sigmoid)
};

struct ConvBParams {
  const unsigned short* x;  // [B][Tin][Cin] bf16, channel-last
  int64_t x_bs;             // batch stride in elements
  int Cin, Tin;
  int in_act;
  float in_slope;
  const unsigned short* wpk;
  const float* bias;
  const float* bias_b;   // [B][bias_b_stride] per-utterance f32 bias (speaker conditioning) or null
  int64_t bias_b_stride;
  int M, N, ktaps, dil, pad, off_lo, span, nchunks;
  unsigned short* out;  // [B][Tout][cout]
  int64_t o_bs;
  int cout, Tout;
  int up, up_pad;
  const unsigned short* res;  // [B][Tout][cout] or null
  int64_t r_bs;
  int accum;
  float out_div;
  int B;
  // fused WaveNet epilogues of the 16-bit flow (set by the caller; conv_bf16.hip):
  //  epi_mode 1, in_layer packed with gate_h = H: out [B][T][H] = tanh(a) * sigmoid(b) of the rounded
  //    conv outputs a = channel c, b = channel H + c (commons.py:98-105)
  //  epi_mode 2, res_skip: h = (h + rs[:H]) * mask (16 bit, in place), skip (+)= rs[H:] in f32
  //    (modules.py:79-86); `out` is not written
  //  epi_mode 3, conv_post (k_conv_post_mfma16): out_f32 [B][T] = tanh(row 0)
  int epi_mode;
  float* out_f32;
  unsigned short* wn_h;   // [B][T][H]
  float* wn_skip;         // [B][T][H] f32
  const float* wn_mask;   // [B][T]
  int wn_H, wn_last, wn_first;
  int lds_rows;  // rows of one LDS staging buffer (set by the launcher: whole staging passes)
  int tag;    // 1: MRF ResBlock launch (own kernel symbol for profiles, no code difference)
  int basic;  // 1: always conv_bf16_kernel (the decoder's UNFUSED diagnostic mode: the form the newer kernels are held to)
};

// fused ResBlock1 pair (resblock16.hip): out = (x + c2(lrelu(c1(lrelu(x)))) [+ out]) / div
constexpr int RESPAIR_MAX_SPAN = 64;  // (k-1)*dilation of c1 the fused kernel is sized for
constexpr int RESPAIR2_MAX_SPAN32 = 80;  // ... of the C = 32 ResBlock2 chain
struct ResPairParams {
  const unsigned short* x;  // [B][T][C] residual stream, channel-last
  unsigned short* out;      // [B][T][C] (never aliases x)
  const unsigned short *wpk1, *wpk2;
  const float *bias1, *bias2;
  int T, B;
  int ktaps, dil;  // c1: ktaps taps at dilation dil; c2: ktaps taps at dilation 1 (ResBlock1) / dil2
  int dil2;        // ResBlock2 chain only
  int accum;       // add the previous contents of out (running MRF sum)
  float out_div;
  float slope;     // leaky-relu slope in front of both convs
  int ntiles, nblocks;
};
bool resblock_pair16_supported(const PackedConvB& c1, const PackedConvB& c2);
int32_t launch_resblock_pair16(const PackedConvB& c1, const PackedConvB& c2, ResPairParams p,
                               hipStream_t stream);
// a whole ResBlock2 (two residual convs, c2 at its own dilation) in one launch
bool resblock2_chain16_supported(const PackedConvB& c1, const PackedConvB& c2, int max_waste_pct);
int32_t launch_resblock2_chain16(const PackedConvB& c1, const PackedConvB& c2, ResPairParams p,
                                 hipStream_t stream);

// a whole MRF stage of ResBlock2 blocks (up to three chains of two residual convs) in one launch, the running
// sum kept in registers (resblock2_stage16.hip)
constexpr int RESSTAGE2_MAX_CHAINS = 3;
struct ResStage2Params {
  const unsigned short* x;  // [B][T][C] channel-last
  unsigned short* out;      // [B][T][C] (never aliases x)
  const unsigned short *wpk1[RESSTAGE2_MAX_CHAINS], *wpk2[RESSTAGE2_MAX_CHAINS];
  const float *bias1[RESSTAGE2_MAX_CHAINS], *bias2[RESSTAGE2_MAX_CHAINS];
  int ktaps[RESSTAGE2_MAX_CHAINS], dil1[RESSTAGE2_MAX_CHAINS], dil2[RESSTAGE2_MAX_CHAINS];
  int nchain;
  int origin;     // widest c2 halo of the stage: column c of every chain <-> time n0 - origin + c
  int h1max;      // widest c1 halo of the stage (shared staging: the lrelu(x) tile's halo)
  int xrows;      // rows of the lrelu(x) LDS tile (whole staging passes)
  int T, B;
  float out_div;  // applied with the last chain (x = xs / num_kernels)
  float slope;
  int ntiles, nblocks;
};
int resblock2_stage16_nto(const PackedConvB* const* c1, const PackedConvB* const* c2, int nchain, int max_waste_pct);
int32_t launch_resblock2_stage16(const PackedConvB* const* c1, const PackedConvB* const* c2, int nchain,
                                 ResStage2Params p, hipStream_t stream);

int32_t pack_conv_weight_bf16(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k,
                              int dil, int pad, int transposed, int up, int f16, hipStream_t stream,
                              PackedConvB* out, int gate_h = 0);
void free_packed_bf16(PackedConvB* pc);
int32_t launch_conv_bf16(const PackedConvB& pc, ConvBParams p, hipStream_t stream);
int resblock_pair16_ntc(int C, bool rb2 = false);
int32_t k_cf32_to_cl16(const float* x, unsigned short* out, int B, int C, int T, int f16,
                       hipStream_t s);
// ... with batch / channel strides and an optional row mask [B][>= T]: out = round16(x * mask)
int32_t k_cf32_to_cl16_strided(const float* x, int64_t x_bs, int64_t x_cs, const float* mask, int64_t mask_stride,
                               unsigned short* out, int B, int C, int T, int f16, hipStream_t s);
// ---- 16-bit WaveNet layers of the flow (wn16.hip), channel-last like the 16-bit decoder ----------
// acts[row][c] = tanh(xin[row][c]) * sigmoid(xin[row][H + c])   (commons.py:98-105), rows = B*T
int32_t k_gate_cl16(const unsigned short* xin, unsigned short* acts, int64_t rows, int H, int f16,
                    hipStream_t s);
// WN residual / skip update (modules.py:79-86) on channel-last tensors: h 16-bit, skip f32;
// mask [rows] (B*T).  rs has 2H channels (H when `last`).
int32_t k_wn_update_cl16(const unsigned short* rs, unsigned short* h, float* skip, const float* mask,
                         int last, int first, int64_t rows, int H, int f16, hipStream_t s);
// f32 channel-last [B][T][C] -> f32 channel-first [B][C][T]
int32_t k_cl32_to_cf32(const float* x, float* out, int B, int C, int T, hipStream_t s);

int32_t k_conv_post_bf16(const unsigned short* x, const float* w, int k, int B, int C, int T,
                         float* out, int f16, hipStream_t s);
int32_t k_conv_post_mfma16(const PackedConvB& pc, const unsigned short* x, int B, int C, int T,
                           float* out, hipStream_t s);

}  // namespace wetts
