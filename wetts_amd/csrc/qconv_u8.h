// uint8 dynamic-quantisation conv (qconv_u8.hip): the decoder variant `export_onnx.py --quant` creates.
#pragma once
#include "common.h"

namespace wetts {

struct QuantStats {        // device: running min / max of a tensor in an order-preserving encoding
  unsigned min_ord, max_ord;
};
struct QuantWeightParams { // device: per-tensor weight scale / zero point (uint8, asymmetric)
  float scale;
  int zp;
};

struct PackedQConv {
  signed char* wpk = nullptr;          // [ceil(Cout/128)*4][k][Cp/32][64][16] bytes: w_q - 128
  int* rowsum = nullptr;               // [rows] sum_K (w_q - 128)
  QuantWeightParams* wparams = nullptr;
  QuantStats* wstats = nullptr;
  const float* bias = nullptr;         // f32 or null
  int Cout = 0, Cin = 0, Cp = 0, ktaps = 0, dil = 1, pad = 0;
};

struct QConvIO {
  const float* x;          // [B][Cin][T] f32 (strides below), quantised per launch over the whole tensor
  int64_t x_bs, x_cs;
  const float* mask;       // optional [B][>=T]: x * mask before the activation (conv_pre's z * y_mask)
  int64_t mask_stride;
  int in_act;              // 1: leaky-relu with in_slope in front of the quantiser (the graph's LeakyRelu node)
  float in_slope;
  float* out;              // [B][Cout][T] f32
  int64_t o_bs, o_cs;
  const float* res;        // optional residual added to the dequantised output
  int64_t r_bs, r_cs;
  const float* bias_b;     // optional per-utterance bias [B][bias_b_stride]
  int64_t bias_b_stride;
  int accum;               // add the previous contents of out
  float out_div;
  int B, T;
  // range bookkeeping across convs (round 3): a tensor's (min, max) is what DynamicQuantizeLinear needs first, and
  // reading the tensor once more for it was 30 % of the uint8 step.  The conv that PRODUCES a tensor can record the
  // range of what it writes (out_stats: a slot the caller has reset), and the conv that consumes it takes the slot
  // (in_stats) instead of launching the range pass; the activation in front of the quantiser is monotone, so it is
  // applied to the two scalars.  Exact: the same two f32 values either way.
  const QuantStats* in_stats;
  QuantStats* out_stats;
};

struct QConvParams {       // kernel arguments (filled by launch_qconv)
  const signed char* xs;
  const int* colsum;
  const QuantStats* stats;
  const signed char* wpk;
  const int* rowsum;
  const QuantWeightParams* wparams;
  const float* bias;
  const float* bias_b;
  int64_t bias_b_stride;
  int M, Cp, ktaps, dil, pad, T, B;
  float* out;
  int64_t o_bs, o_cs;
  const float* res;
  int64_t r_bs, r_cs;
  int accum;
  float out_div;
  int in_act;              // activation the consumer applies in front of its quantiser (for `stats`)
  float in_slope;
  QuantStats* out_stats;   // or null
  float2* out_partial;     // [grid blocks] (min, max) of what each block wrote; reduced into out_stats by a tiny kernel
};

// Conv1d weights only (onnxruntime's dynamic quantisation leaves ConvTranspose in float)
int32_t pack_qconv_weight(const float* w_dev, const float* bias_dev, int Cout, int Cin, int k, int dil,
                          int pad, hipStream_t s, PackedQConv* pc);
void free_packed_qconv(PackedQConv* pc);
int64_t qconv_scratch_bytes(int B, int Cin, int T);  // int8 image + per-frame channel sums + stats
// resets n range slots (min = +inf, max = -inf in the ordered encoding)
int32_t k_qstats_reset(QuantStats* slots, int n, hipStream_t s);
// range of a plain [B][C][T] tensor into a (reset) slot -- for tensors no quantised conv produced
int32_t k_qminmax(const float* x, int B, int C, int T, QuantStats* slot, hipStream_t s);
int32_t launch_qconv(const PackedQConv& pc, QConvIO io, void* scratch, int64_t scratch_bytes,
                     hipStream_t s);
// x = tanh(x) in place (the generator's last op, decoders.py:80)
int32_t k_tanh_inplace(float* x, int64_t n, hipStream_t s);

}  // namespace wetts
