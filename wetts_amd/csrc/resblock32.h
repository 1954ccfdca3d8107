// Fused float32 ResBlock1 pair (resblock32.hip): out = (x + c2(lrelu(c1(lrelu(x)))) [+ out]) / div
#pragma once
#include "common.h"

namespace wetts {

constexpr int RESPAIR32_MAX_SPAN = 64;  // (k-1)*dilation of c1 the staging loop is sized for

struct ResPair32Params {
  const float* x;  // [B][C][T] residual stream
  float* out;      // [B][C][T] (never aliases x)
  const float *wpk1, *wpk2;  // pack_conv_weight layouts of c1 / c2
  const float *bias1, *bias2;
  int T, B;
  int ktaps, dil;  // c1: ktaps taps at dilation dil; c2: ktaps taps at dilation 1 (ResBlock1) / dil2
  int dil2;        // ResBlock2 chain only: dilation of the second conv
  int accum;       // add the previous contents of out (running MRF sum)
  float out_div;
  float slope;     // leaky-relu slope in front of both convs
  int ntiles, nblocks, Wp;  // filled by the launcher
};

// max_lds_bytes: largest x tile the caller accepts (80 KB keeps two blocks per CU)
bool resblock_pair32_supported(const PackedConv& c1, const PackedConv& c2, int max_lds_bytes);
int32_t launch_resblock_pair32(const PackedConv& c1, const PackedConv& c2, ResPair32Params p,
                               hipStream_t stream);

// ResBlock2 (decoders.py:205-214) as one launch: out = (t1 + c2(lrelu(t1)) [+ out]) / div with
// t1 = x + c1(lrelu(x)); c2 keeps its own dilation.  max_waste_pct bounds the share of tile columns
// the second conv's halo discards ((k-1)*d2 of NTC).
bool resblock2_chain32_supported(const PackedConv& c1, const PackedConv& c2, int max_lds_bytes,
                                 int max_waste_pct);
int32_t launch_resblock2_chain32(const PackedConv& c1, const PackedConv& c2, ResPair32Params p,
                                 hipStream_t stream);

// ---- ResBlock1 chains (resblock_chain32.hip): npairs (c1, c2) pairs in one launch ----------------
constexpr int RESCHAIN32_MAX_PAIRS = 3;  // a whole ResBlock1 (decoders.py:157-170)
constexpr int RESCHAIN32_MAX_M = 28;     // widest half-width (k-1)/2 * dilation the tile margins hold

struct ResChain32Params {
  const float* x;  // [B][C][T] residual stream (16-byte aligned rows: T % 4 == 0)
  float* out;      // [B][C][T] (never aliases x)
  const float* wpk[2 * RESCHAIN32_MAX_PAIRS];   // pack_conv_weight layouts: c1_0, c2_0, c1_1, ...
  const float* bias[2 * RESCHAIN32_MAX_PAIRS];
  int dil[RESCHAIN32_MAX_PAIRS];                // dilation of c1_p (c2_p runs at 1)
  int npairs, ktaps;
  int T, B;
  int accum;       // add the previous contents of out (running MRF sum)
  float out_div;
  float slope;
  int S, Mmin, NTO, ntiles, nblocks, Wp;  // filled by the launcher
  // ragged batch: utterance b holds lens[b] * len_mul samples (a multiple of 4); T stays the row stride (common.h)
  const int64_t* lens;
  int len_mul;
  int bstride;  // filled by the launcher: utterance walk stride of the ragged block order
};

// c1 / c2: npairs descriptors each.  max_waste_pct bounds the share of tile columns the chain's halo
// discards (2 S of NTC); max_lds_bytes the tile (80 KB keeps two blocks per CU).
bool resblock_chain32_supported(const PackedConv* c1, const PackedConv* c2, int npairs,
                                int max_lds_bytes, int max_waste_pct);
int resblock_chain32_nto(int C, int ktaps, const int* dil, int npairs);  // valid outputs per tile
int32_t launch_resblock_chain32(const PackedConv* c1, const PackedConv* c2, int npairs,
                                ResChain32Params p, hipStream_t stream);

}  // namespace wetts
