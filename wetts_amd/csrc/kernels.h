// Launchers for the non-GEMM kernels of the VITS infer() path (kernels.hip, attention.hip,
// mas.hip).  All tensors float32 channel-first [B,C,T] contiguous unless noted.
#pragma once
#include "common.h"

namespace wetts {

// a3 emb lookup * sqrt(H) * mask + sequence_mask  (encoders.py:48-53, commons.py:113-117)
int32_t k_embed_mask(const int64_t* ids, const int64_t* lengths, const float* emb, int n_vocab,
                     int B, int H, int T, float* x_out, float* mask_out, int32_t* status,
                     hipStream_t s);

// a7 channel LayerNorm (normalization.py:16-19), fused with the surrounding elementwise ops:
//   v = a (+ add);  y = LN(v)*gamma+beta;  if gelu: y = gelu_erf(y);  if res: y += res;
//   if mask: y *= mask[b,t]
int32_t k_layernorm(const float* a, const float* add, const float* gamma, const float* beta,
                    const float* res, const float* mask, int gelu, int B, int C, int T, float* out,
                    hipStream_t s);

// DDSConv depthwise dilated conv on (x*mask)  (duration_predictors.py:49-50)
// a whole DDSConv (three dwconv / LN / GELU / 1x1 / LN / GELU / residual layers) in one launch (dds_fused.hip)
struct DdsFusedParams {
  const float* x;     // [B][C][T]
  float* out;         // [B][C][T], never aliases x (blocks read their neighbours' columns of x)
  const float* mask;  // [B][T]
  const float *sep_w[3], *sep_b[3], *n1g[3], *n1b[3], *n2g[3], *n2b[3];
  const float* wpk[3];   // the 1x1 convs, conv_mfma's packed layout
  const float* bias[3];
  int B, C, T, G;        // G = 2 * (C / 16): groups of eight input channels
};
bool dds_fused_supported(int mode, int C, int Cw, int nchunks, int B, int T);
int32_t k_dds_fused(DdsFusedParams p, hipStream_t s);
// Tvalid (> 0): zero padding starts at column Tvalid of a row whose stride is T (rows padded to 4 columns)
int32_t k_dwconv(const float* x, const float* mask, const float* w, const float* bias, int k,
                 int dil, int B, int C, int T, float* out, hipStream_t s, int Tvalid = 0);

// out[b,c] = bias[c] + sum_k W[c,k] * g[b,k]   (every `cond`/`cond_layer` 1x1 conv on g[B,gin,1])
int32_t k_cond_linear(const float* g, const float* W, const float* bias, int B, int Cout, int K,
                      float* out, hipStream_t s);

int32_t k_gather_rows(const int64_t* idx, const float* table, int n_rows, int B, int C, float* out,
                      int32_t* status, hipStream_t s);

// x[b,c,t] += v[b,c]
int32_t k_add_bias_b(float* x, const float* v, int B, int C, int T, hipStream_t s);
// x[b,c,t] = (x[b,c,t] + v[b,c]) * mask[b,t]   (speaker-conditioned Encoder, attentions.py:74-78)
int32_t k_add_bias_b_mask(float* x, const float* v, const float* mask, int B, int C, int T,
                          hipStream_t s);
// out = a * scale
int32_t k_scale(const float* a, float scale, int64_t n, float* out, hipStream_t s);

// ConvFlow.pre (1 -> C 1x1 conv) + DDSConv's `x + g`   (duration_predictors.py:92-93,46-47)
int32_t k_convflow_pre(const float* z, int ch0, const float* w, const float* bias, const float* g,
                       int B, int C, int T, float* out, hipStream_t s);

// Rational-quadratic spline inverse with linear tails + cat + mask  (transforms.py:47-187,
// duration_predictors.py:96-118).  z [B,2,T] updated in place: channel ch0 *= mask,
// channel ch1 = spline^-1(z[ch1]) * mask.  h [B, 3*bins-1, T].
int32_t k_spline_inverse(float* z, int ch0, int ch1, const float* h, const float* mask,
                         int num_bins, float tail_bound, float inv_sqrt_div, int B, int T,
                         int32_t* status, hipStream_t s);

// ElementwiseAffine reverse on channel ch -> logw  (duration_predictors.py:139-141,259-262)
int32_t k_affine_reverse(const float* z, int ch, const float* m, const float* logs, int param_idx,
                         const float* mask, int B, int T, float* logw, hipStream_t s);

// commons.fused_add_tanh_sigmoid_multiply (commons.py:98-105): a [B,2H,T] -> out [B,H,T]
int32_t k_gate(const float* a, int B, int H, int T, float* out, hipStream_t s);

// WN residual/skip update (modules.py:79-86)
int32_t k_wn_update(const float* rs, float* h, float* skip, const float* mask, int last, int first,
                    int B, int H, int T, hipStream_t s);

// Flip + coupling reverse (flows.py:494-513, modules.py:100-106):
//   out[c] = c < half ? xin[C-1-c] : (xin[C-1-c] - m[c-half]) * mask
int32_t k_coupling_flip(const float* xin, const float* m, const float* mask, int B, int C, int T,
                        float* out, hipStream_t s);

// conv_post: lrelu(0.01) -> Conv1d(C,1,7,pad 3, no bias) -> tanh  (decoders.py:78-80)
// lens != null: ragged batch -- utterance b holds lens[b] * len_mul samples, the rest of its output row is 0
int32_t k_conv_post_tanh(const float* x, const float* w, int k, int B, int C, int T, float* out,
                         hipStream_t s, const int64_t* lens = nullptr, int len_mul = 0);

// a10 (models.py:254-256)
int32_t k_durations_to_lengths(const float* logw, const float* mask, float length_scale, int B,
                               int T, float* w_ceil, float* cum, int64_t* y_lengths,
                               int32_t* status, hipStream_t s);

// a10-a12 (models.py:257-267; commons.py:113-136)
int32_t k_frame_index(const float* cum, const int64_t* y_lengths, int B, int Tx, int Ty,
                      int32_t* frame2phone, float* y_mask, hipStream_t s);
int32_t k_expand_prior(const float* stats, const int32_t* frame2phone, const float* eps,
                       int64_t eps_bs, int64_t eps_cs, float noise_scale, int B, int C, int Tx, int Ty, float* m_exp,
                       float* logs_exp, float* z_p, hipStream_t s);
int32_t k_attn_path(const int32_t* frame2phone, int B, int Tx, int Ty, float* attn, hipStream_t s);

// a16 (inference.py:100-110)
int32_t k_audio_to_int16(const float* audio, const int64_t* lengths, int B, int64_t L, int16_t* pcm,
                         hipStream_t s);

// ---- VocosGenerator (decoders.py:251-308) ---------------------------------------------------
// ReflectionPad1d([1,0]) of (z * y_mask)[:, :, :L]: out [B,C,L+1], out[..,0] = in[..,1]
//   Fs (>= L + 1): row stride of out, columns L + 1 .. Fs - 1 zero-filled
int32_t k_vocos_pad(const float* z, int64_t z_bs, int64_t z_cs, const float* mask,
                    int64_t mask_stride, int B, int C, int L, float* out, hipStream_t s, int Fs = 0);
// spec [B, 2*half, F] (log-magnitude rows then phase rows) -> [B, 2*half, F] real rows then
// imaginary rows of  min(exp(mag), 1e2) * (cos(phase) + i sin(phase))   (decoders.py:297-303)
//   ri_bs: batch stride of ri (rows beyond 2 * half are the caller's padding of the iSTFT GEMM's reduction)
int32_t k_vocos_spec(const float* spec, int B, int half, int F, float* ri, hipStream_t s, int64_t ri_bs = 0);
// windowed inverse-rDFT basis as a 1x1 conv weight [n_fft][2*half]: frame[n] = hann[n] * irfft(S)[n]
//   inv_scale != 0: OnnxSTFT's inverse_basis (utils/stft.py:272-290) = float32(irfft * inv_scale) * float32(hann),
//   inv_scale = hop / n_fft
int32_t k_istft_basis(int n_fft, float* w, hipStream_t s, double inv_scale = 0.0);
// ConvNeXtLayer front half (decoders.py:241-243): out = LayerNorm_C(dw_conv k3 (x)) in one pass (rows of stride T,
// zero padding from column Tvalid); false = shape not covered, run k_dwconv + k_layernorm
bool k_convnext_dwln(const float* x, const float* w, const float* wb, const float* gamma, const float* beta, int B, int C,
                     int T, int Tvalid, float* out, hipStream_t s, int32_t* rc);
// overlap-add + window-envelope normalisation + centre trim of torch.istft (center=True):
// frames [B, n_fft, F] -> audio [B, (F-1)*hop]
//   Fs: row stride of frames (>= F)
//   envelope = 0: OnnxSTFT.inverse (utils/stft.py:325-340) -- the same overlap-add and trim, no envelope division
int32_t k_istft_ola(const float* frames, int B, int n_fft, int hop, int F, float* audio,
                    hipStream_t s, int Fs = 0, int envelope = 1);
// out[r, :] = a[r, :] * scale[r]   (rows x cols), folds ConvNeXtLayer.scale into pw_conv2
int32_t k_scale_rows(const float* a, const float* scale, int rows, int cols, float* out,
                     hipStream_t s);

// VITS2 "pre_conv" flow (flows.py:145-147): x0 = Flip(x)[:, :C/2] = x[:, C-1 .. C/2] copied out
// in natural order, raw (x0) and masked (x0m = x0 * mask)
int32_t k_flip_half(const float* x, const float* mask, int B, int C, int T, float* x0, float* x0m,
                    hipStream_t s);
// VITS2 "mono_layer_*" flows, MonoTransformerFlowLayer reverse (flows.py:287-300,302-324): the input split
// (x0 = x[:, :C/2] * sc raw and masked) and the coupling out = [x0 * sc, (x1 - m) * sc * mask]
int32_t k_mono_split(const float* x, const float* mask, int B, int C, int T, float sc, float* x0, float* x0m,
                     hipStream_t s);
int32_t k_mono_coupling(const float* x, const float* m, const float* mask, int B, int C, int T, float sc, float* out,
                        hipStream_t s);
// rows re-strided: dst[r][c] = c < cols_src ? src[r][c] : 0, c < cols_dst
int32_t k_copy_rows(const float* src, int64_t src_stride, int cols_src, float* dst, int64_t dst_stride, int cols_dst,
                    int64_t rows, hipStream_t s);
// out = a + b (out may alias a)
int32_t k_add(const float* a, const float* b, int64_t n, float* out, hipStream_t s);

// a5 windowed relative-position attention (attentions.py:235-282), banded form.
//   qkv: q,k,v [B,H*dk,T];  out [B,H*dk,T];  scores workspace: B*H*T*T + B*H*dk*T (transposed v
//   of the MFMA path) + B*H*(2*window+1)*T (relative-key table, window >= 0) floats
//   q, k, v may be slices of one fused projection [B, 3*H*dk, T]: qkv_batch_stride is their
//   batch stride in floats (H*dk*T when they are separate contiguous tensors)
//   window < 0 (no relative terms; the VITS2 flow encoders) and dk <= 48: one flash-style kernel, the T*T
//   part of the workspace stays untouched (WETTS_ATTN_FLASH=0 selects the three-kernel path)
int32_t k_rel_attention(const float* q, const float* k, const float* v, int64_t qkv_batch_stride,
                        const float* mask, const float* emb_rel_k, const float* emb_rel_v,
                        int window, int B, int n_heads, int dk, int T, float* scores, float* out,
                        hipStream_t s);

// scores-workspace floats k_rel_attention needs in FRONT of its vT / rel-table regions: B*H*T*T on
// the three-kernel path, 0 on the flash path (shared by the workspace sizing in model.hip)
int64_t attn_score_elems(int window, int dk, int B, int n_heads, int T);

// standard-normal draws (Philox4x32-10 + Box-Muller); element i depends only on (seed, offset, i)
int32_t k_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, hipStream_t s);
// out = x * mask[b,t]
int32_t k_mask_rows(const float* x, const float* mask, int B, int C, int T, float* out,
                    hipStream_t s);

// a15 MAS
int32_t k_mas(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int B, int Ty,
              int Tx, int32_t* path, float* values, hipStream_t s);

}  // namespace wetts
