/*
 * wetts_hip.h -- C ABI of libwetts_hip.so: the MI355X (gfx950) native VITS inference path.
 *
 * This is the drop-in boundary for ONE hot path of wenet-e2e/wetts: SynthesizerTrn.infer()
 * (text encoder -> duration -> length regulate -> flow^-1 -> HiFi-GAN) plus the standalone
 * monotonic-alignment search.  The reference has no FFI seam on this path (it is nn.Modules all
 * the way down), so every entry point below cites the reference Python / C++
 * interface whose arithmetic it replaces.  Citations are relative to the reference repo root.
 *
 * Conventions
 *   - plain C: pointers + sizes only, no torch / HIP types in any signature (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - every data pointer is a DEVICE pointer owned by the caller unless the name ends in `_host`.
 *   - activations are contiguous float32, channel-first [B, C, T]; ids / lengths are int64;
 *     masks are float 0/1 [B, T] (the reference's [B,1,T] with the unit dim dropped).
 *   - all launches are stream-ordered and asynchronous; the stage calls never allocate: scratch comes
 *     from the caller's workspace (wetts_workspace_bytes()).  Device memory is allocated by wetts_create()
 *     and by the two precision setters (wetts_set_decoder_precision / wetts_set_flow_precision), which build
 *     their 16-bit / uint8 weight copies when called, on the default stream, and return after it has drained
 *     (set-up calls, like create); wetts_dynamic_quant_conv1d and wetts_set_mrf_timing are validation / measurement
 *     aids and say so at their declarations.
 *   - return value: 0 = ok, negative = error (WETTS_E_*); wetts_last_error() gives the message
 *     of the calling thread's last failure.  Kernels never fall back to a CPU path.
 *   - thread-compatible: one handle may be used from one thread at a time (the reference's
 *     VitsModel has the same contract, runtime/core/model/vits_model.h:30-66).
 */
#ifndef WETTS_HIP_H_
#define WETTS_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WETTS_ABI_VERSION 9

#define WETTS_OK 0
#define WETTS_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define WETTS_E_HIP (-2)       /* HIP runtime / launch error               */
#define WETTS_E_WORKSPACE (-3) /* caller workspace too small               */
#define WETTS_E_DOMAIN (-4)    /* reference would raise (spline domain...) */

/* Device-side error flags: conditions only the data can reveal, where the reference raises from
 * inside a module.  Kernels OR these bits into the int32 status word registered with
 * wetts_set_status_word(); the host reads the word back together with y_lengths (the one D2H of
 * infer()) and raises / returns like the reference does. */
#define WETTS_STATUS_SPLINE_DOMAIN 1      /* `assert (discriminant >= 0).all()` transforms.py:171 */
#define WETTS_STATUS_PHONE_ID_RANGE 2     /* nn.Embedding IndexError, encoders.py:48 */
#define WETTS_STATUS_SPEAKER_ID_RANGE 4   /* emb_g IndexError, models.py:239 */
#define WETTS_STATUS_DURATION_NONFINITE 8 /* NaN / inf durations reach .long(), models.py:256 */

#define WETTS_MAX_STAGES 8
#define WETTS_MAX_RB_KERNELS 8
#define WETTS_MAX_RB_DILATIONS 8

/* Hyper-parameters of one model.  Field meaning == ctor arguments of the reference
 * SynthesizerTrn (wetts/vits/model/models.py:19-51) as splatted from `hps.model`
 * (wetts/vits/inference.py:72-76). */
typedef struct wetts_config {
  int32_t n_vocab;
  int32_t inter_channels;  /* 192 */
  int32_t hidden_channels; /* 192 */
  int32_t filter_channels; /* 768 */
  int32_t n_heads;         /* 2   */
  int32_t n_layers;        /* 6   */
  int32_t kernel_size;     /* 3 (encoder FFN) */
  int32_t window_size;     /* 4 (attentions.py:20) */
  int32_t resblock;        /* 1 or 2 (decoders.py:35) */
  int32_t n_resblock_kernels;
  int32_t resblock_kernel_sizes[WETTS_MAX_RB_KERNELS];
  int32_t n_resblock_dilations; /* per resblock: 3 for "1", 2 for "2" */
  int32_t resblock_dilation_sizes[WETTS_MAX_RB_KERNELS][WETTS_MAX_RB_DILATIONS];
  int32_t n_upsamples;
  int32_t upsample_rates[WETTS_MAX_STAGES];
  int32_t upsample_kernel_sizes[WETTS_MAX_STAGES];
  int32_t upsample_initial_channel;
  int32_t n_speakers;   /* 0 => no emb_g, g = None */
  int32_t gin_channels; /* 256 */
  int32_t use_sdp;      /* 1: StochasticDurationPredictor, 0: DurationPredictor */
  int32_t flow_n_flows;     /* 4  (models.py:133-142) */
  int32_t flow_wn_layers;   /* 4  */
  int32_t flow_kernel_size; /* 5  */
  int32_t sdp_n_flows;      /* 4  (models.py:145-150) */
  int32_t dp_filter_channels; /* 256 (models.py:152-156) */
  /* vocoder: 0 = HiFi-GAN Generator (decoders.py:15-88, the fields above), 1 = VocosGenerator
   * (decoders.py:251-308: ConvNeXt stack + iSTFT head; examples/baker/configs/vocos.json:38-52) */
  int32_t vocoder_type;
  int32_t vocos_channels;    /* 512  */
  int32_t vocos_h_channels;  /* 1536 */
  int32_t vocos_num_layers;  /* 8    */
  int32_t istft_n_fft;       /* 1024 (vocos_out_channels = n_fft + 2) */
  int32_t istft_hop_length;  /* 256  */
  int32_t istft_win_length;  /* 1024 (must equal n_fft) */
  /* VITS2 flows (models.py:73-79, flows.py:340-360): 0 = ResidualCouplingLayer, 1 = "pre_conv"
   * (ResidualCouplingTransformersLayer, flows.py:95-177: 2-layer window-less Encoder on x0),
   * 2 = "pre_conv2" (ResidualCouplingTransformersLayer2, flows.py:16-92: 1-layer Encoder on
   * pre(x0) with the flow's kernel size and the default relative window),
   * 3 = "mono_layer_inter_residual", 4 = "mono_layer_post_residual" (the default when a config sets
   * use_transformer_flows without a type, models.py:74-75): [ResidualCouplingLayer, Flip,
   * MonoTransformerFlowLayer] per flow (flows.py:391-425); the mono layer (flows.py:242-324) is a
   * coupling on I/2 channels whose statistics come from the same 2-layer window-less Encoder + a 1x1 post,
   * 4 with residual_connection=True (reverse: x0 / 2, (x1 - m) / (1 + exp(-logs))) */
  int32_t transformer_flows;
  /* speaker-conditioned text encoder (models.py:87-101, attentions.py:39-48,74-78): at layer 2
   * x = (x + spk_emb_linear(g)) * x_mask */
  int32_t use_spk_conditioned_encoder;
  /* SynthesizerTrn(..., is_onnx=...) (models.py:50,111 -> VocosGenerator, decoders.py:279-283): the iSTFT arithmetic of a
   * Vocos model at create; see wetts_set_istft_mode().  export_onnx.py:59 forces it to 1 for every exported graph.
   * HiFi-GAN models ignore it, as the reference does. */
  int32_t is_onnx;
  int32_t reserved[6];
} wetts_config_t;

typedef struct wetts_model wetts_model_t; /* opaque */

/* ---- library / weight-blob layout (no GPU required) ------------------------------------- */

int32_t wetts_abi_version(void);
/* Message of the calling thread's last failed call.  The reference reports the same conditions as Python exceptions
 * raised from inside its modules (e.g. the spline's discriminant assert, transforms.py:171; an embedding index outside
 * its table, encoders.py:48 / models.py:239). */
const char* wetts_last_error(void);

/* The weight blob is one flat float32 array holding every inference tensor of the reference
 * state_dict in its natural PyTorch layout, weight-norm pairs already folded
 * (w = g * v / ||v||, norm over all dims but 0 -- torch.nn.utils.weight_norm dim=0, as used at
 * decoders.py:41-48,96-153, modules.py:35,49,58).  Names are the reference's state_dict keys
 * with `weight_g`/`weight_v` collapsed to `weight`.  The library is the single source of truth
 * for the order: the host loader enumerates it with these three calls.
 * Replaces: utils/task.py:31-56 load_checkpoint + decoders.py:84-88 remove_weight_norm. */
int32_t wetts_blob_num_tensors(const wetts_config_t* cfg);
/* name_buf gets a NUL-terminated key; shape gets up to 4 dims (unused = 0).
 * offset / numel are in floats. */
int32_t wetts_blob_tensor_info(const wetts_config_t* cfg, int32_t index, char* name_buf,
                               size_t name_buf_len, int64_t* offset, int64_t* numel,
                               int64_t shape[4]);
int64_t wetts_blob_numel(const wetts_config_t* cfg);

/* ---- model lifetime ---------------------------------------------------------------------- */

/* Builds a model on the CURRENT HIP device from a device-resident blob (e.g. the buffer an
 * RCCL broadcast just filled).  Repacks conv weights into MFMA fragment order on the device.
 * The blob may be freed after the call returns.  Replaces SynthesizerTrn.__init__ + .to(device)
 * + load_checkpoint (inference.py:72-80). */
int32_t wetts_create(const wetts_config_t* cfg, const float* blob_dev, int64_t blob_numel,
                     void* stream, wetts_model_t** out);
/* Frees the packed weights and the model (the reference: garbage collection of the nn.Module built at
 * inference.py:66-80).  Not stream-ordered: the caller synchronises the streams that used the model first. */
void wetts_destroy(wetts_model_t* m);

/* Registers the device int32 the stage calls below OR their WETTS_STATUS_* bits into (caller-owned;
 * zeroed here, stream-ordered; NULL = flags are dropped, ids are still clamped for memory safety).
 * wetts_infer() uses a word of its own workspace and maps it to its return code. */
int32_t wetts_set_status_word(const wetts_model_t* m, int32_t* status_dev, void* stream);

/* Seed of the model's own standard-normal stream (Philox4x32-10), used by wetts_infer() when
 * eps_w / eps_z are NULL -- the reference draws them with torch.randn / torch.randn_like from the global generator
 * (duration_predictors.py:257-258, models.py:267), i.e. under the caller's torch.manual_seed. */
int32_t wetts_set_seed(const wetts_model_t* m, uint64_t seed);

/* total upsampling factor (prod upsample_rates, decoders.py:30-47; the iSTFT hop for Vocos, decoders.py:283) == the
 * `hop_length` of the checkpoint's config (examples/baker/configs/v1.json:23). */
int32_t wetts_hop_length(const wetts_model_t* m);

/* Copies the model's folded float32 weights (the blob wetts_create() was given, in
 * wetts_blob_tensor_info() order) into out_dev[numel], stream-ordered.  Backs the drop-in module's
 * state_dict() (the reference saves checkpoints from it, utils/task.py:59-76). */
int32_t wetts_get_blob(const wetts_model_t* m, float* out_dev, int64_t numel, void* stream);

/* Scratch needed by any of the stage calls below for a batch of B utterances, Tx phonemes
 * (padded) and Ty frames (padded).  Pass Ty = 0 for the pre-length-regulation stages only.
 * (The reference allocates every intermediate through PyTorch's caching allocator inside infer(), models.py:228-280;
 * here no stage call allocates.) */
int64_t wetts_workspace_bytes(const wetts_model_t* m, int32_t B, int32_t Tx, int32_t Ty);

/* ---- stage entry points (stream-ordered) -------------------------------------------------- */

/* a2 emb_g lookup (models.py:238-241).  g_out [B, gin].  sid may be NULL iff n_speakers==0
 * (g_out is then zero-filled and ignored downstream, matching g=None). */
int32_t wetts_speaker_embedding(const wetts_model_t* m, const int64_t* sid, int32_t B,
                                float* g_out, void* stream);

/* a3-a7 TextEncoder.forward (encoders.py:47-57; attentions.py:70-87,225-282,403-411;
 * normalization.py:16-19; commons.py:113-117).
 *   x [B,Tx] int64, x_lengths [B] int64
 *   x_enc [B,H,Tx], stats [B,2*inter,Tx] (m_p = channels [0,inter), logs_p = the rest),
 *   x_mask [B,Tx].  g [B,gin] (may be NULL) is only read by speaker-conditioned encoders
 *   (`enc_p(x, x_lengths, g=g)`, models.py:243). */
int32_t wetts_text_encoder(const wetts_model_t* m, const int64_t* x, const int64_t* x_lengths,
                           const float* g, int32_t B, int32_t Tx, float* x_enc, float* stats,
                           float* x_mask, void* workspace, int64_t workspace_bytes, void* stream);

/* a8 StochasticDurationPredictor.forward(reverse=True) (duration_predictors.py:213-219,254-263;
 * transforms.py:47-187).  eps_w [B,2,Tx] is the caller's standard-normal draw (the reference
 * calls torch.randn at :257); it is scaled by noise_scale_w inside.  logw [B,Tx].
 * status_dev (int32[1], may be NULL = the word registered with wetts_set_status_word) gets
 * WETTS_STATUS_SPLINE_DOMAIN OR-ed in on the device if the reference would have failed
 * `assert (discriminant >= 0)` (transforms.py:171); it is NOT cleared here. */
int32_t wetts_duration_sdp(const wetts_model_t* m, const float* x_enc, const float* x_mask,
                           const float* g, const float* eps_w, float noise_scale_w, int32_t B,
                           int32_t Tx, float* logw, int32_t* status_dev, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* a9 DurationPredictor.forward (duration_predictors.py:297-311). */
int32_t wetts_duration_dp(const wetts_model_t* m, const float* x_enc, const float* x_mask,
                          const float* g, int32_t B, int32_t Tx, float* logw, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* a10 durations -> lengths (models.py:254-256): w = exp(logw)*mask*length_scale,
 * w_ceil = ceil(w) [B,Tx], cum = inclusive cumsum(w_ceil) [B,Tx] (float, exact integers),
 * y_lengths = clamp_min(sum,1) [B] int64.  The caller reads y_lengths back (the one host sync
 * the reference also has: commons.py:114-115 `length.max()`).  status_dev (may be NULL) gets
 * WETTS_STATUS_DURATION_NONFINITE when a total is NaN / inf (y_lengths[b] is then 1). */
int32_t wetts_durations_to_lengths(const float* logw, const float* x_mask, float length_scale,
                                   int32_t B, int32_t Tx, float* w_ceil, float* cum,
                                   int64_t* y_lengths, int32_t* status_dev, void* stream);

/* a10-a12 sequence_mask + generate_path + prior expansion + sampling (models.py:257-267,
 * commons.py:113-136).  Ty = max(y_lengths) chosen by the caller.
 *   frame2phone [B,Ty] int32: the working form of the alignment (index of the phoneme each
 *   frame copies, -1 = none); y_mask [B,Ty]; attn [B,Ty,Tx] (may be NULL to skip materialising
 *   the dense 0/1 path of generate_path); m_p / logs_p come from `stats` [B,2*inter,Tx];
 *   outputs m_p_exp, logs_p_exp (both may be NULL), z_p [B,inter,Ty];
 *   eps_z[b,c,t] at eps_z + b*eps_batch_stride + c*eps_channel_stride + t is the caller's
 *   standard-normal draw (torch.randn_like at models.py:267). */
int32_t wetts_length_regulate(const wetts_model_t* m, const float* stats, const float* cum,
                              const float* x_mask, const int64_t* y_lengths, const float* eps_z,
                              int64_t eps_batch_stride, int64_t eps_channel_stride,
                              float noise_scale, int32_t B, int32_t Tx, int32_t Ty,
                              int32_t* frame2phone, float* y_mask, float* attn, float* m_p_exp,
                              float* logs_p_exp, float* z_p, void* stream);

/* a13 ResidualCouplingTransformersBlock.forward(reverse=True) (flows.py:442-449,494-513;
 * modules.py:60-106; commons.py:98-105).  z_p, z_out [B,inter,Ty]; z_out is NOT masked
 * (the reference returns z unmasked, models.py:280). */
int32_t wetts_flow_reverse(const wetts_model_t* m, const float* z_p, const float* y_mask,
                           const float* g, int32_t B, int32_t Ty, float* z_out, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* a14 Generator.forward (decoders.py:63-82,157-170,205-214).  z [B,inter,L] with arbitrary
 * batch stride `z_batch_stride` (floats) so a time-slice (z*y_mask)[:,:,:max_len] or a
 * streaming chunk (export_decoder_forward, models.py:360-363) needs no copy; y_mask may be
 * NULL (no masking) or [B, >=L] with row stride mask_stride.  audio [B, L*hop]. */
int32_t wetts_hifigan(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                      int64_t z_channel_stride, const float* y_mask, int64_t mask_stride,
                      const float* g, int32_t B, int32_t L, float* audio, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* Ragged form of wetts_hifigan.  The generator has no masks (decoders.py:63-82), so in a padded batch every
 * utterance is decoded to the batch's longest -- and the reference's own CLI avoids that by synthesising one
 * utterance per call (inference.py:83-110).  This entry gives a batch that call's arithmetic: utterance b is
 * decoded over its OWN y_lengths[b] frames (int64 device array, each <= L), exactly as
 * `dec(z[b:b+1, :, :y_lengths[b]], g[b:b+1])` would decode it alone -- zero padding at its own end, tiles behind
 * its end are not computed -- while z / audio keep their dense [B, ., L] / [B, L*hop] layout; samples of row b
 * behind y_lengths[b]*hop are written as 0.  float32 ResBlock1 HiFi-GAN models (wetts_hifigan_ragged_supported
 * returns 1); same workspace as wetts_hifigan. */
int32_t wetts_hifigan_ragged_supported(const wetts_model_t* m);
int32_t wetts_hifigan_ragged(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                             int64_t z_channel_stride, const int64_t* y_lengths, const float* g,
                             int32_t B, int32_t L, float* audio, void* workspace,
                             int64_t workspace_bytes, void* stream);

/* Decoder arithmetic: 0 = float32 (default; exact-f32 MFMA, the parity-gated path),
 * 1 = bfloat16, 2 = IEEE half activations / weights with f32 accumulation (BASELINE.json
 * configs[2] / configs[4] precision; the text encoder, duration predictor and flow stay f32).  The 16-bit weight
 * copies are packed by this call (all or nothing: a failure leaves the model unpacked and is returned here).  At 16 bit every ResBlock1 (c1, c2) pair with <= 128 channels runs as ONE fused
 * kernel (intermediate kept in LDS); OR-ing WETTS_DECODER_UNFUSED into `precision` forces the
 * two-launch form, which is bit-identical (diagnostics / tests). */
/* 3 = uint8 dynamic quantisation: the decoder graph `export_onnx.py --quant` leaves behind
 * (wetts/vits/export_onnx.py:149-157, onnxruntime quantize_dynamic with QUInt8 weights): every Conv1d
 * becomes DynamicQuantizeLinear (per launch, per tensor) -> ConvInteger (int32 accumulate, here on
 * v_mfma_i32_32x32x32_i8) -> scale + bias; ConvTranspose1d and the element-wise ops stay float32.
 * onnxruntime is absent from the reference tree, so this variant's parity is unpinned (the oracle
 * restates the published operator definitions). */
#define WETTS_DECODER_UINT8_DYNAMIC 3
#define WETTS_DECODER_UNFUSED 0x10
/* f32 HiFi-GAN: by default the k = 3 / 7 / 11 ResBlock chains of a stage run on three HIP streams forked from / joined to the
 * call's stream (they are independent until the MRF sum; one chain's launch tails are filled by the others' work).
 * OR-ing WETTS_DECODER_SERIAL into `precision` keeps every launch on the call's stream, one after the other (the round-4
 * schedule with grouped launches): bit-identical, and the form in which a kernel trace's per-kernel durations do not
 * overlap (measurement). */
#define WETTS_DECODER_SERIAL 0x20
int32_t wetts_set_decoder_precision(const wetts_model_t* m, int32_t precision);

/* iSTFT head of a Vocos model (decoders.py:300-304).  The reference has TWO, selected by the module's `is_onnx` flag:
 *   WETTS_ISTFT_TORCH (0)  torchaudio InverseSpectrogram == torch.istft(hann, center=True): frames = irfft(S) * hann,
 *                          overlap-add, DIVIDED by the overlap-added hann^2 envelope, n_fft/2 trimmed per side
 *                          -- what SynthesizerTrn.infer() of a model built by inference.py computes;
 *   WETTS_ISTFT_ONNX  (1)  OnnxSTFT.inverse (utils/stft.py:325-340): conv_transpose1d with
 *                          pinv(scale * rDFT basis)^T * hann, scale = n_fft / hop (== irfft * hann / scale), overlap-add,
 *                          same trim, NO envelope division -- what every graph export_onnx.py writes computes
 *                          (export_onnx.py:59 sets hps.model.is_onnx = True), i.e. the arithmetic behind
 *                          inference_onnx.py, cli/model.py, runtime/core/model/vits_model.cc and the Triton repos.
 *                          At hop = n_fft / 4 the interior is 0.375 x the torch.istft audio; the first and last
 *                          n_fft/2 samples differ in shape as well.
 * The mode is the config's is_onnx at create; this call switches a live model (both bases are built at create; a host
 * flag read at launch time, so it applies to the calls issued after it).  HiFi-GAN models accept and ignore it. */
#define WETTS_ISTFT_TORCH 0
#define WETTS_ISTFT_ONNX 1
int32_t wetts_set_istft_mode(const wetts_model_t* m, int32_t mode);
int32_t wetts_get_istft_mode(const wetts_model_t* m);

/* One Conv1d ("same" padding) as a dynamically quantised ONNX graph computes it -- the building block
 * of WETTS_DECODER_UINT8_DYNAMIC, exposed for validation: DynamicQuantizeLinear over the whole x,
 * per-tensor uint8 weights, ConvInteger on the int8 matrix cores, dequantise + bias.  x [B,Cin,T],
 * w [Cout,Cin,k] (natural PyTorch layout), bias [Cout] or NULL, out [B,Cout,T]; all device pointers.
 * Unlike the stage calls it allocates (packed weights, scratch) and synchronises the stream. */
int32_t wetts_dynamic_quant_conv1d(const float* x, const float* w, const float* bias, int32_t B,
                                   int32_t Cin, int32_t Cout, int32_t k, int32_t dilation,
                                   int32_t padding, int32_t T, float* out, void* stream);

/* Arithmetic of the flow's WaveNet layers (modules.py:60-87: in_layers k = 5, the gate, res_skip 1x1,
 * the residual / skip update): 0 = float32 (default, the parity-gated path), 1 = bfloat16,
 * 2 = IEEE half activations / weights with f32 accumulation and an f32 skip sum; pre / post /
 * cond_layer convs, the coupling and everything else stay f32.  BASELINE.json configs[2] ("bf16")
 * precision for the part of the step that dominates at B = 64.  The 16-bit weight copies are packed by this call. */
int32_t wetts_set_flow_precision(const wetts_model_t* m, int32_t precision);

/* a15 monotonic_align.maximum_path (utils/monotonic_align.py:6-57).  Needs no model.
 *   neg_cent [B,Ty,Tx] float32 (not modified), t_ys / t_xs int32[B] (the mask sums the
 *   reference derives at :16-17), path [B,Ty,Tx] int32 (zero-filled then the 1s written),
 *   workspace >= B*Ty*Tx*4 bytes (the in-place DP table the reference keeps in `values`). */
int32_t wetts_mas(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int32_t B,
                  int32_t Ty, int32_t Tx, int32_t* path, void* workspace, int64_t workspace_bytes,
                  void* stream);

/* a16 inference.py:100-110 output scaling: per utterance peak-normalise to 0.6 full scale,
 * clip, convert to int16.  lengths_samples [B] int64 = valid samples per row (peak is taken
 * over the valid part only when non-NULL, else over all L samples). */
int32_t wetts_audio_to_int16(const float* audio, const int64_t* lengths_samples, int32_t B,
                             int64_t L, int16_t* pcm, void* stream);

/* Standard-normal draws on the device (Philox4x32-10 + Box-Muller): what the reference gets from
 * torch.randn (duration_predictors.py:257) / torch.randn_like (models.py:267).  Element i is a
 * function of (seed, offset, i) only; a draw of n values consumes ceil(n/4) counter steps. */
int32_t wetts_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream);

/* out[b,c,t] = x[b,c,t] * mask[b,t]  (`z * y_mask`, models.py:322). */
int32_t wetts_mask_rows(const float* x, const float* mask, int32_t B, int32_t C, int32_t T,
                        float* out, void* stream);

/* a1 SynthesizerTrn.infer (models.py:228-280) composed for native hosts (the shape of
 * runtime/core/model/vits_model.h:37 Forward()).  Synchronises the stream once internally to
 * read y_lengths.  The caller provides capacity for max_frames frames; on return
 * *frames_out = max(y_lengths) (<= max_frames, else WETTS_E_WORKSPACE), y_lengths_host[B].
 * eps_w [B,2,Tx] and eps_z [B,inter,max_frames] (row stride max_frames) are standard-normal
 * draws; either may be NULL, in which case it is drawn here from the model's Philox stream
 * (wetts_set_seed) -- eps_w as [B,2,Tx] before the duration predictor, eps_z as a packed [B,inter,*frames_out]
 * tensor once the frame count is known, the two draws the Python and C++ hosts make: a seed gives the same
 * audio whatever `max_frames` the caller passed.  Errors the reference raises from inside its modules come back as return codes
 * after that one synchronisation: WETTS_E_DOMAIN (spline discriminant, transforms.py:171; non-finite
 * durations) and WETTS_E_INVALID (phoneme / speaker id outside its table).  audio needs capacity
 * B*max_frames*hop floats and is written PACKED as [B, (*frames_out)*hop].  workspace >= wetts_infer_workspace_bytes(). */
int64_t wetts_infer_workspace_bytes(const wetts_model_t* m, int32_t B, int32_t Tx,
                                    int32_t max_frames);
int32_t wetts_infer(const wetts_model_t* m, const int64_t* x, const int64_t* x_lengths,
                    const int64_t* sid, const float* eps_w, const float* eps_z,
                    float noise_scale, float length_scale, float noise_scale_w, int32_t B,
                    int32_t Tx, int32_t max_frames, float* audio, int64_t* y_lengths_host,
                    int32_t* frames_out, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- measurement helpers ------------------------------------------------------------------ */

/* Algorithmic FLOPs / bytes of wetts_hifigan for one frame (SURVEY.md §8d formulas), so
 * bench.py prices the roofline from the same shapes the kernels run. */
int32_t wetts_hifigan_cost(const wetts_config_t* cfg, double* flops_per_frame,
                           double* bytes_per_frame_perconv, double* mrf_flops_per_frame,
                           double* mrf_bytes_per_frame_perconv);

/* Live timing of the dominant kernel class (every ResBlock conv of the MRF stack) inside
 * normal wetts_hifigan / wetts_infer calls: when enabled, a HIP event pair is recorded on the
 * call's stream around each stage's ResBlock launches (no synchronisation at record time).
 * wetts_read_mrf_timing synchronises the recorded events and returns the summed device time,
 * the number of MRF conv launches and of hifigan calls since enabling; set(…,0/1) resets. */
int32_t wetts_set_mrf_timing(const wetts_model_t* m, int32_t enable);
int32_t wetts_read_mrf_timing(const wetts_model_t* m, double* mrf_ms, int64_t* conv_launches,
                              int32_t* hifigan_calls);
/* Algorithmic HBM bytes of the same launches AT THE GRANULARITY THEY WERE LAUNCHED WITH, summed since enabling: per
 * launch every [B][C][T] plane it has to read or write once (a fused ResBlock / stage: x in, running sum in and out; a
 * single conv: input, residual, output).  SURVEY.md 8(d)'s per-conv figure (wetts_hifigan_cost) counts planes a fused
 * launch never moves through HBM, so a bandwidth fraction priced with it can exceed 1; this one cannot. */
int32_t wetts_read_mrf_bytes(const wetts_model_t* m, double* launched_bytes);

/* Times `iters` launches of the dominant MRF conv kernel class (all ResBlock convs of the
 * decoder) with HIP events on `stream`; returns total ms and the number of conv launches.
 * Used by bench.py for roofline.achieved. */
int32_t wetts_profile_hifigan(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                              int64_t z_channel_stride, const float* g, int32_t B, int32_t L,
                              float* audio, void* workspace, int64_t workspace_bytes, void* stream,
                              double* mrf_ms, double* total_ms, int32_t* mrf_launches);

#ifdef __cplusplus
}
#endif
#endif /* WETTS_HIP_H_ */
