// libwetts_hip.so -- C-ABI implementation: blob layout, model creation (weight packing) and the
// stream-ordered stage entry points that compose the reference's SynthesizerTrn.infer()
// (wetts/vits/model/models.py:228-280).  See include/wetts_hip.h for the contract.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"
#include "conv_bf16.h"
#include "resblock32.h"
#include "qconv_u8.h"
#include "kernels.h"

namespace wetts {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// blob layout
// ---------------------------------------------------------------------------------------------
struct TensorSpec {
  std::string name;
  int64_t shape[4];
  int64_t offset, numel;
};

struct Layout {
  std::vector<TensorSpec> specs;
  std::map<std::string, int> index;
  int64_t total = 0;
  void add(const std::string& name, int64_t d0, int64_t d1 = 0, int64_t d2 = 0, int64_t d3 = 0) {
    TensorSpec t;
    t.name = name;
    t.shape[0] = d0; t.shape[1] = d1; t.shape[2] = d2; t.shape[3] = d3;
    t.numel = d0 * (d1 ? d1 : 1) * (d2 ? d2 : 1) * (d3 ? d3 : 1);
    t.offset = total;
    total += (t.numel + 63) / 64 * 64;  // keep every tensor 256-byte aligned
    index[name] = (int)specs.size();
    specs.push_back(t);
  }
};

static std::string S(const char* fmt, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

static int validate_config(const wetts_config_t* c) {
  WETTS_REQUIRE(c != nullptr, "null config");
  WETTS_REQUIRE(c->n_vocab > 0 && c->inter_channels > 0 && c->hidden_channels > 0, "bad channels");
  WETTS_REQUIRE(c->inter_channels % 2 == 0, "inter_channels must be even (flows.py:469)");
  WETTS_REQUIRE(c->n_heads > 0 && c->hidden_channels % c->n_heads == 0, "bad n_heads");
  WETTS_REQUIRE(c->n_layers >= 0 && c->n_layers <= 64, "bad n_layers");
  WETTS_REQUIRE(c->kernel_size >= 1 && c->kernel_size <= 15, "bad kernel_size");
  WETTS_REQUIRE(c->resblock == 1 || c->resblock == 2, "resblock must be 1 or 2");
  WETTS_REQUIRE(c->n_resblock_kernels >= 1 && c->n_resblock_kernels <= WETTS_MAX_RB_KERNELS,
                "bad n_resblock_kernels");
  WETTS_REQUIRE(c->n_resblock_dilations >= 1 && c->n_resblock_dilations <= WETTS_MAX_RB_DILATIONS,
                "bad n_resblock_dilations");
  WETTS_REQUIRE(c->n_upsamples >= 1 && c->n_upsamples <= WETTS_MAX_STAGES, "bad n_upsamples");
  WETTS_REQUIRE(c->upsample_initial_channel >> c->n_upsamples >= 1, "too many upsample stages");
  for (int i = 0; i < c->n_upsamples; ++i) {
    WETTS_REQUIRE(c->upsample_rates[i] >= 1 && c->upsample_kernel_sizes[i] >= c->upsample_rates[i],
                  "bad upsample stage %d", i);
    WETTS_REQUIRE((c->upsample_kernel_sizes[i] - c->upsample_rates[i]) % 2 == 0,
                  "upsample kernel-stride must be even (decoders.py:47)");
  }
  for (int j = 0; j < c->n_resblock_kernels; ++j)
    WETTS_REQUIRE(c->resblock_kernel_sizes[j] % 2 == 1, "resblock kernel sizes must be odd");
  WETTS_REQUIRE(c->n_speakers >= 0 && c->gin_channels >= 0, "bad speaker config");
  WETTS_REQUIRE(c->n_speakers == 0 || c->gin_channels > 0, "n_speakers>0 needs gin_channels");
  WETTS_REQUIRE(c->window_size >= 0 && c->window_size <= 16, "bad window_size");
  WETTS_REQUIRE(c->flow_n_flows >= 1 && c->flow_wn_layers >= 1 && c->flow_kernel_size % 2 == 1,
                "bad flow config");
  WETTS_REQUIRE(c->sdp_n_flows >= 2, "bad sdp_n_flows");
  WETTS_REQUIRE(c->vocoder_type == 0 || c->vocoder_type == 1, "vocoder_type must be 0 (hifigan) or 1 (vocos)");
  WETTS_REQUIRE(c->transformer_flows >= 0 && c->transformer_flows <= 4,
                "transformer_flows must be 0, 1 (pre_conv), 2 (pre_conv2), 3 (mono_layer_inter_residual) or "
                "4 (mono_layer_post_residual)");
  WETTS_REQUIRE(c->transformer_flows == 0 || c->transformer_flows == 2 || (c->inter_channels / 2) % 2 == 0,
                "pre_conv / mono_layer flows need inter_channels/2 divisible by their 2 heads");
  WETTS_REQUIRE(c->transformer_flows != 2 || c->hidden_channels % 2 == 0,
                "pre_conv2 flows need hidden_channels divisible by their 2 heads");
  WETTS_REQUIRE(c->use_spk_conditioned_encoder == 0 ||
                    (c->n_speakers > 0 && c->gin_channels > 0 && c->n_layers > 2),
                "speaker-conditioned encoder needs speakers, gin_channels and n_layers > 2 "
                "(cond_layer_idx = 2, attentions.py:44-48)");
  WETTS_REQUIRE(c->is_onnx == 0 || c->is_onnx == 1, "is_onnx must be 0 or 1");
  if (c->vocoder_type == 1) {
    WETTS_REQUIRE(c->vocos_channels > 0 && c->vocos_h_channels > 0 && c->vocos_num_layers >= 1 &&
                      c->vocos_num_layers <= 64, "bad vocos channels / layers");
    WETTS_REQUIRE(c->istft_n_fft >= 4 && c->istft_n_fft % 2 == 0 && c->istft_hop_length >= 1 &&
                      c->istft_hop_length <= c->istft_n_fft, "bad istft config");
    WETTS_REQUIRE(c->istft_win_length == c->istft_n_fft, "istft win_length must equal n_fft");
  }
  return WETTS_OK;
}

static bool has_g(const wetts_config_t* c) { return c->n_speakers > 0 && c->gin_channels > 0; }
// "mono_layer_*" flow types (flows.py:391-425): [ResidualCouplingLayer, Flip, MonoTransformerFlowLayer] per flow, so
// coupling layer f sits at flow.flows.{3f} and its mono layer at flow.flows.{3f+2}; every other type is [layer, Flip]
static bool mono_flows(const wetts_config_t* c) { return c->transformer_flows >= 3; }
static int flow_key_stride(const wetts_config_t* c) { return mono_flows(c) ? 3 : 2; }
// the flow types whose Encoder runs on the I/2 channels of x0 (2 layers, 2 heads, window_size=None, FFN kernel 3)
static bool half_enc_flows(const wetts_config_t* c) { return c->transformer_flows == 1 || mono_flows(c); }
// that Encoder's tensors under `p`.pre_transformer (flows.py:111-119 and :256-264 build the same module)
static void add_half_encoder(Layout& L, const std::string& p, int Hh) {
  for (int l = 0; l < 2; ++l) {
    std::string a = p + S(".pre_transformer.attn_layers.%d", l);
    for (const char* n : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
      L.add(a + "." + n + ".weight", Hh, Hh, 1);
      L.add(a + "." + n + ".bias", Hh);
    }
    L.add(p + S(".pre_transformer.norm_layers_1.%d.gamma", l), Hh);
    L.add(p + S(".pre_transformer.norm_layers_1.%d.beta", l), Hh);
    std::string ff = p + S(".pre_transformer.ffn_layers.%d", l);
    L.add(ff + ".conv_1.weight", Hh, Hh, 3);
    L.add(ff + ".conv_1.bias", Hh);
    L.add(ff + ".conv_2.weight", Hh, Hh, 3);
    L.add(ff + ".conv_2.bias", Hh);
    L.add(p + S(".pre_transformer.norm_layers_2.%d.gamma", l), Hh);
    L.add(p + S(".pre_transformer.norm_layers_2.%d.beta", l), Hh);
  }
}

static void add_dds(Layout& L, const std::string& p, int C, int k, int n) {
  for (int i = 0; i < n; ++i) {
    L.add(p + S(".convs_sep.%d.weight", i), C, 1, k);
    L.add(p + S(".convs_sep.%d.bias", i), C);
    L.add(p + S(".convs_1x1.%d.weight", i), C, C, 1);
    L.add(p + S(".convs_1x1.%d.bias", i), C);
    L.add(p + S(".norms_1.%d.gamma", i), C);
    L.add(p + S(".norms_1.%d.beta", i), C);
    L.add(p + S(".norms_2.%d.gamma", i), C);
    L.add(p + S(".norms_2.%d.beta", i), C);
  }
}

static void build_layout(const wetts_config_t* c, Layout& L) {
  const int H = c->hidden_channels, I = c->inter_channels, F = c->filter_channels;
  const int dk = H / c->n_heads, W = 2 * c->window_size + 1, gin = c->gin_channels;
  L.add("enc_p.emb.weight", c->n_vocab, H);
  for (int l = 0; l < c->n_layers; ++l) {
    std::string a = S("enc_p.encoder.attn_layers.%d", l);
    L.add(a + ".emb_rel_k", 1, W, dk);
    L.add(a + ".emb_rel_v", 1, W, dk);
    for (const char* n : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
      L.add(a + "." + n + ".weight", H, H, 1);
      L.add(a + "." + n + ".bias", H);
    }
    L.add(S("enc_p.encoder.norm_layers_1.%d.gamma", l), H);
    L.add(S("enc_p.encoder.norm_layers_1.%d.beta", l), H);
    std::string f = S("enc_p.encoder.ffn_layers.%d", l);
    L.add(f + ".conv_1.weight", F, H, c->kernel_size);
    L.add(f + ".conv_1.bias", F);
    L.add(f + ".conv_2.weight", H, F, c->kernel_size);
    L.add(f + ".conv_2.bias", H);
    L.add(S("enc_p.encoder.norm_layers_2.%d.gamma", l), H);
    L.add(S("enc_p.encoder.norm_layers_2.%d.beta", l), H);
  }
  if (c->use_spk_conditioned_encoder) {  // nn.Linear(gin, hidden) (attentions.py:41-43)
    L.add("enc_p.encoder.spk_emb_linear.weight", H, gin);
    L.add("enc_p.encoder.spk_emb_linear.bias", H);
  }
  L.add("enc_p.proj.weight", 2 * I, H, 1);
  L.add("enc_p.proj.bias", 2 * I);
  if (has_g(c)) L.add("emb_g.weight", c->n_speakers, gin);

  if (c->use_sdp) {
    // StochasticDurationPredictor(hidden, 192, 3, 0.5, 4): filter_channels := in_channels
    // (duration_predictors.py:166); only the tensors the reverse branch touches.
    const int C = H;
    L.add("dp.pre.weight", C, H, 1);
    L.add("dp.pre.bias", C);
    L.add("dp.proj.weight", C, C, 1);
    L.add("dp.proj.bias", C);
    add_dds(L, "dp.convs", C, 3, 3);
    if (has_g(c)) {
      L.add("dp.cond.weight", C, gin, 1);
      L.add("dp.cond.bias", C);
    }
    L.add("dp.flows.0.m", 2, 1);
    L.add("dp.flows.0.logs", 2, 1);
    // flows = [EA, CF1, Flip, CF2, Flip, ...]; reverse drops CF1 (duration_predictors.py:255-256)
    for (int f = 1; f < c->sdp_n_flows; ++f) {
      std::string p = S("dp.flows.%d", 2 * f + 1);
      L.add(p + ".pre.weight", C, 1, 1);
      L.add(p + ".pre.bias", C);
      add_dds(L, p + ".convs", C, 3, 3);
      L.add(p + ".proj.weight", 29, C, 1);
      L.add(p + ".proj.bias", 29);
    }
  } else {
    const int Fd = c->dp_filter_channels;
    L.add("dp.conv_1.weight", Fd, H, 3);
    L.add("dp.conv_1.bias", Fd);
    L.add("dp.norm_1.gamma", Fd);
    L.add("dp.norm_1.beta", Fd);
    L.add("dp.conv_2.weight", Fd, Fd, 3);
    L.add("dp.conv_2.bias", Fd);
    L.add("dp.norm_2.gamma", Fd);
    L.add("dp.norm_2.beta", Fd);
    L.add("dp.proj.weight", 1, Fd, 1);
    L.add("dp.proj.bias", 1);
    if (has_g(c)) {
      L.add("dp.cond.weight", H, gin, 1);
      L.add("dp.cond.bias", H);
    }
  }

  for (int f = 0; f < c->flow_n_flows; ++f) {
    std::string p = S("flow.flows.%d", flow_key_stride(c) * f);
    // Encoder(half, half, n_heads=2, n_layers=2, kernel_size=3, window_size=None), flows.py:111-119
    if (c->transformer_flows == 1) add_half_encoder(L, p, I / 2);
    if (c->transformer_flows == 2) {
      // Encoder(hidden, hidden, n_heads=2, n_layers=1, kernel_size=flow kernel, window 4), flows.py:40-48
      const int dkf = H / 2, Wf = 2 * 4 + 1, fk2 = c->flow_kernel_size;
      std::string a = p + ".pre_transformer.attn_layers.0";
      L.add(a + ".emb_rel_k", 1, Wf, dkf);
      L.add(a + ".emb_rel_v", 1, Wf, dkf);
      for (const char* n : {"conv_q", "conv_k", "conv_v", "conv_o"}) {
        L.add(a + "." + n + ".weight", H, H, 1);
        L.add(a + "." + n + ".bias", H);
      }
      L.add(p + ".pre_transformer.norm_layers_1.0.gamma", H);
      L.add(p + ".pre_transformer.norm_layers_1.0.beta", H);
      L.add(p + ".pre_transformer.ffn_layers.0.conv_1.weight", H, H, fk2);
      L.add(p + ".pre_transformer.ffn_layers.0.conv_1.bias", H);
      L.add(p + ".pre_transformer.ffn_layers.0.conv_2.weight", H, H, fk2);
      L.add(p + ".pre_transformer.ffn_layers.0.conv_2.bias", H);
      L.add(p + ".pre_transformer.norm_layers_2.0.gamma", H);
      L.add(p + ".pre_transformer.norm_layers_2.0.beta", H);
    }
    L.add(p + ".pre.weight", H, I / 2, 1);
    L.add(p + ".pre.bias", H);
    for (int i = 0; i < c->flow_wn_layers; ++i) {
      L.add(p + S(".enc.in_layers.%d.weight", i), 2 * H, H, c->flow_kernel_size);
      L.add(p + S(".enc.in_layers.%d.bias", i), 2 * H);
      int rs = (i < c->flow_wn_layers - 1) ? 2 * H : H;
      L.add(p + S(".enc.res_skip_layers.%d.weight", i), rs, H, 1);
      L.add(p + S(".enc.res_skip_layers.%d.bias", i), rs);
    }
    if (has_g(c)) {
      L.add(p + ".enc.cond_layer.weight", 2 * H * c->flow_wn_layers, gin, 1);
      L.add(p + ".enc.cond_layer.bias", 2 * H * c->flow_wn_layers);
    }
    L.add(p + ".post.weight", I / 2, H, 1);
    L.add(p + ".post.bias", I / 2);
    if (mono_flows(c)) {
      // MonoTransformerFlowLayer(channels, hidden, mean_only=True): pre_transformer + post 1x1 on I/2 channels
      // (flows.py:242-269)
      std::string q = S("flow.flows.%d", 3 * f + 2);
      add_half_encoder(L, q, I / 2);
      L.add(q + ".post.weight", I / 2, I / 2, 1);
      L.add(q + ".post.bias", I / 2);
    }
  }

  if (c->vocoder_type == 1) {  // VocosGenerator (decoders.py:251-284)
    const int VC = c->vocos_channels, VH = c->vocos_h_channels, VO = c->istft_n_fft + 2;
    L.add("dec.in_conv.weight", VC, I, 1);
    L.add("dec.in_conv.bias", VC);
    if (has_g(c)) {
      L.add("dec.cond.weight", VC, gin, 1);
      L.add("dec.cond.bias", VC);
    }
    L.add("dec.norm_pre.gamma", VC);
    L.add("dec.norm_pre.beta", VC);
    for (int l = 0; l < c->vocos_num_layers; ++l) {
      std::string p = S("dec.layers.%d", l);
      L.add(p + ".dw_conv.weight", VC, 1, 3);
      L.add(p + ".dw_conv.bias", VC);
      L.add(p + ".norm.gamma", VC);
      L.add(p + ".norm.beta", VC);
      L.add(p + ".pw_conv1.weight", VH, VC, 1);
      L.add(p + ".pw_conv1.bias", VH);
      L.add(p + ".pw_conv2.weight", VC, VH, 1);
      L.add(p + ".pw_conv2.bias", VC);
      L.add(p + ".scale", 1, VC, 1);
    }
    L.add("dec.norm_post.gamma", VC);
    L.add("dec.norm_post.beta", VC);
    L.add("dec.out_conv.weight", VO, VC, 1);
    L.add("dec.out_conv.bias", VO);
    return;
  }
  const int C0 = c->upsample_initial_channel;
  L.add("dec.conv_pre.weight", C0, I, 7);
  L.add("dec.conv_pre.bias", C0);
  int ch = C0;
  for (int i = 0; i < c->n_upsamples; ++i) {
    L.add(S("dec.ups.%d.weight", i), ch, ch / 2, c->upsample_kernel_sizes[i]);
    L.add(S("dec.ups.%d.bias", i), ch / 2);
    ch /= 2;
    for (int j = 0; j < c->n_resblock_kernels; ++j) {
      int n = i * c->n_resblock_kernels + j;
      int k = c->resblock_kernel_sizes[j];
      for (int d = 0; d < c->n_resblock_dilations; ++d) {
        if (c->resblock == 1) {
          L.add(S("dec.resblocks.%d.convs1.%d.weight", n, d), ch, ch, k);
          L.add(S("dec.resblocks.%d.convs1.%d.bias", n, d), ch);
          L.add(S("dec.resblocks.%d.convs2.%d.weight", n, d), ch, ch, k);
          L.add(S("dec.resblocks.%d.convs2.%d.bias", n, d), ch);
        } else {
          L.add(S("dec.resblocks.%d.convs.%d.weight", n, d), ch, ch, k);
          L.add(S("dec.resblocks.%d.convs.%d.bias", n, d), ch);
        }
      }
    }
  }
  L.add("dec.conv_post.weight", 1, ch, 7);
  if (has_g(c)) {
    L.add("dec.cond.weight", C0, gin, 1);
    L.add("dec.cond.bias", C0);
  }
}

// ---------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------
struct DDS {
  const float *sep_w[3], *sep_b[3], *n1g[3], *n1b[3], *n2g[3], *n2b[3];
  PackedConv c1x1[3];
};

struct EncLayer {
  const float *rel_k, *rel_v, *n1g, *n1b, *n2g, *n2b;
  PackedConv qkv, o, f1, f2;  // qkv: conv_q / conv_k / conv_v concatenated along Cout (one launch)
};

struct ConvFlowW {
  const float *pre_w, *pre_b;
  DDS dds;
  PackedConv proj;
};

struct FlowW {
  PackedConv pre, post;
  std::vector<PackedConv> in_layers, res_skip;
  const float *cond_w = nullptr, *cond_b = nullptr;
  std::vector<EncLayer> pre_tr;  // VITS2 "pre_conv": Encoder on x0 (flows.py:111-119), else empty
  // "mono_layer_*": the MonoTransformerFlowLayer that follows this coupling layer (flows.py:242-324)
  std::vector<EncLayer> mono_tr;
  PackedConv mono_post;
};

struct RB {
  std::vector<PackedConv> c1, c2;  // c2 empty for ResBlock2
};

struct ConvNeXt {  // ConvNeXtLayer (decoders.py:221-248); `scale` folded into pw2
  const float *dw_w, *dw_b, *ng, *nb;
  PackedConv pw1, pw2;
};

}  // namespace wetts

using namespace wetts;

struct wetts_model {
  wetts_config_t cfg;
  Layout layout;
  float* blob = nullptr;
  std::vector<PackedConv*> all_packed;
  // encoder
  const float* emb = nullptr;
  std::vector<EncLayer> enc;
  PackedConv enc_proj;
  const float *enc_spk_w = nullptr, *enc_spk_b = nullptr;  // spk_emb_linear (speaker-conditioned)
  const float* emb_g = nullptr;
  // sdp
  PackedConv sdp_pre, sdp_proj;
  DDS sdp_dds;
  const float *dp_cond_w = nullptr, *dp_cond_b = nullptr;
  const float *ea_m = nullptr, *ea_logs = nullptr;
  std::vector<ConvFlowW> cflows;  // CF2..CFn in module order
  // dp
  PackedConv dp_c1, dp_c2, dp_proj;
  const float *dp_n1g = nullptr, *dp_n1b = nullptr, *dp_n2g = nullptr, *dp_n2b = nullptr;
  // flow
  std::vector<FlowW> flows;
  // decoder
  PackedConv conv_pre;
  std::vector<PackedConv> ups;
  std::vector<RB> rbs;
  const float* conv_post_w = nullptr;
  const float *dec_cond_w = nullptr, *dec_cond_b = nullptr;
  int hop = 1;
  // vocos decoder
  PackedConv v_in, v_out, v_istft, v_istft_onnx;
  mutable int istft_mode = 0;  // WETTS_ISTFT_TORCH / WETTS_ISTFT_ONNX (cfg.is_onnx at create, wetts_set_istft_mode)
  std::vector<ConvNeXt> v_layers;
  const float *v_npre_g = nullptr, *v_npre_b = nullptr, *v_npost_g = nullptr, *v_npost_b = nullptr;
  std::vector<float*> v_owned;  // device buffers built at create (scaled pw2 weights, iSTFT basis)
  // live MRF timing (wetts_set_mrf_timing): event pairs recorded around each stage's ResBlock
  // launches, resolved lazily by wetts_read_mrf_timing so the timed region is not perturbed.
  mutable bool mrf_timing = false;
  mutable std::vector<std::pair<hipEvent_t, hipEvent_t>> mrf_events;
  mutable int64_t mrf_launches = 0;
  mutable int32_t mrf_calls = 0;
  // algorithmic HBM bytes of the class AT THE GRANULARITY IT WAS LAUNCHED WITH (wetts_read_mrf_bytes): per launch, every
  // [B][C][T] plane it has to read or write once (a fused ResBlock: x in, sum in / out; a single conv: input, residual,
  // output) -- what bench.py prices the HBM roofline of the fused 16-bit classes with, instead of SURVEY 8(d)'s
  // per-conv figure, which counts planes a fused launch never moves
  mutable double mrf_bytes = 0;
  // MRF chains: the n_k ResBlocks of a stage are independent until their sum, so they run on
  // separate HIP streams (forked from / joined to the caller's stream with events); one chain's
  // launch tail and prologue/epilogue phases overlap another chain's MFMA work.
  // bf16 decoder (opt-in, wetts_set_decoder_precision): weights packed by the setter
  mutable int dec_precision = 0;  // 0 = f32, 1 = bf16, 2 = f16
  mutable int dec_unfused = 0;    // diagnostic: run ResBlock1 pairs as two conv launches
  mutable int dec_serial = 0;     // WETTS_DECODER_SERIAL: no three-stream fork of a stage's chains
  int fuse32_lds = 160 * 1024;    // largest f32 pair tile run fused (WETTS_FUSE32_LDS, bytes)
  int fuse32_kmax128 = 11;        // C>=128 pairs with this many taps or more stay unfused
  int fuse32_maxc = 32;           // widest stage whose f32 pairs run fused (WETTS_FUSE32_MAXC): since the
                                  // chunked kernel runs at 4 waves/SIMD it wins at C >= 64
                                  // (profiles/r01_conv32_fused_pair.txt; step 79.9 none / 78.4 C=32 / 78.9 all)
  int fuse32_kmax = 99;           // largest tap count fused at f32 (WETTS_FUSE32_KMAX)
  int fuse32_kwide = 3;           // ... but pairs with at most this many taps fuse at any width: k = 3 pairs are
                                  // short on MFMA work per byte (C=64 +10 %, C=128 +3 %; WETTS_FUSE32_KWIDE)
  int fuse2_maxc = 128;           // widest stage whose f32 ResBlock2 chains run fused (WETTS_FUSE2_MAXC)
  int fuse_min_blocks = 128;      // fused pair kernels need this many tiles (else: small unfused tiles)
  // ResBlock1 chain kernel (resblock_chain32.hip; profiles/r02_resblock_chain.txt): a whole ResBlock1
  // in one launch where the chain's halo discards at most this share of the tile (C = 32: k = 3, 7;
  // C = 64: k = 3), single pairs at C = 32 (every k) and for k <= chain_pair_kmax at any width
  int small_max_tiles = 256;      // conv launches of at most this many 64x64 tiles use conv_small_kernel (0: off)
  int chain_whole_waste_pct = 15; // WETTS_CHAIN_WHOLE_PCT (0 disables whole-ResBlock launches)
  int chain_whole_maxc = 64;      // WETTS_CHAIN_WHOLE_MAXC
  int chain_pair_maxc = 32;       // WETTS_CHAIN_PAIR_MAXC: widest stage whose pairs all run on the chain kernel
  int chain_pair_kmax = 3;        // WETTS_CHAIN_PAIR_KMAX: ... and pairs with at most this many taps at any width
  int fuse2_waste_pct = 20;       // ResBlock2 chains: max % of tile columns lost to c2's halo at C = 32
                                  // (HBM-bound, +6..30 %); half of it at C >= 64 (profiles/r01_conv32_rb2_chain.txt)
  // 16-bit WaveNet layers of the flow (opt-in, wetts_set_flow_precision): weights packed by the setter
  mutable int flow_precision = 0;  // 0 = f32, 1 = bf16, 2 = f16
  mutable std::vector<std::vector<PackedConvB>> b_wn_in, b_wn_rs;  // [flow][layer]
  mutable PackedConvB b_post;          // conv_post as a 32-row conv (row 0 = the weights), 32-channel models
  mutable float* b_post_wpad = nullptr;
  // uint8 dynamic-quantisation decoder (precision 3): Conv1d weights quantised by the setter
  mutable PackedQConv q_pre, q_cond, q_post;
  mutable std::vector<std::vector<PackedQConv>> q_c1, q_c2;  // per resblock
  mutable bool q_packed = false;
  // which 16-bit type the flow / decoder copies currently hold (0 = none); set only after EVERY layer is packed
  mutable int flow_packed_prec = 0, dec_packed_prec = 0;
  mutable std::vector<PackedConvB> b_ups;
  mutable PackedConvB b_pre;  // conv_pre of the 16-bit decoder (round 6: on the 16-bit kernel like every other conv of the mode)
  mutable std::vector<std::vector<PackedConvB>> b_c1, b_c2;  // per resblock
  // device status word the stage calls OR their WETTS_STATUS_* bits into (wetts_set_status_word)
  mutable int32_t* status_word = nullptr;
  // the model's own standard-normal stream (wetts_infer with eps == NULL)
  mutable uint64_t rng_seed = 0, rng_offset = 0;
  int mrf_streams = 1;
  // Stages of at most this many channels run their k = 3 / 7 / 11 chains on three streams (0: none; the default covers every
  // stage).  Round 5, same box, 20 steps (profiles/r05_mrf_fork_ab.txt): grouped launches on one stream 60.46 / 60.37
  // ms/step; fork for C <= 32: 60.31; <= 64: 60.16; <= 128: 60.08; every stage: 59.76 -- the chain-kernel launches and the
  // last c2 of each chain (which must add the running sum in chain order) cannot be grouped, and their ramps and tails are
  // what the other chains' work fills.  Bit-identical to the one-stream schedule (same launches, same sum order, by events).
  int mrf_fork_maxc = 1 << 20;
  // WETTS_TUNE dds_fused: a DDSConv of the duration predictor in one launch (dds_fused.hip).  1: for small launches
  // (B * ceil(Tx / 6) <= 128 blocks of 32 columns, 6 of them valid: encoder call 1.70 -> 1.63 ms at B = 1, Tx = 64);
  // 2: always (64-column tiles; no faster than the 12 launches it replaces, profiles/r03_dds_fused_ab.txt); 0: never
  // WETTS_TUNE stage2_pct: 16-bit decoder, a whole stage of ResBlock2 blocks in one launch when the widest c2 halo
  // wastes at most this share of the tile (resblock2_stage16.hip; bit-identical to chain by chain); 0 = never
  int stage2_pct = 30;
  int dds_fused = 1;
  int wn_gate = 1;      // WETTS_TUNE wn_gate: the f32 flow's gate in the in_layer conv's epilogue (0: gate_kernel on the 2H-row tensor)
  int wn_fuse = 1;      // WETTS_TUNE wn_fuse: the f32 flow's residual / skip update in the res_skip conv's epilogue (1: small launches, 2: always, 0: wn_update_kernel)
  int small_fork = 1;   // WETTS_TUNE small_fork: the chains of a small (streaming-window) stage on their own streams
  int conv_groups = 1;  // WETTS_TUNE conv_groups: independent single convs of a ResBlock1 step in one launch (0: one each)
  hipStream_t aux_stream[WETTS_MAX_RB_KERNELS] = {};
  hipEvent_t ev_fork = nullptr, ev_chain[WETTS_MAX_RB_KERNELS] = {};
  bool fork_ok = false;  // every handle above exists (else: the serial grouped schedule)

  const float* T(const std::string& name) const {
    auto it = layout.index.find(name);
    if (it == layout.index.end()) return nullptr;
    return blob + layout.specs[it->second].offset;
  }
};

namespace wetts {

struct Bump {
  char* base;
  int64_t cap, off = 0;
  bool ok = true;
  Bump(void* p, int64_t c) : base((char*)p), cap(c) {}
  template <typename T>
  T* take(int64_t n) {
    int64_t bytes = align_up(n * (int64_t)sizeof(T), 256);
    if (off + bytes > cap) {
      ok = false;
      return nullptr;
    }
    T* r = (T*)(base + off);
    off += bytes;
    return r;
  }
};

static int32_t pack(wetts_model* m, const std::string& wname, const std::string& bname, int Cout,
                    int Cin, int k, int dil, int pad, int transposed, int up, hipStream_t s,
                    PackedConv* pc, int rev_in = 0, int gate_H = 0) {
  const float* w = m->T(wname);
  WETTS_REQUIRE(w != nullptr, "tensor %s missing from layout", wname.c_str());
  const float* b = bname.empty() ? nullptr : m->T(bname);
  WETTS_TRY(pack_conv_weight(w, b, Cout, Cin, k, dil, pad, transposed, up, s, pc, rev_in, gate_H));
  m->all_packed.push_back(pc);
  return WETTS_OK;
}

// conv_q, conv_k, conv_v of a MultiHeadAttention (attentions.py:225-233) as ONE 1x1 conv with
// 3*H output channels: the three weights / biases are concatenated into an owned device buffer
static int32_t pack_qkv(wetts_model* m, const std::string& a, int H, hipStream_t s, PackedConv* pc) {
  float *w = nullptr, *b = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&w, (size_t)3 * H * H * sizeof(float)));
  m->v_owned.push_back(w);
  WETTS_HIP_CHECK(hipMalloc((void**)&b, (size_t)3 * H * sizeof(float)));
  m->v_owned.push_back(b);
  const char* names[3] = {"conv_q", "conv_k", "conv_v"};
  for (int i = 0; i < 3; ++i) {
    const float* wi = m->T(a + "." + names[i] + ".weight");
    const float* bi = m->T(a + "." + names[i] + ".bias");
    WETTS_REQUIRE(wi && bi, "tensor %s.%s missing from layout", a.c_str(), names[i]);
    WETTS_HIP_CHECK(hipMemcpyAsync(w + (size_t)i * H * H, wi, (size_t)H * H * sizeof(float),
                                   hipMemcpyDeviceToDevice, s));
    WETTS_HIP_CHECK(hipMemcpyAsync(b + (size_t)i * H, bi, (size_t)H * sizeof(float),
                                   hipMemcpyDeviceToDevice, s));
  }
  WETTS_TRY(pack_conv_weight(w, b, 3 * H, H, 1, 1, 0, 0, 0, s, pc));
  m->all_packed.push_back(pc);
  return WETTS_OK;
}

static int32_t load_dds(wetts_model* m, const std::string& p, int C, hipStream_t s, DDS* d) {
  for (int i = 0; i < 3; ++i) {
    d->sep_w[i] = m->T(p + S(".convs_sep.%d.weight", i));
    d->sep_b[i] = m->T(p + S(".convs_sep.%d.bias", i));
    d->n1g[i] = m->T(p + S(".norms_1.%d.gamma", i));
    d->n1b[i] = m->T(p + S(".norms_1.%d.beta", i));
    d->n2g[i] = m->T(p + S(".norms_2.%d.gamma", i));
    d->n2b[i] = m->T(p + S(".norms_2.%d.beta", i));
    WETTS_TRY(pack(m, p + S(".convs_1x1.%d.weight", i), p + S(".convs_1x1.%d.bias", i), C, C, 1, 1,
                   0, 0, 0, s, &d->c1x1[i]));
  }
  return WETTS_OK;
}

// VocosGenerator weights (decoders.py:251-284): 1x1 convs packed for the MFMA conv kernel,
// ConvNeXtLayer.scale folded into pw_conv2 (scale * (W u + b) = (scale W) u + scale b), and the
// iSTFT head expressed as one more 1x1 conv whose weight is the windowed inverse-rDFT basis.
static int32_t build_vocos(wetts_model* m, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, VC = c->vocos_channels, VH = c->vocos_h_channels;
  const int NF = c->istft_n_fft, VO = NF + 2;
  WETTS_TRY(pack(m, "dec.in_conv.weight", "dec.in_conv.bias", VC, I, 1, 1, 0, 0, 0, s, &m->v_in));
  m->dec_cond_w = m->T("dec.cond.weight");
  m->dec_cond_b = m->T("dec.cond.bias");
  m->v_npre_g = m->T("dec.norm_pre.gamma");
  m->v_npre_b = m->T("dec.norm_pre.beta");
  m->v_npost_g = m->T("dec.norm_post.gamma");
  m->v_npost_b = m->T("dec.norm_post.beta");
  m->v_layers.resize(c->vocos_num_layers);
  for (int l = 0; l < c->vocos_num_layers; ++l) {
    ConvNeXt& cn = m->v_layers[l];
    std::string p = S("dec.layers.%d", l);
    cn.dw_w = m->T(p + ".dw_conv.weight");
    cn.dw_b = m->T(p + ".dw_conv.bias");
    cn.ng = m->T(p + ".norm.gamma");
    cn.nb = m->T(p + ".norm.beta");
    WETTS_TRY(pack(m, p + ".pw_conv1.weight", p + ".pw_conv1.bias", VH, VC, 1, 1, 0, 0, 0, s, &cn.pw1));
    const float* w2 = m->T(p + ".pw_conv2.weight");
    const float* b2 = m->T(p + ".pw_conv2.bias");
    const float* sc = m->T(p + ".scale");
    WETTS_REQUIRE(w2 && b2 && sc, "vocos layer %d tensors missing", l);
    float *w2s = nullptr, *b2s = nullptr;
    WETTS_HIP_CHECK(hipMalloc((void**)&w2s, (size_t)VC * VH * sizeof(float)));
    m->v_owned.push_back(w2s);
    WETTS_HIP_CHECK(hipMalloc((void**)&b2s, (size_t)VC * sizeof(float)));
    m->v_owned.push_back(b2s);
    WETTS_TRY(k_scale_rows(w2, sc, VC, VH, w2s, s));
    WETTS_TRY(k_scale_rows(b2, sc, VC, 1, b2s, s));
    WETTS_TRY(pack_conv_weight(w2s, b2s, VC, VH, 1, 1, 0, 0, 0, s, &cn.pw2));
    m->all_packed.push_back(&cn.pw2);
  }
  WETTS_TRY(pack(m, "dec.out_conv.weight", "dec.out_conv.bias", VO, VC, 1, 1, 0, 0, 0, s, &m->v_out));
  float* basis = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&basis, (size_t)NF * VO * sizeof(float)));
  m->v_owned.push_back(basis);
  WETTS_TRY(k_istft_basis(NF, basis, s));
  WETTS_TRY(pack_conv_weight(basis, nullptr, NF, VO, 1, 1, 0, 0, 0, s, &m->v_istft));
  m->all_packed.push_back(&m->v_istft);
  // the second head: OnnxSTFT's inverse basis (utils/stft.py:272-290), used when the model is_onnx
  float* basis_onnx = nullptr;
  WETTS_HIP_CHECK(hipMalloc((void**)&basis_onnx, (size_t)NF * VO * sizeof(float)));
  m->v_owned.push_back(basis_onnx);
  WETTS_TRY(k_istft_basis(NF, basis_onnx, s, (double)c->istft_hop_length / (double)NF));
  WETTS_TRY(pack_conv_weight(basis_onnx, nullptr, NF, VO, 1, 1, 0, 0, 0, s, &m->v_istft_onnx));
  m->all_packed.push_back(&m->v_istft_onnx);
  m->istft_mode = c->is_onnx ? WETTS_ISTFT_ONNX : WETTS_ISTFT_TORCH;
  m->hop = c->istft_hop_length;
  return WETTS_OK;
}

static int32_t build_model(wetts_model* m, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, I = c->inter_channels, F = c->filter_channels;
  const int ks = c->kernel_size;
  m->emb = m->T("enc_p.emb.weight");
  m->enc_spk_w = m->T("enc_p.encoder.spk_emb_linear.weight");
  m->enc_spk_b = m->T("enc_p.encoder.spk_emb_linear.bias");
  m->enc.resize(c->n_layers);
  for (int l = 0; l < c->n_layers; ++l) {
    EncLayer& e = m->enc[l];
    std::string a = S("enc_p.encoder.attn_layers.%d", l);
    e.rel_k = m->T(a + ".emb_rel_k");
    e.rel_v = m->T(a + ".emb_rel_v");
    WETTS_TRY(pack_qkv(m, a, H, s, &e.qkv));
    WETTS_TRY(pack(m, a + ".conv_o.weight", a + ".conv_o.bias", H, H, 1, 1, 0, 0, 0, s, &e.o));
    e.n1g = m->T(S("enc_p.encoder.norm_layers_1.%d.gamma", l));
    e.n1b = m->T(S("enc_p.encoder.norm_layers_1.%d.beta", l));
    e.n2g = m->T(S("enc_p.encoder.norm_layers_2.%d.gamma", l));
    e.n2b = m->T(S("enc_p.encoder.norm_layers_2.%d.beta", l));
    std::string f = S("enc_p.encoder.ffn_layers.%d", l);
    // FFN _same_padding: pad_l = (k-1)//2, pad_r = k//2 (attentions.py:422-429)
    WETTS_TRY(pack(m, f + ".conv_1.weight", f + ".conv_1.bias", F, H, ks, 1, (ks - 1) / 2, 0, 0, s,
                   &e.f1));
    WETTS_TRY(pack(m, f + ".conv_2.weight", f + ".conv_2.bias", H, F, ks, 1, (ks - 1) / 2, 0, 0, s,
                   &e.f2));
  }
  WETTS_TRY(pack(m, "enc_p.proj.weight", "enc_p.proj.bias", 2 * I, H, 1, 1, 0, 0, 0, s,
                 &m->enc_proj));
  m->emb_g = m->T("emb_g.weight");

  m->dp_cond_w = m->T("dp.cond.weight");
  m->dp_cond_b = m->T("dp.cond.bias");
  if (c->use_sdp) {
    WETTS_TRY(pack(m, "dp.pre.weight", "dp.pre.bias", H, H, 1, 1, 0, 0, 0, s, &m->sdp_pre));
    WETTS_TRY(pack(m, "dp.proj.weight", "dp.proj.bias", H, H, 1, 1, 0, 0, 0, s, &m->sdp_proj));
    WETTS_TRY(load_dds(m, "dp.convs", H, s, &m->sdp_dds));
    m->ea_m = m->T("dp.flows.0.m");
    m->ea_logs = m->T("dp.flows.0.logs");
    m->cflows.resize(c->sdp_n_flows - 1);
    for (int f = 1; f < c->sdp_n_flows; ++f) {
      ConvFlowW& cf = m->cflows[f - 1];
      std::string p = S("dp.flows.%d", 2 * f + 1);
      cf.pre_w = m->T(p + ".pre.weight");
      cf.pre_b = m->T(p + ".pre.bias");
      WETTS_TRY(load_dds(m, p + ".convs", H, s, &cf.dds));
      WETTS_TRY(pack(m, p + ".proj.weight", p + ".proj.bias", 29, H, 1, 1, 0, 0, 0, s, &cf.proj));
    }
  } else {
    const int Fd = c->dp_filter_channels;
    WETTS_TRY(pack(m, "dp.conv_1.weight", "dp.conv_1.bias", Fd, H, 3, 1, 1, 0, 0, s, &m->dp_c1));
    WETTS_TRY(pack(m, "dp.conv_2.weight", "dp.conv_2.bias", Fd, Fd, 3, 1, 1, 0, 0, s, &m->dp_c2));
    WETTS_TRY(pack(m, "dp.proj.weight", "dp.proj.bias", 1, Fd, 1, 1, 0, 0, 0, s, &m->dp_proj));
    m->dp_n1g = m->T("dp.norm_1.gamma");
    m->dp_n1b = m->T("dp.norm_1.beta");
    m->dp_n2g = m->T("dp.norm_2.gamma");
    m->dp_n2b = m->T("dp.norm_2.beta");
  }

  m->flows.resize(c->flow_n_flows);
  for (int f = 0; f < c->flow_n_flows; ++f) {
    FlowW& fw = m->flows[f];
    std::string p = S("flow.flows.%d", flow_key_stride(c) * f);
    // plain coupling layers read x0 = Flip(x)[:I/2] = x[I-1 .. I/2]: the Flip is folded into `pre` by packing
    // its input channels in reverse, so the conv reads channels I/2 .. I-1 in place (no index arithmetic
    // while staging); the pre_conv transformer flow materialises x0 and keeps the natural order
    WETTS_TRY(pack(m, p + ".pre.weight", p + ".pre.bias", H, I / 2, 1, 1, 0, 0, 0, s, &fw.pre,
                   c->transformer_flows == 1 ? 0 : 1));
    WETTS_TRY(pack(m, p + ".post.weight", p + ".post.bias", I / 2, H, 1, 1, 0, 0, 0, s, &fw.post));
    fw.in_layers.resize(c->flow_wn_layers);
    fw.res_skip.resize(c->flow_wn_layers);
    const int fk = c->flow_kernel_size;
    for (int i = 0; i < c->flow_wn_layers; ++i) {
      // WN dilation_rate = 1 (models.py:133-138) => dilation 1**i = 1, padding (k-1)/2
      // rows interleaved (tanh row i, sigmoid row H + i): the gate runs in the conv's epilogue (OUT_GATE, common.h)
      WETTS_TRY(pack(m, p + S(".enc.in_layers.%d.weight", i), p + S(".enc.in_layers.%d.bias", i),
                     2 * H, H, fk, 1, (fk - 1) / 2, 0, 0, s, &fw.in_layers[i], 0, H));
      int rs = (i < c->flow_wn_layers - 1) ? 2 * H : H;
      WETTS_TRY(pack(m, p + S(".enc.res_skip_layers.%d.weight", i),
                     p + S(".enc.res_skip_layers.%d.bias", i), rs, H, 1, 1, 0, 0, 0, s,
                     &fw.res_skip[i]));
    }
    fw.cond_w = m->T(p + ".enc.cond_layer.weight");
    fw.cond_b = m->T(p + ".enc.cond_layer.bias");
    if (c->transformer_flows == 2) {
      fw.pre_tr.resize(1);
      EncLayer& e = fw.pre_tr[0];
      std::string a = p + ".pre_transformer.attn_layers.0";
      e.rel_k = m->T(a + ".emb_rel_k");
      e.rel_v = m->T(a + ".emb_rel_v");
      WETTS_TRY(pack_qkv(m, a, H, s, &e.qkv));
      WETTS_TRY(pack(m, a + ".conv_o.weight", a + ".conv_o.bias", H, H, 1, 1, 0, 0, 0, s, &e.o));
      e.n1g = m->T(p + ".pre_transformer.norm_layers_1.0.gamma");
      e.n1b = m->T(p + ".pre_transformer.norm_layers_1.0.beta");
      e.n2g = m->T(p + ".pre_transformer.norm_layers_2.0.gamma");
      e.n2b = m->T(p + ".pre_transformer.norm_layers_2.0.beta");
      std::string ff = p + ".pre_transformer.ffn_layers.0";
      WETTS_TRY(pack(m, ff + ".conv_1.weight", ff + ".conv_1.bias", H, H, fk, 1, (fk - 1) / 2, 0, 0, s, &e.f1));
      WETTS_TRY(pack(m, ff + ".conv_2.weight", ff + ".conv_2.bias", H, H, fk, 1, (fk - 1) / 2, 0, 0, s, &e.f2));
    }
    if (mono_flows(c)) {
      const std::string q = S("flow.flows.%d", 3 * f + 2);
      WETTS_TRY(pack(m, q + ".post.weight", q + ".post.bias", I / 2, I / 2, 1, 1, 0, 0, 0, s, &fw.mono_post));
    }
    if (half_enc_flows(c)) {
      const int Hh = I / 2;
      std::vector<EncLayer>& tr = mono_flows(c) ? fw.mono_tr : fw.pre_tr;
      if (mono_flows(c)) p = S("flow.flows.%d", 3 * f + 2);
      tr.resize(2);
      for (int l = 0; l < 2; ++l) {
        EncLayer& e = tr[l];
        std::string a = p + S(".pre_transformer.attn_layers.%d", l);
        e.rel_k = e.rel_v = nullptr;  // window_size=None
        WETTS_TRY(pack_qkv(m, a, Hh, s, &e.qkv));
        WETTS_TRY(pack(m, a + ".conv_o.weight", a + ".conv_o.bias", Hh, Hh, 1, 1, 0, 0, 0, s, &e.o));
        e.n1g = m->T(p + S(".pre_transformer.norm_layers_1.%d.gamma", l));
        e.n1b = m->T(p + S(".pre_transformer.norm_layers_1.%d.beta", l));
        e.n2g = m->T(p + S(".pre_transformer.norm_layers_2.%d.gamma", l));
        e.n2b = m->T(p + S(".pre_transformer.norm_layers_2.%d.beta", l));
        std::string ff = p + S(".pre_transformer.ffn_layers.%d", l);
        WETTS_TRY(pack(m, ff + ".conv_1.weight", ff + ".conv_1.bias", Hh, Hh, 3, 1, 1, 0, 0, s, &e.f1));
        WETTS_TRY(pack(m, ff + ".conv_2.weight", ff + ".conv_2.bias", Hh, Hh, 3, 1, 1, 0, 0, s, &e.f2));
      }
    }
  }

  if (c->vocoder_type == 1) return build_vocos(m, s);

  const int C0 = c->upsample_initial_channel;
  WETTS_TRY(pack(m, "dec.conv_pre.weight", "dec.conv_pre.bias", C0, I, 7, 1, 3, 0, 0, s,
                 &m->conv_pre));
  m->ups.resize(c->n_upsamples);
  m->rbs.resize(c->n_upsamples * c->n_resblock_kernels);
  int ch = C0;
  m->hop = 1;
  for (int i = 0; i < c->n_upsamples; ++i) {
    const int u = c->upsample_rates[i], uk = c->upsample_kernel_sizes[i];
    WETTS_TRY(pack(m, S("dec.ups.%d.weight", i), S("dec.ups.%d.bias", i), ch / 2, ch, uk, 1,
                   (uk - u) / 2, 1, u, s, &m->ups[i]));
    ch /= 2;
    m->hop *= u;
    for (int j = 0; j < c->n_resblock_kernels; ++j) {
      int n = i * c->n_resblock_kernels + j;
      int k = c->resblock_kernel_sizes[j];
      RB& rb = m->rbs[n];
      rb.c1.resize(c->n_resblock_dilations);
      if (c->resblock == 1) rb.c2.resize(c->n_resblock_dilations);
      for (int d = 0; d < c->n_resblock_dilations; ++d) {
        int dil = c->resblock_dilation_sizes[j][d];
        int pad = (k * dil - dil) / 2;  // get_padding, commons.py:13-14
        if (c->resblock == 1) {
          WETTS_TRY(pack(m, S("dec.resblocks.%d.convs1.%d.weight", n, d),
                         S("dec.resblocks.%d.convs1.%d.bias", n, d), ch, ch, k, dil, pad, 0, 0, s,
                         &rb.c1[d]));
          WETTS_TRY(pack(m, S("dec.resblocks.%d.convs2.%d.weight", n, d),
                         S("dec.resblocks.%d.convs2.%d.bias", n, d), ch, ch, k, 1, (k - 1) / 2, 0,
                         0, s, &rb.c2[d]));
        } else {
          WETTS_TRY(pack(m, S("dec.resblocks.%d.convs.%d.weight", n, d),
                         S("dec.resblocks.%d.convs.%d.bias", n, d), ch, ch, k, dil, pad, 0, 0, s,
                         &rb.c1[d]));
        }
      }
    }
  }
  m->conv_post_w = m->T("dec.conv_post.weight");
  m->dec_cond_w = m->T("dec.cond.weight");
  m->dec_cond_b = m->T("dec.cond.bias");
  return WETTS_OK;
}


// ---------------------------------------------------------------------------------------------
// workspace sizing
// ---------------------------------------------------------------------------------------------
static int64_t A256(int64_t n_floats) { return align_up(n_floats * 4, 256); }

static int64_t ws_encoder(const wetts_config_t* c, int B, int Tx) {
  const int64_t H = c->hidden_channels, F = c->filter_channels, nh = c->n_heads;
  int64_t n = 0;
  n += 5 * A256(B * H * Tx);          // q,k,v,att,y
  n += A256(B * F * Tx);              // ffn hidden
  n += A256(B * nh * (int64_t)Tx * Tx + B * H * Tx + B * nh * (2 * c->window_size + 1) * Tx);  // scores + vT + rel
  n += A256(B * H * Tx);              // x ping
  n += A256(B * H);                   // speaker conditioning vector
  return n;
}

static int64_t ws_sdp(const wetts_config_t* c, int B, int Tx) {
  const int64_t H = c->hidden_channels;
  return 5 * A256(B * H * Tx) + A256(B * 32 * Tx) + A256(B * 2 * Tx) + A256(B * H) + 256;
}

static int64_t ws_dp(const wetts_config_t* c, int B, int Tx) {
  const int64_t H = c->hidden_channels, Fd = c->dp_filter_channels;
  return A256(B * H * Tx) + 2 * A256(B * Fd * Tx) + A256(B * H);
}

static int64_t ws_flow(const wetts_config_t* c, int B, int Ty_) {
  const int64_t H = c->hidden_channels, I = c->inter_channels;
  const int64_t Ty = ((int64_t)Ty_ + 3) & ~3ll;  // rows padded to 4 frames (see wetts_flow_reverse)
  int64_t n = 3 * A256(B * I * Ty) + A256(B * Ty) + 3 * A256(B * H * Ty) + 2 * A256(B * 2 * H * Ty) +
              A256(B * (I / 2) * Ty) + A256(B * 2 * H * c->flow_wn_layers);
  // 16-bit WN mode: channel-last h, gate output (H), in_layer / res_skip outputs (2H) at 16 bit,
  // the skip sum at f32 channel-last
  n += 2 * align_up(B * H * Ty * 2, 256) + 2 * align_up(B * 2 * H * Ty * 2, 256) + A256(B * H * Ty);
  if (c->transformer_flows != 0) {
    const int64_t He = half_enc_flows(c) ? I / 2 : H;
    // the T*T score region only exists on the three-kernel attention path; the flash kernel (every
    // reference config: window-less, dk = 48) needs the transposed v and nothing else
    n += 9 * A256(B * He * Ty) +
         A256(attn_score_elems(half_enc_flows(c) ? -1 : 4, (int)(He / 2), B, 2, Ty) + B * He * Ty +
              (int64_t)B * 2 * 9 * Ty);
  }
  return n;
}

static int64_t dec_max_elems(const wetts_config_t* c, int B, int L) {
  int64_t ch = c->upsample_initial_channel, len = L, mx = (int64_t)B * ch * len;
  for (int i = 0; i < c->n_upsamples; ++i) {
    ch /= 2;
    len *= c->upsample_rates[i];
    int64_t e = (int64_t)B * ch * len;
    if (e > mx) mx = e;
  }
  return mx;
}

// The ConvNeXt stack keeps its rows padded to a multiple of 4 frames (16-byte rows: the LDS-DMA GEMM of gemm_pw.hip
// and 16-byte staging need that), and the iSTFT GEMM's input has its n_fft + 2 rows padded to a multiple of 16.
static int64_t vocos_fs(int L) { return ((int64_t)L + 1 + 3) & ~3ll; }
static int64_t vocos_rows(const wetts_config_t* c) { return ((int64_t)c->istft_n_fft + 2 + 15) & ~15ll; }

static int64_t ws_vocos(const wetts_config_t* c, int B, int L) {
  const int64_t F = vocos_fs(L), VC = c->vocos_channels, VH = c->vocos_h_channels, NF = c->istft_n_fft;
  return A256(B * c->inter_channels * F) + 3 * A256(B * VC * F) + A256(B * VH * F) +
         A256(B * (NF + 2) * F) + A256(B * vocos_rows(c) * F) + A256(B * NF * F) + A256(B * VC);
}

static int64_t ws_decoder(const wetts_config_t* c, int B, int L) {
  if (c->vocoder_type == 1) return ws_vocos(c, B, L);
  const int64_t f32need = (3 + 3 * (int64_t)c->n_resblock_kernels) * A256(dec_max_elems(c, B, L)) +
                          A256((int64_t)B * c->upsample_initial_channel);
  // uint8 variant: six stage-sized f32 buffers + the int8 image / channel sums of the widest input
  int64_t lenmax = L;
  for (int i = 0; i < c->n_upsamples; ++i) lenmax *= c->upsample_rates[i];
  const int64_t u8need = 6 * A256(dec_max_elems(c, B, L)) + A256((int64_t)B * c->upsample_initial_channel) +
                         align_up(dec_max_elems(c, B, L) + 32 * (int64_t)B * lenmax, 256) +
                         align_up(4 * (int64_t)B * lenmax, 256) + 2048 + 16384 +  // + the range slots
                         align_up((int64_t)B * (lenmax / 128 + 1) * 64, 256);      // + the per-block range records
  return f32need > u8need ? f32need : u8need;
}

}  // namespace wetts

// =============================================================================================
// C ABI
// =============================================================================================
extern "C" {

int32_t wetts_abi_version(void) { return WETTS_ABI_VERSION; }
const char* wetts_last_error(void) { return g_err; }

int32_t wetts_blob_num_tensors(const wetts_config_t* cfg) {
  if (validate_config(cfg) != WETTS_OK) return WETTS_E_INVALID;
  Layout L;
  build_layout(cfg, L);
  return (int32_t)L.specs.size();
}

int32_t wetts_blob_tensor_info(const wetts_config_t* cfg, int32_t index, char* name_buf,
                               size_t name_buf_len, int64_t* offset, int64_t* numel,
                               int64_t shape[4]) {
  WETTS_TRY(validate_config(cfg));
  Layout L;
  build_layout(cfg, L);
  WETTS_REQUIRE(index >= 0 && index < (int32_t)L.specs.size(), "tensor index %d out of range",
                index);
  const TensorSpec& t = L.specs[index];
  WETTS_REQUIRE(name_buf && name_buf_len > t.name.size(), "name buffer too small");
  strcpy(name_buf, t.name.c_str());
  if (offset) *offset = t.offset;
  if (numel) *numel = t.numel;
  if (shape)
    for (int i = 0; i < 4; ++i) shape[i] = t.shape[i];
  return WETTS_OK;
}

int64_t wetts_blob_numel(const wetts_config_t* cfg) {
  if (validate_config(cfg) != WETTS_OK) return WETTS_E_INVALID;
  Layout L;
  build_layout(cfg, L);
  return L.total;
}

int32_t wetts_create(const wetts_config_t* cfg, const float* blob_dev, int64_t blob_numel,
                     void* stream, wetts_model_t** out) {
  WETTS_TRY(validate_config(cfg));
  WETTS_REQUIRE(out != nullptr && blob_dev != nullptr, "null argument");
  hipStream_t s = (hipStream_t)stream;
  wetts_model* m = new wetts_model();
  m->cfg = *cfg;
  build_layout(cfg, m->layout);
  if (blob_numel != m->layout.total) {
    set_error("blob has %lld floats, layout needs %lld", (long long)blob_numel,
              (long long)m->layout.total);
    delete m;
    return WETTS_E_INVALID;
  }
  hipError_t e = hipMalloc((void**)&m->blob, (size_t)blob_numel * sizeof(float));
  if (e != hipSuccess) {
    set_error("hipMalloc(blob) failed: %s", hipGetErrorString(e));
    delete m;
    return WETTS_E_HIP;
  }
  e = hipMemcpyAsync(m->blob, blob_dev, (size_t)blob_numel * sizeof(float),
                     hipMemcpyDeviceToDevice, s);
  int32_t r = (e == hipSuccess) ? build_model(m, s) : WETTS_E_HIP;
  {
    // A/B switches for measurements (none is needed in production): ONE variable,
    // WETTS_TUNE="name=value,name=value", names as in the table below
    struct Knob { const char* name; int* field; };
    const Knob knobs[] = {
        // (round 6: the table is down to the switches a TEST needs to reach a code path -- fused / unfused forms that
        // must stay bit-identical, coverage shapes; the twelve whose A/B is settled are constants of the model now)
        {"stage2_pct", &m->stage2_pct}, {"dds_fused", &m->dds_fused}, {"wn_fuse", &m->wn_fuse}, {"wn_gate", &m->wn_gate},
        {"fuse2_waste_pct", &m->fuse2_waste_pct}, {"fuse_min_blocks", &m->fuse_min_blocks},
        {"chain_whole_pct", &m->chain_whole_waste_pct}, {"chain_whole_maxc", &m->chain_whole_maxc},
        {"small_max_tiles", &m->small_max_tiles},
    };
    if (const char* env = getenv("WETTS_TUNE")) {
      std::string all(env);
      size_t pos = 0;
      while (pos < all.size()) {
        size_t end = all.find(',', pos);
        if (end == std::string::npos) end = all.size();
        const std::string item = all.substr(pos, end - pos);
        const size_t eq = item.find('=');
        bool known = false;
        if (eq != std::string::npos)
          for (const Knob& k : knobs)
            if (item.compare(0, eq, k.name) == 0 && strlen(k.name) == eq) {
              *k.field = atoi(item.c_str() + eq + 1);
              known = true;
            }
        if (!known && !item.empty()) {
          set_error("WETTS_TUNE: unknown setting '%s'", item.c_str());
          r = WETTS_E_INVALID;
        }
        pos = end + 1;
      }
    }
    if (r == WETTS_OK && !m->wn_gate) {  // A/B: the in_layers back in the reference's row order, gate_kernel behind them
      const int H = cfg->hidden_channels, fk = cfg->flow_kernel_size;
      for (int f = 0; f < cfg->flow_n_flows && r == WETTS_OK; ++f)
        for (int i = 0; i < cfg->flow_wn_layers && r == WETTS_OK; ++i) {
          const std::string p = S("flow.flows.%d.enc.in_layers.%d", flow_key_stride(cfg) * f, i);
          free_packed(&m->flows[f].in_layers[i]);
          r = pack_conv_weight(m->T(p + ".weight"), m->T(p + ".bias"), 2 * H, H, fk, 1, (fk - 1) / 2, 0, 0, s,
                               &m->flows[f].in_layers[i]);
        }
    }
    if (m->mrf_streams < 1) m->mrf_streams = 1;
    if (m->mrf_streams > cfg->n_resblock_kernels) m->mrf_streams = cfg->n_resblock_kernels;
    // the fork's handles; if any cannot be had the model keeps to the one-stream schedule (fork_ok = false): a null
    // aux stream would be the legacy default stream -- illegal inside a graph capture, implicitly synchronising outside
    m->fork_ok = hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming) == hipSuccess && m->ev_fork;
    for (int j = 0; j < cfg->n_resblock_kernels; ++j) {
      if (hipEventCreateWithFlags(&m->ev_chain[j], hipEventDisableTiming) != hipSuccess || !m->ev_chain[j]) m->fork_ok = false;
      if (j > 0)  // one per ResBlock chain: the default f32 fork, mrf_streams and small_fork (streaming windows) use them
        if (hipStreamCreateWithFlags(&m->aux_stream[j], hipStreamNonBlocking) != hipSuccess || !m->aux_stream[j]) m->fork_ok = false;
    }
    (void)hipGetLastError();  // a failed creation must not surface as the next launch's error
  }
  if (r == WETTS_OK) {
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      set_error("weight packing failed: %s", hipGetErrorString(e));
      r = WETTS_E_HIP;
    }
  }
  if (r != WETTS_OK) {
    wetts_destroy(m);
    return r;
  }
  *out = m;
  return WETTS_OK;
}

void wetts_destroy(wetts_model_t* m) {
  if (!m) return;
  for (PackedConv* pc : m->all_packed) free_packed(pc);
  for (auto& pc : m->b_ups) free_packed_bf16(&pc);
  for (auto& v : m->b_c1) for (auto& pc : v) free_packed_bf16(&pc);
  for (auto& v : m->b_c2) for (auto& pc : v) free_packed_bf16(&pc);
  for (auto& v : m->b_wn_in) for (auto& pc : v) free_packed_bf16(&pc);
  for (auto& v : m->b_wn_rs) for (auto& pc : v) free_packed_bf16(&pc);
  free_packed_bf16(&m->b_post);
  if (m->b_post_wpad) (void)hipFree(m->b_post_wpad);
  free_packed_qconv(&m->q_pre);
  free_packed_qconv(&m->q_cond);
  free_packed_qconv(&m->q_post);
  for (auto& v : m->q_c1) for (auto& pc : v) free_packed_qconv(&pc);
  for (auto& v : m->q_c2) for (auto& pc : v) free_packed_qconv(&pc);
  for (int j = 0; j < WETTS_MAX_RB_KERNELS; ++j) {
    if (m->aux_stream[j]) (void)hipStreamDestroy(m->aux_stream[j]);
    if (m->ev_chain[j]) (void)hipEventDestroy(m->ev_chain[j]);
  }
  if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
  for (float* p : m->v_owned) (void)hipFree(p);
  if (m->blob) (void)hipFree(m->blob);
  delete m;
}

int32_t wetts_hop_length(const wetts_model_t* m) { return m ? m->hop : WETTS_E_INVALID; }

int32_t wetts_set_istft_mode(const wetts_model_t* m, int32_t mode) {
  WETTS_REQUIRE(m, "null model");
  WETTS_REQUIRE(mode == WETTS_ISTFT_TORCH || mode == WETTS_ISTFT_ONNX, "istft mode must be 0 (torch.istft) or 1 (OnnxSTFT.inverse), got %d", mode);
  m->istft_mode = mode;  // HiFi-GAN models carry it unused (decoders.py:17-61 takes no is_onnx)
  return WETTS_OK;
}

int32_t wetts_get_istft_mode(const wetts_model_t* m) { return m ? m->istft_mode : WETTS_E_INVALID; }

int32_t wetts_get_blob(const wetts_model_t* m, float* out_dev, int64_t numel, void* stream) {
  WETTS_REQUIRE(m && out_dev, "null argument");
  WETTS_REQUIRE(numel == m->layout.total, "blob has %lld floats, asked for %lld", (long long)m->layout.total,
                (long long)numel);
  WETTS_HIP_CHECK(hipMemcpyAsync(out_dev, m->blob, (size_t)numel * sizeof(float), hipMemcpyDeviceToDevice,
                                 (hipStream_t)stream));
  return WETTS_OK;
}

int64_t wetts_workspace_bytes(const wetts_model_t* m, int32_t B, int32_t Tx, int32_t Ty) {
  if (!m || B < 0 || Tx < 0 || Ty < 0) return WETTS_E_INVALID;
  const wetts_config_t* c = &m->cfg;
  int64_t a = ws_encoder(c, B, Tx);
  int64_t b = c->use_sdp ? ws_sdp(c, B, Tx) : ws_dp(c, B, Tx);
  int64_t need = a > b ? a : b;
  if (Ty > 0) {
    int64_t f = ws_flow(c, B, Ty), d = ws_decoder(c, B, Ty);
    if (f > need) need = f;
    if (d > need) need = d;
  }
  return need + 4096;
}

int32_t wetts_speaker_embedding(const wetts_model_t* m, const int64_t* sid, int32_t B,
                                float* g_out, void* stream) {
  WETTS_REQUIRE(m && g_out, "null argument");
  hipStream_t s = (hipStream_t)stream;
  const int gin = m->cfg.gin_channels > 0 ? m->cfg.gin_channels : 1;
  if (!has_g(&m->cfg)) {
    WETTS_HIP_CHECK(hipMemsetAsync(g_out, 0, (size_t)B * gin * sizeof(float), s));
    return WETTS_OK;
  }
  WETTS_REQUIRE(sid != nullptr, "sid required when n_speakers > 0");
  return k_gather_rows(sid, m->emb_g, m->cfg.n_speakers, B, gin, g_out, m->status_word, s);
}

// ---------------------------------------------------------------------------------------------
namespace wetts {
// attentions.Encoder.forward (attentions.py:70-87) on xa [B,H,T] in place (x already masked):
// n x { x = LN1(x + MHA(x)); x = LN2(x + FFN(x)) }, final x * mask fused into the last LayerNorm.
// window < 0: no relative-position terms (window_size=None).  Scratch: q,k,v,att,y,xb [B,H,T],
// hid [B,F,T], sc [B,nh,T,T].
static int32_t run_enc_layers(const std::vector<EncLayer>& layers, float* xa, const float* x_mask,
                              int B, int H, int F, int nh, int window, int T, float* q, float* k,
                              float* v, float* att, float* y, float* hid, float* sc, float* xb,
                              hipStream_t s, const float* spk_cond = nullptr, int cond_idx = -1) {
  const int dk = H / nh, n = (int)layers.size();
  for (int l = 0; l < n; ++l) {
    const EncLayer& e = layers[l];
    const bool last = (l == n - 1);
    // speaker-conditioned encoder (attentions.py:74-78): x = (x + spk_emb_linear(g)) * x_mask
    if (spk_cond && l == cond_idx) WETTS_TRY(k_add_bias_b_mask(xa, spk_cond, x_mask, B, H, T, s));
    // q, k, v = one [B, 3H, T] projection (q / k / v scratch are contiguous: q is its base)
    WETTS_TRY(launch_conv(e.qkv, conv_io(xa, H, T, q, 3 * H, B), s));
    const float* qp = q;
    WETTS_TRY(k_rel_attention(qp, qp + (int64_t)H * T, qp + (int64_t)2 * H * T, (int64_t)3 * H * T,
                              x_mask, e.rel_k, e.rel_v, window, B, nh, dk, T, sc, att, s));
    (void)k; (void)v;
    WETTS_TRY(launch_conv(e.o, conv_io(att, H, T, y, H, B), s));
    // x = norm_layers_1(x + y), written MASKED: columns with mask 0 are zero from here on.  That is what the FFN wants
    // as its input (conv_1(pad(x * mask))), and nothing else can tell: the residual and LayerNorm below are per
    // column, masked columns never reach a valid one (attention masks its keys, the convs read zeros either way)
    // and the stack's output is masked.  The convs then see a plain input: LDS-DMA / 16-byte staging.
    WETTS_TRY(k_layernorm(xa, y, e.n1g, e.n1b, nullptr, x_mask, 0, B, H, T, xb, s));
    // FFN: conv_1(pad(x*mask)) -> relu -> conv_2(pad(.*mask)) * mask; the hidden tensor is masked by conv_1's epilogue
    {
      ConvParams p = conv_io(xb, H, T, hid, F, B);
      p.out_act = OUT_RELU;
      p.out_mask = x_mask;
      p.out_mask_stride = T;
      WETTS_TRY(launch_conv(e.f1, p, s));
    }
    {
      ConvParams p = conv_io(hid, F, T, y, H, B);
      p.out_mask = x_mask;
      p.out_mask_stride = T;
      WETTS_TRY(launch_conv(e.f2, p, s));
    }
    // x = norm_layers_2(x + y); the final `x = x * x_mask` is fused into the last layer
    WETTS_TRY(k_layernorm(xb, y, e.n2g, e.n2b, nullptr, last ? x_mask : nullptr, 0, B, H, T, xa, s));
  }
  return WETTS_OK;
}
}  // namespace wetts

int32_t wetts_text_encoder(const wetts_model_t* m, const int64_t* x, const int64_t* x_lengths,
                           const float* g, int32_t B, int32_t Tx, float* x_enc, float* stats,
                           float* x_mask, void* workspace, int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && x && x_lengths && x_enc && stats && x_mask, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  if (B == 0 || Tx == 0) return WETTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, F = c->filter_channels, I = c->inter_channels;
  const int nh = c->n_heads;
  Bump ws(workspace, workspace_bytes);
  float* q = ws.take<float>((int64_t)B * H * Tx);
  float* k = ws.take<float>((int64_t)B * H * Tx);
  float* v = ws.take<float>((int64_t)B * H * Tx);
  float* att = ws.take<float>((int64_t)B * H * Tx);
  float* y = ws.take<float>((int64_t)B * H * Tx);
  float* hid = ws.take<float>((int64_t)B * F * Tx);
  float* sc = ws.take<float>((int64_t)B * nh * Tx * Tx + (int64_t)B * H * Tx +
                             (int64_t)B * nh * (2 * c->window_size + 1) * Tx);  // + vT + rel table
  float* xb = ws.take<float>((int64_t)B * H * Tx);
  float* spk = ws.take<float>((int64_t)B * H);
  if (!ws.ok) {
    set_error("text_encoder: workspace too small (%lld bytes)", (long long)workspace_bytes);
    return WETTS_E_WORKSPACE;
  }
  const bool spk_on = c->use_spk_conditioned_encoder && has_g(c);
  if (spk_on) {
    WETTS_REQUIRE(g != nullptr, "speaker-conditioned encoder needs g");
    WETTS_TRY(k_cond_linear(g, m->enc_spk_w, m->enc_spk_b, B, H, c->gin_channels, spk, s));
  }
  // x = emb(x)*sqrt(H), masked (encoders.py:48-53; Encoder.forward x = x * x_mask, attentions.py:72)
  float* xa = x_enc;  // current activations live in xa
  WETTS_TRY(k_embed_mask(x, x_lengths, m->emb, c->n_vocab, B, H, Tx, xa, x_mask, m->status_word, s));
  WETTS_TRY(run_enc_layers(m->enc, xa, x_mask, B, H, F, nh, c->window_size, Tx, q, k, v, att, y,
                           hid, sc, xb, s, spk_on ? spk : nullptr, 2));
  if (c->n_layers == 0) {
    // Encoder with no layers still masks its input; embed_mask already did.
  }
  // stats = proj(x) * x_mask
  ConvParams p = conv_io(xa, H, Tx, stats, 2 * I, B);
  p.out_mask = x_mask;
  p.out_mask_stride = Tx;
  WETTS_TRY(launch_conv(m->enc_proj, p, s));
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
namespace wetts {
// DDSConv.forward (duration_predictors.py:45-57); `x + g` is the caller's.  *xout = where the result is: x itself
// (updated in place, layer by layer) or t1 (the one-launch kernel of dds_fused.hip, which cannot write in place).
static int32_t run_dds(const wetts_model* m, const DDS& d, float* x, const float* mask, int B, int C, int T, float* t1,
                       float* t2, hipStream_t s, float** xout) {
  *xout = x;
  if (m->dds_fused && dds_fused_supported(m->dds_fused, C, d.c1x1[0].M, d.c1x1[0].nchunks, B, T)) {
    DdsFusedParams p;
    memset(&p, 0, sizeof(p));
    p.x = x;
    p.out = t1;
    p.mask = mask;
    for (int i = 0; i < 3; ++i) {
      p.sep_w[i] = d.sep_w[i]; p.sep_b[i] = d.sep_b[i];
      p.n1g[i] = d.n1g[i]; p.n1b[i] = d.n1b[i];
      p.n2g[i] = d.n2g[i]; p.n2b[i] = d.n2b[i];
      p.wpk[i] = d.c1x1[i].wpk; p.bias[i] = d.c1x1[i].bias;
    }
    p.B = B; p.C = C; p.T = T;
    p.G = 2 * d.c1x1[0].nchunks;
    WETTS_TRY(k_dds_fused(p, s));
    *xout = t1;
    return WETTS_OK;
  }
  int dil = 1;
  for (int i = 0; i < 3; ++i) {
    WETTS_TRY(k_dwconv(x, mask, d.sep_w[i], d.sep_b[i], 3, dil, B, C, T, t1, s));
    WETTS_TRY(k_layernorm(t1, nullptr, d.n1g[i], d.n1b[i], nullptr, nullptr, 1, B, C, T, t2, s));
    WETTS_TRY(launch_conv(d.c1x1[i], conv_io(t2, C, T, t1, C, B), s));
    // x = x + gelu(norm_2(y)); the trailing `x * x_mask` is fused into the last layer
    WETTS_TRY(k_layernorm(t1, nullptr, d.n2g[i], d.n2b[i], x, i == 2 ? mask : nullptr, 1, B, C, T,
                          x, s));
    dil *= 3;
  }
  return WETTS_OK;
}
}  // namespace wetts

int32_t wetts_duration_sdp(const wetts_model_t* m, const float* x_enc, const float* x_mask,
                           const float* g, const float* eps_w, float noise_scale_w, int32_t B,
                           int32_t Tx, float* logw, int32_t* status_dev, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && x_enc && x_mask && eps_w && logw, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  WETTS_REQUIRE(m->cfg.use_sdp, "model was built with use_sdp=false");
  if (B == 0 || Tx == 0) return WETTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels;
  Bump ws(workspace, workspace_bytes);
  float* xd = ws.take<float>((int64_t)B * H * Tx);
  float* t1 = ws.take<float>((int64_t)B * H * Tx);
  float* t2 = ws.take<float>((int64_t)B * H * Tx);
  float* hh = ws.take<float>((int64_t)B * H * Tx);
  float* xg = ws.take<float>((int64_t)B * H * Tx);
  float* hp = ws.take<float>((int64_t)B * 32 * Tx);
  float* z = ws.take<float>((int64_t)B * 2 * Tx);
  float* cond = ws.take<float>((int64_t)B * H);
  if (!ws.ok) {
    set_error("duration_sdp: workspace too small");
    return WETTS_E_WORKSPACE;
  }
  if (!status_dev) status_dev = m->status_word;
  // x = pre(x) + cond(g)
  {
    ConvParams p = conv_io(x_enc, H, Tx, xd, H, B);
    if (has_g(c) && g) {
      WETTS_TRY(k_cond_linear(g, m->dp_cond_w, m->dp_cond_b, B, H, c->gin_channels, cond, s));
      p.bias_b = cond;
      p.bias_b_stride = H;
    }
    WETTS_TRY(launch_conv(m->sdp_pre, p, s));
  }
  float* xdo = xd;
  WETTS_TRY(run_dds(m, m->sdp_dds, xd, x_mask, B, H, Tx, t1, t2, s, &xdo));
  {
    ConvParams p = conv_io(xdo, H, Tx, xg, H, B);  // x = proj(x) * x_mask
    p.out_mask = x_mask;
    p.out_mask_stride = Tx;
    WETTS_TRY(launch_conv(m->sdp_proj, p, s));
  }
  // z = randn * noise_scale  (not masked, duration_predictors.py:257-258)
  WETTS_TRY(k_scale(eps_w, noise_scale_w, (int64_t)B * 2 * Tx, z, s));
  // reversed flows with the useless CF1 removed: [Flip, CFn, Flip, ..., CF2, Flip, EA]
  int swapped = 0;  // logical channel c lives at physical channel c ^ swapped
  for (int f = (int)m->cflows.size() - 1; f >= 0; --f) {
    swapped ^= 1;  // Flip
    const ConvFlowW& cf = m->cflows[f];
    const int ch0 = 0 ^ swapped, ch1 = 1 ^ swapped;
    // h = pre(x0); DDSConv(h, mask, g=x): h = h + g first
    WETTS_TRY(k_convflow_pre(z, ch0, cf.pre_w, cf.pre_b, xg, B, H, Tx, hh, s));
    float* hho = hh;
    WETTS_TRY(run_dds(m, cf.dds, hh, x_mask, B, H, Tx, t1, t2, s, &hho));
    {
      ConvParams p = conv_io(hho, H, Tx, hp, 29, B);  // h = proj(h) * x_mask
      p.out_mask = x_mask;
      p.out_mask_stride = Tx;
      WETTS_TRY(launch_conv(cf.proj, p, s));
    }
    WETTS_TRY(k_spline_inverse(z, ch0, ch1, hp, x_mask, 10, 5.0f, (float)sqrt((double)H), B, Tx,
                               status_dev, s));
  }
  swapped ^= 1;  // the Flip in front of ElementwiseAffine
  // EA reverse, then logw = z0
  WETTS_TRY(k_affine_reverse(z, 0 ^ swapped, m->ea_m, m->ea_logs, 0, x_mask, B, Tx, logw, s));
  return WETTS_OK;
}

int32_t wetts_duration_dp(const wetts_model_t* m, const float* x_enc, const float* x_mask,
                          const float* g, int32_t B, int32_t Tx, float* logw, void* workspace,
                          int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && x_enc && x_mask && logw, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  WETTS_REQUIRE(!m->cfg.use_sdp, "model was built with use_sdp=true");
  if (B == 0 || Tx == 0) return WETTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, Fd = c->dp_filter_channels;
  Bump ws(workspace, workspace_bytes);
  float* xc = ws.take<float>((int64_t)B * H * Tx);
  float* a = ws.take<float>((int64_t)B * Fd * Tx);
  float* b = ws.take<float>((int64_t)B * Fd * Tx);
  float* cond = ws.take<float>((int64_t)B * H);
  if (!ws.ok) {
    set_error("duration_dp: workspace too small");
    return WETTS_E_WORKSPACE;
  }
  const float* xin = x_enc;
  if (has_g(c) && g) {  // x = x + cond(g)
    WETTS_TRY(k_cond_linear(g, m->dp_cond_w, m->dp_cond_b, B, H, c->gin_channels, cond, s));
    WETTS_HIP_CHECK(hipMemcpyAsync(xc, x_enc, (size_t)B * H * Tx * 4, hipMemcpyDeviceToDevice, s));
    WETTS_TRY(k_add_bias_b(xc, cond, B, H, Tx, s));
    xin = xc;
  }
  {
    ConvParams p = conv_io(xin, H, Tx, a, Fd, B);  // relu(conv_1(x*mask))
    p.in_mask = x_mask;
    p.in_mask_stride = Tx;
    p.out_act = OUT_RELU;
    WETTS_TRY(launch_conv(m->dp_c1, p, s));
  }
  WETTS_TRY(k_layernorm(a, nullptr, m->dp_n1g, m->dp_n1b, nullptr, nullptr, 0, B, Fd, Tx, b, s));
  {
    ConvParams p = conv_io(b, Fd, Tx, a, Fd, B);
    p.in_mask = x_mask;
    p.in_mask_stride = Tx;
    p.out_act = OUT_RELU;
    WETTS_TRY(launch_conv(m->dp_c2, p, s));
  }
  WETTS_TRY(k_layernorm(a, nullptr, m->dp_n2g, m->dp_n2b, nullptr, nullptr, 0, B, Fd, Tx, b, s));
  {
    ConvParams p = conv_io(b, Fd, Tx, logw, 1, B);  // proj(x*mask) * mask
    p.in_mask = x_mask;
    p.in_mask_stride = Tx;
    p.out_mask = x_mask;
    p.out_mask_stride = Tx;
    WETTS_TRY(launch_conv(m->dp_proj, p, s));
  }
  return WETTS_OK;
}

int32_t wetts_durations_to_lengths(const float* logw, const float* x_mask, float length_scale,
                                   int32_t B, int32_t Tx, float* w_ceil, float* cum,
                                   int64_t* y_lengths, int32_t* status_dev, void* stream) {
  WETTS_REQUIRE(logw && x_mask && w_ceil && cum && y_lengths, "null argument");
  return k_durations_to_lengths(logw, x_mask, length_scale, B, Tx, w_ceil, cum, y_lengths,
                                status_dev, (hipStream_t)stream);
}

int32_t wetts_set_status_word(const wetts_model_t* m, int32_t* status_dev, void* stream) {
  WETTS_REQUIRE(m != nullptr, "null model");
  if (status_dev)
    WETTS_HIP_CHECK(hipMemsetAsync(status_dev, 0, sizeof(int32_t), (hipStream_t)stream));
  m->status_word = status_dev;
  return WETTS_OK;
}

int32_t wetts_set_seed(const wetts_model_t* m, uint64_t seed) {
  WETTS_REQUIRE(m != nullptr, "null model");
  m->rng_seed = seed;
  m->rng_offset = 0;
  return WETTS_OK;
}

int32_t wetts_randn(float* out, int64_t n, uint64_t seed, uint64_t offset, void* stream) {
  WETTS_REQUIRE(out != nullptr || n == 0, "null argument");
  return k_randn(out, n, seed, offset, (hipStream_t)stream);
}

int32_t wetts_mask_rows(const float* x, const float* mask, int32_t B, int32_t C, int32_t T,
                        float* out, void* stream) {
  WETTS_REQUIRE(x && mask && out, "null argument");
  return k_mask_rows(x, mask, B, C, T, out, (hipStream_t)stream);
}

int32_t wetts_length_regulate(const wetts_model_t* m, const float* stats, const float* cum,
                              const float* x_mask, const int64_t* y_lengths, const float* eps_z,
                              int64_t eps_batch_stride, int64_t eps_channel_stride,
                              float noise_scale, int32_t B, int32_t Tx, int32_t Ty,
                              int32_t* frame2phone, float* y_mask, float* attn, float* m_p_exp,
                              float* logs_p_exp, float* z_p, void* stream) {
  WETTS_REQUIRE(m && stats && cum && y_lengths && eps_z && frame2phone && y_mask && z_p,
                "null argument");
  (void)x_mask;
  hipStream_t s = (hipStream_t)stream;
  const int I = m->cfg.inter_channels;
  WETTS_TRY(k_frame_index(cum, y_lengths, B, Tx, Ty, frame2phone, y_mask, s));
  WETTS_TRY(k_expand_prior(stats, frame2phone, eps_z, eps_batch_stride, eps_channel_stride,
                           noise_scale, B, I, Tx, Ty, m_p_exp, logs_p_exp, z_p, s));
  if (attn) WETTS_TRY(k_attn_path(frame2phone, B, Tx, Ty, attn, s));
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
namespace wetts {
// 16-bit copies of the WN conv weights (in_layers k = 5, res_skip 1x1) of every coupling layer
// gate and residual / skip update run in the epilogues of the two convs of a layer when the convs take the
// 128-row, 64-channel-chunk tile those epilogues are instantiated for (conv_bf16.hip)
static bool wn16_fused(int H) { return H % 16 == 0 && H >= 128; }

static void free_flow_bf16(const wetts_model* m) {
  for (auto& v : m->b_wn_in) for (auto& pc : v) free_packed_bf16(&pc);
  for (auto& v : m->b_wn_rs) for (auto& pc : v) free_packed_bf16(&pc);
  m->b_wn_in.clear();
  m->b_wn_rs.clear();
  m->flow_packed_prec = 0;
}

static int32_t pack_flow_bf16_layers(const wetts_model* m, int f16, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, NL = c->flow_wn_layers, fk = c->flow_kernel_size;
  m->b_wn_in.assign(c->flow_n_flows, std::vector<PackedConvB>(NL));
  m->b_wn_rs.assign(c->flow_n_flows, std::vector<PackedConvB>(NL));
  for (int f = 0; f < c->flow_n_flows; ++f) {
    const std::string p = S("flow.flows.%d", flow_key_stride(c) * f);
    for (int i = 0; i < NL; ++i) {
      WETTS_TRY(pack_conv_weight_bf16(m->T(p + S(".enc.in_layers.%d.weight", i)),
                                      m->T(p + S(".enc.in_layers.%d.bias", i)), 2 * H, H, fk, 1,
                                      (fk - 1) / 2, 0, 0, f16, s, &m->b_wn_in[f][i],
                                      wn16_fused(H) ? H : 0));
      const int rs = (i < NL - 1) ? 2 * H : H;
      WETTS_TRY(pack_conv_weight_bf16(m->T(p + S(".enc.res_skip_layers.%d.weight", i)),
                                      m->T(p + S(".enc.res_skip_layers.%d.bias", i)), rs, H, 1, 1, 0,
                                      0, 0, f16, s, &m->b_wn_rs[f][i]));
    }
  }
  return WETTS_OK;
}

// 16-bit copies of the flow's WN weights for the precision in force.  All or nothing: a failure part-way
// (hipMalloc) frees what was built and leaves the model "not packed", so the next call retries cleanly.
static int32_t pack_flow_bf16(const wetts_model* m, hipStream_t s) {
  const int want = m->flow_precision;
  if (want == 0 || m->flow_packed_prec == want) return WETTS_OK;
  free_flow_bf16(m);
  const int32_t rc = pack_flow_bf16_layers(m, want == 2 ? 1 : 0, s);
  if (rc != WETTS_OK) {
    free_flow_bf16(m);
    return rc;
  }
  m->flow_packed_prec = want;
  return WETTS_OK;
}
}  // namespace wetts

int32_t wetts_set_flow_precision(const wetts_model_t* m, int32_t precision) {
  WETTS_REQUIRE(m != nullptr, "null model");
  WETTS_REQUIRE(precision >= 0 && precision <= 2, "precision must be 0 (f32), 1 (bf16) or 2 (f16)");
  WETTS_REQUIRE(precision == 0 || m->cfg.hidden_channels % 32 == 0,
                "the 16-bit flow needs hidden_channels to be a multiple of 32");
  m->flow_precision = precision;
  // packed here, not on first use: the stage calls never allocate (header contract)
  WETTS_TRY(pack_flow_bf16(m, nullptr));
  if (precision) WETTS_HIP_CHECK(hipStreamSynchronize(nullptr));
  return WETTS_OK;
}

int32_t wetts_flow_reverse(const wetts_model_t* m, const float* z_p_in, const float* y_mask_in,
                           const float* g, int32_t B, int32_t Ty_in, float* z_out_user, void* workspace,
                           int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && z_p_in && y_mask_in && z_out_user, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  if (B == 0 || Ty_in == 0) return WETTS_OK;
  hipStream_t s = (hipStream_t)stream;
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, I = c->inter_channels, NL = c->flow_wn_layers;
  Bump ws(workspace, workspace_bytes);
  // The flow works on rows of Ty = Ty_in rounded up to a multiple of 4 frames: every internal tensor then has
  // 16-byte rows (LDS-DMA GEMM for the 1x1 convs, 16-byte staging for the k = 5 convs).  The extra frames are
  // frames with mask 0 -- exactly what a shorter utterance's tail is in a padded batch: every masked tensor is zero
  // there and the un-masked ones (gate output, skip sum) only feed 1x1 convs or masked inputs.  When Ty_in is not a
  // multiple of 4, z_p and the mask are copied into padded rows first and z is copied out at the end.
  const int Ty = (Ty_in + 3) & ~3;
  const bool repad = Ty != Ty_in;
  float* zp_pad = ws.take<float>(repad ? (int64_t)B * I * Ty : 0);
  float* mask_pad = ws.take<float>(repad ? (int64_t)B * Ty : 0);
  float* xa = ws.take<float>((int64_t)B * I * Ty);
  float* xb = ws.take<float>((int64_t)B * I * Ty);
  float* h = ws.take<float>((int64_t)B * H * Ty);
  float* acts = ws.take<float>((int64_t)B * H * Ty);
  float* skip = ws.take<float>((int64_t)B * H * Ty);
  float* xin = ws.take<float>((int64_t)B * 2 * H * Ty);
  float* rs = ws.take<float>((int64_t)B * 2 * H * Ty);
  float* mm = ws.take<float>((int64_t)B * (I / 2) * Ty);
  float* gl = ws.take<float>((int64_t)B * 2 * H * NL);
  unsigned short* h16 = ws.take<unsigned short>((int64_t)B * H * Ty);
  unsigned short* acts16 = ws.take<unsigned short>((int64_t)B * H * Ty);
  unsigned short* xin16 = ws.take<unsigned short>((int64_t)B * 2 * H * Ty);
  unsigned short* rs16 = ws.take<unsigned short>((int64_t)B * 2 * H * Ty);
  float* skip_cl = ws.take<float>((int64_t)B * H * Ty);
  const int wn16 = m->flow_precision;  // 0: f32 WN, 1 / 2: bf16 / f16 convs and activations
  if (wn16) WETTS_TRY(pack_flow_bf16(m, s));
  float *tx0 = nullptr, *txm = nullptr, *tq = nullptr, *tk = nullptr, *tv = nullptr,
        *tatt = nullptr, *ty = nullptr, *thid = nullptr, *txb = nullptr, *tsc = nullptr;
  if (c->transformer_flows != 0) {
    // encoder width: x0 (I/2 channels) for pre_conv, the hidden h for pre_conv2
    const int64_t nh2 = (int64_t)B * (half_enc_flows(c) ? I / 2 : H) * Ty;
    tx0 = ws.take<float>(nh2); txm = ws.take<float>(nh2); tq = ws.take<float>(nh2);
    tk = ws.take<float>(nh2); tv = ws.take<float>(nh2); tatt = ws.take<float>(nh2);
    ty = ws.take<float>(nh2); thid = ws.take<float>(nh2); txb = ws.take<float>(nh2);
    const int He = half_enc_flows(c) ? I / 2 : H;
    tsc = ws.take<float>(attn_score_elems(half_enc_flows(c) ? -1 : 4, He / 2, B, 2, Ty) + nh2 +
                         (int64_t)B * 2 * 9 * Ty);  // scores (three-kernel path only) + vT + rel table
  }
  if (!ws.ok) {
    set_error("flow_reverse: workspace too small");
    return WETTS_E_WORKSPACE;
  }
  const float* z_p = z_p_in;
  const float* y_mask = y_mask_in;
  float* z_out = z_out_user;
  if (repad) {
    WETTS_TRY(k_copy_rows(z_p_in, Ty_in, Ty_in, zp_pad, Ty, Ty, (int64_t)B * I, s));
    WETTS_TRY(k_copy_rows(y_mask_in, Ty_in, Ty_in, mask_pad, Ty, Ty, B, s));
    z_p = zp_pad;
    y_mask = mask_pad;
  }
  const float* cur = z_p;
  for (int f = c->flow_n_flows - 1; f >= 0; --f) {
    const FlowW& fw = m->flows[f];
    if (mono_flows(c)) {
      // MonoTransformerFlowLayer.forward(reverse=True) ahead of this flow's Flip + coupling layer (reversed
      // [RCL, Flip, Mono] list, flows.py:391-425,444-446), channels in natural order: x0 = cur[:, :I/2].
      //   inter (3, flows.py:302-324): h = pre_transformer(x0 * mask, mask) + x0;  m = post(h) * mask;
      //                                out = [x0, (x1 - m) * mask]
      //   post  (4, flows.py:287-300): x0 = x0 / 2;  m = post(pre_transformer(x0, mask)) * mask  (the Encoder masks its
      //                                input itself, attentions.py:72);  out = [x0, (x1 - m) / (1 + exp(-0)) * mask]
      const int Hh = I / 2;
      const float sc = c->transformer_flows == 4 ? 0.5f : 1.f;
      float* mdst = (cur == xa) ? xb : xa;
      WETTS_TRY(k_mono_split(cur, y_mask, B, I, Ty, sc, tx0, txm, s));  // x0 * sc, and x0 * sc * mask
      WETTS_TRY(run_enc_layers(fw.mono_tr, txm, y_mask, B, Hh, Hh, 2, -1, Ty, tq, tk, tv, tatt, ty, thid, tsc, txb, s));
      if (c->transformer_flows == 3) WETTS_TRY(k_add(txm, tx0, (int64_t)B * Hh * Ty, txm, s));
      ConvParams p = conv_io(txm, Hh, Ty, mm, Hh, B);
      p.out_mask = y_mask;
      p.out_mask_stride = Ty;
      WETTS_TRY(launch_conv(fw.mono_post, p, s));
      WETTS_TRY(k_mono_coupling(cur, mm, y_mask, B, I, Ty, sc, mdst, s));
      cur = mdst;
    }
    float* dst = (f == 0 && !repad) ? z_out : ((cur == xa) ? xb : xa);
    // Flip then ResidualCouplingLayer(reverse): x0 = flipped[:I/2] = cur[I-1 .. I/2]
    if (c->transformer_flows == 1) {
      // VITS2 "pre_conv" (flows.py:145-150): x0_ = pre_transformer(x0 * mask, mask) + x0,
      // h = pre(x0_) * mask.  x0 is materialised (channel-reversed half) because the encoder
      // layers work in place.
      const int Hh = I / 2;
      WETTS_TRY(k_flip_half(cur, y_mask, B, I, Ty, tx0, txm, s));  // raw x0, and x0 * mask
      WETTS_TRY(run_enc_layers(fw.pre_tr, txm, y_mask, B, Hh, Hh, 2, -1, Ty, tq, tk, tv, tatt, ty,
                               thid, tsc, txb, s));
      WETTS_TRY(k_add(txm, tx0, (int64_t)B * Hh * Ty, txm, s));  // vits2 residual connection
      ConvParams p = conv_io(txm, Hh, Ty, h, H, B);               // h = pre(x0_) * mask
      p.out_mask = y_mask;
      p.out_mask_stride = Ty;
      WETTS_TRY(launch_conv(fw.pre, p, s));
    } else {
      ConvParams p = conv_io(cur + (int64_t)(I / 2) * Ty, I / 2, Ty, h, H, B);  // h = pre(x0) * mask
      p.x_bs = (int64_t)I * Ty;  // (x0 = the upper half of the channels, reversed inside the weights)
      p.out_mask = y_mask;
      p.out_mask_stride = Ty;
      WETTS_TRY(launch_conv(fw.pre, p, s));
      if (c->transformer_flows == 2) {
        // "pre_conv2" (flows.py:64-67): h = h + pre_transformer(h * mask, mask)
        WETTS_HIP_CHECK(hipMemcpyAsync(txm, h, (size_t)B * H * Ty * sizeof(float),
                                       hipMemcpyDeviceToDevice, s));
        WETTS_TRY(run_enc_layers(fw.pre_tr, txm, y_mask, B, H, H, 2, 4, Ty, tq, tk, tv, tatt, ty,
                                 thid, tsc, txb, s));
        WETTS_TRY(k_add(h, txm, (int64_t)B * H * Ty, h, s));
      }
    }
    const bool use_g = has_g(c) && g;
    if (use_g)
      WETTS_TRY(k_cond_linear(g, fw.cond_w, fw.cond_b, B, 2 * H * NL, c->gin_channels, gl, s));
    if (wn16) {
      // WN at 16 bit (wn16.hip): channel-last activations, f32 accumulation, the skip sum in f32
      const int f16 = wn16 == 2 ? 1 : 0;
      const int64_t rows = (int64_t)B * Ty;
      WETTS_TRY(k_cf32_to_cl16(h, h16, B, H, Ty, f16, s));
      for (int i = 0; i < NL; ++i) {
        const bool last = (i == NL - 1);
        ConvBParams p1;
        memset(&p1, 0, sizeof(p1));
        p1.x = h16; p1.x_bs = (int64_t)H * Ty; p1.Cin = H; p1.Tin = Ty; p1.in_act = IN_NONE;
        p1.out = xin16; p1.o_bs = (int64_t)2 * H * Ty; p1.cout = 2 * H; p1.Tout = Ty;
        p1.out_div = 1.f; p1.B = B;
        if (use_g) {
          p1.bias_b = gl + (int64_t)i * 2 * H;
          p1.bias_b_stride = (int64_t)2 * H * NL;
        }
        const bool fusedwn = wn16_fused(H);
        if (fusedwn) {  // gate in the epilogue: acts16 written directly
          p1.epi_mode = 1;
          p1.wn_H = H;
          p1.out = acts16;
          p1.o_bs = (int64_t)H * Ty;
        }
        WETTS_TRY(launch_conv_bf16(m->b_wn_in[f][i], p1, s));
        if (!fusedwn) WETTS_TRY(k_gate_cl16(xin16, acts16, rows, H, f16, s));
        const int RC = last ? H : 2 * H;
        ConvBParams p2;
        memset(&p2, 0, sizeof(p2));
        p2.x = acts16; p2.x_bs = (int64_t)H * Ty; p2.Cin = H; p2.Tin = Ty; p2.in_act = IN_NONE;
        p2.out = rs16; p2.o_bs = (int64_t)RC * Ty; p2.cout = RC; p2.Tout = Ty;
        p2.out_div = 1.f; p2.B = B;
        if (fusedwn) {  // h / skip update in the epilogue: rs16 is never written
          p2.epi_mode = 2;
          p2.wn_H = H;
          p2.wn_h = h16;
          p2.wn_skip = skip_cl;
          p2.wn_mask = y_mask;
          p2.wn_last = last ? 1 : 0;
          p2.wn_first = i == 0 ? 1 : 0;
        }
        WETTS_TRY(launch_conv_bf16(m->b_wn_rs[f][i], p2, s));
        if (!fusedwn)
          WETTS_TRY(k_wn_update_cl16(rs16, h16, skip_cl, y_mask, last ? 1 : 0, i == 0 ? 1 : 0, rows, H, f16, s));
      }
      WETTS_TRY(k_cl32_to_cf32(skip_cl, skip, B, H, Ty, s));
    } else
    for (int i = 0; i < NL; ++i) {
      {
        // acts = tanh(a[:H]) * sigmoid(a[H:]),  a = in_layer(h) (+ g_l)  (modules.py:71-78, commons.py:98-105): the gate
        // runs in the conv's epilogue (rows packed interleaved), x_in is never written
        const bool gate = fw.in_layers[i].gate_H > 0;
        ConvParams p = conv_io(h, H, Ty, gate ? acts : xin, gate ? H : 2 * H, B);
        if (gate) p.out_act = OUT_GATE;
        if (use_g) {
          p.bias_b = gl + (int64_t)i * 2 * H;
          p.bias_b_stride = (int64_t)2 * H * NL;
        }
        WETTS_TRY(launch_conv(fw.in_layers[i], p, s));
        if (!gate) WETTS_TRY(k_gate(xin, B, H, Ty, acts, s));
      }
      const bool last = (i == NL - 1);
      ConvParams p2 = conv_io(acts, H, Ty, rs, last ? H : 2 * H, B);
      // residual / skip update in the conv's epilogue (rs never exists, one launch less per layer) where launches are
      // the cost: the calls the small-launch conv kernel takes.  Big batches keep the specialised conv epilogue and
      // the separate update -- the generic epilogue costs them 0.2 % of the headline step (profiles/r03_wn_fuse_ab.txt)
      // ... and launches the LDS-DMA GEMM takes (gemm_pw.hip: h / skip rows come in through the accumulator init, so
      // its epilogue stays a plain store; rows of the flow are 16-byte aligned here, see the top of this function)
      const bool fuse_upd = m->wn_fuse == 2 ||
                            (m->wn_fuse == 1 && ((int64_t)cdiv(last ? H : 2 * H, 64) * cdiv(Ty, 64) * B <= m->small_max_tiles ||
                                                 (H % 32) == 0));
      if (fuse_upd) {
        p2.wn_h = h;
        p2.wn_skip = skip;
        p2.wn_mask = y_mask;
        p2.wn_mask_stride = Ty;
        p2.wn_H = H;
        p2.wn_last = last ? 1 : 0;
        p2.wn_first = i == 0 ? 1 : 0;
      }
      WETTS_TRY(launch_conv(fw.res_skip[i], p2, s));
      if (!fuse_upd) WETTS_TRY(k_wn_update(rs, h, skip, y_mask, last ? 1 : 0, i == 0 ? 1 : 0, B, H, Ty, s));
    }
    {
      ConvParams p = conv_io(skip, H, Ty, mm, I / 2, B);  // m = post(output*mask) * mask
      p.in_mask = y_mask;
      p.in_mask_stride = Ty;
      p.out_mask = y_mask;
      p.out_mask_stride = Ty;
      WETTS_TRY(launch_conv(fw.post, p, s));
    }
    WETTS_TRY(k_coupling_flip(cur, mm, y_mask, B, I, Ty, dst, s));
    cur = dst;
  }
  if (repad) WETTS_TRY(k_copy_rows(cur, Ty, Ty_in, z_out_user, Ty_in, Ty_in, (int64_t)B * I, s));
  return WETTS_OK;
}

// ---------------------------------------------------------------------------------------------
namespace wetts {
struct DecTiming {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double mrf_ms = 0;
  int launches = 0;
  bool on = false;
};

// one MRF launch group for the live counters: `n` kernels, `planes` [B][ch][len] tensors of `esz`-byte elements moved once
static inline void mrf_count(const wetts_model* m, int n, double planes, int ch, int64_t len, int B, int esz) {
  if (!m->mrf_timing) return;
  m->mrf_launches += n;
  m->mrf_bytes += planes * (double)ch * (double)len * (double)B * (double)esz;
}

// VocosGenerator.forward (decoders.py:286-305) on (z * y_mask)[:, :, :L]:
//   pad -> in_conv (+ cond(g)) -> LN -> 8 x ConvNeXt -> LN -> out_conv -> exp/clamp, cos/sin
//   -> iSTFT (n_fft, hop, hann, center) = windowed inverse-rDFT GEMM + overlap-add / envelope
static int32_t run_vocos(const wetts_model* m, const float* z, int64_t z_bs, int64_t z_cs,
                         const float* y_mask, int64_t mask_stride, const float* g, int B, int L,
                         float* audio, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, VC = c->vocos_channels, VH = c->vocos_h_channels;
  const int NF = c->istft_n_fft, VO = NF + 2, F = L + 1;
  const int Fs = (int)vocos_fs(L), VOp = (int)vocos_rows(c);  // padded row stride / padded iSTFT reduction
  // nn.ReflectionPad1d([1, 0]) needs at least two frames (PyTorch raises otherwise)
  WETTS_REQUIRE(L >= 2, "vocos decoder needs at least 2 frames (ReflectionPad1d([1,0]), decoders.py:265)");
  Bump ws(workspace, workspace_bytes);
  float* xpad = ws.take<float>((int64_t)B * I * Fs);
  float* h = ws.take<float>((int64_t)B * VC * Fs);
  float* t1 = ws.take<float>((int64_t)B * VC * Fs);
  float* t2 = ws.take<float>((int64_t)B * VC * Fs);
  float* u = ws.take<float>((int64_t)B * VH * Fs);
  float* spec = ws.take<float>((int64_t)B * VO * Fs);
  float* ri = ws.take<float>((int64_t)B * VOp * Fs);
  float* frames = ws.take<float>((int64_t)B * NF * Fs);
  float* cond = ws.take<float>((int64_t)B * VC);
  if (!ws.ok) {
    set_error("vocos: workspace too small (need %lld bytes)", (long long)ws_decoder(c, B, L));
    return WETTS_E_WORKSPACE;
  }
  // Every tensor below is [B][C][Fs]: Fs - F (< 4) junk columns per row ride along.  All ops are column-wise
  // except the depthwise conv, which takes the valid length, so junk never reaches a valid column; it stays finite
  // (zero-filled here, LayerNorm / GELU / GEMMs of finite values after that).
  WETTS_TRY(k_vocos_pad(z, z_bs, z_cs, y_mask, mask_stride, B, I, L, xpad, s, Fs));
  {
    ConvParams p = conv_io(xpad, I, Fs, t1, VC, B);
    if (has_g(c) && g) {
      WETTS_TRY(k_cond_linear(g, m->dec_cond_w, m->dec_cond_b, B, VC, c->gin_channels, cond, s));
      p.bias_b = cond;
      p.bias_b_stride = VC;
    }
    WETTS_TRY(launch_conv(m->v_in, p, s));
  }
  WETTS_TRY(k_layernorm(t1, nullptr, m->v_npre_g, m->v_npre_b, nullptr, nullptr, 0, B, VC, Fs, h, s));
  // live timing of the dominant class (the ConvNeXt stack = 99 % of the flops), same mechanism as
  // the HiFi-GAN MRF events
  hipEvent_t lv0 = nullptr, lv1 = nullptr;
  if (m->mrf_timing) {
    WETTS_HIP_CHECK(hipEventCreate(&lv0));
    WETTS_HIP_CHECK(hipEventCreate(&lv1));
    WETTS_HIP_CHECK(hipEventRecord(lv0, s));
  }
  for (const ConvNeXt& cn : m->v_layers) {
    int32_t rc = WETTS_OK;
    if (!k_convnext_dwln(h, cn.dw_w, cn.dw_b, cn.ng, cn.nb, B, VC, Fs, F, t2, s, &rc)) {
      WETTS_TRY(k_dwconv(h, nullptr, cn.dw_w, cn.dw_b, 3, 1, B, VC, Fs, t1, s, F));
      WETTS_TRY(k_layernorm(t1, nullptr, cn.ng, cn.nb, nullptr, nullptr, 0, B, VC, Fs, t2, s));
    }
    WETTS_TRY(rc);
    ConvParams p1 = conv_io(t2, VC, Fs, u, VH, B);
    p1.out_act = OUT_GELU;
    p1.tag = 1;  // the Vocos models' dominant class: own kernel symbol for the profiles
    WETTS_TRY(launch_conv(cn.pw1, p1, s));
    ConvParams p2 = conv_io(u, VH, Fs, t1, VC, B);  // x = res + scale * pw2(u)
    p2.tag = 1;
    p2.res = h;
    p2.r_bs = (int64_t)VC * Fs;
    p2.r_cs = Fs;
    WETTS_TRY(launch_conv(cn.pw2, p2, s));
    float* sw = h; h = t1; t1 = sw;
  }
  if (m->mrf_timing) {
    WETTS_HIP_CHECK(hipEventRecord(lv1, s));
    m->mrf_events.emplace_back(lv0, lv1);
    // the two pointwise GEMMs per layer: pw1 reads t2 [VC], writes u [VH]; pw2 reads u [VH] + the residual h [VC], writes [VC]
    mrf_count(m, 2 * (int)m->v_layers.size(), (double)m->v_layers.size() * (3.0 * VC + 2.0 * VH), 1, Fs, B, 4);
    m->mrf_calls += 1;
  }
  WETTS_TRY(k_layernorm(h, nullptr, m->v_npost_g, m->v_npost_b, nullptr, nullptr, 0, B, VC, Fs, t2, s));
  WETTS_TRY(launch_conv(m->v_out, conv_io(t2, VC, Fs, spec, VO, B), s));
  // rows VO .. VOp - 1 of every item: zeros against the zero tail of the packed basis (k_rows_padded)
  if (VOp > VO)
    WETTS_HIP_CHECK(hipMemset2DAsync(ri + (int64_t)VO * Fs, (size_t)VOp * Fs * sizeof(float), 0,
                                     (size_t)(VOp - VO) * Fs * sizeof(float), (size_t)B, s));
  WETTS_TRY(k_vocos_spec(spec, B, VO / 2, Fs, ri, s, (int64_t)VOp * Fs));
  {
    ConvParams p = conv_io(ri, VO, Fs, frames, NF, B);
    p.x_bs = (int64_t)VOp * Fs;
    p.k_rows_padded = 1;
    WETTS_TRY(launch_conv(m->istft_mode == WETTS_ISTFT_ONNX ? m->v_istft_onnx : m->v_istft, p, s));
  }
  return k_istft_ola(frames, B, NF, c->istft_hop_length, F, audio, s, Fs, m->istft_mode == WETTS_ISTFT_ONNX ? 0 : 1);
}

// output columns per block of the fused ResBlock pair kernels (resblock32.hip / resblock16.hip)
static int pair_nto(int C, int ktaps) {
  const int wm = C >= 128 ? 4 : (C >= 64 ? 2 : 1);  // m-blocks of 32 rows (C in {32, 64, 128})
  const int nto = 128 * (4 / wm) - (ktaps - 1);
  return nto > 0 ? nto : 1;
}

// One stage's ResBlock1 sum  xsum = (sum_j ResBlock1_j(xu)) / nk  at f32 (decoders.py:157-170,72-77), dilation-major:
// the k = 3 / 7 / 11 chains of a stage are independent until the final sum, so the convs that run as single launches
// at step d of every chain go out TOGETHER (launch_conv_group: one grid, one tail instead of three -- at the C = 256
// stage a launch is 1.7 rounds of blocks and 15 % of its time is the half-empty last round).  Per output element the
// arithmetic and its order are those of the chain-by-chain loop in run_hifigan (the running sum is still added in
// chain order j = 0, 1, 2 by the last launch of each chain), so the results are bit-identical to it.
static int32_t run_stage_rb1_grouped(const wetts_model* m, int i, const float* xu, float* xsum,
                                     float* (*chain_buf)[3], int ch, int len, int spf, int B, const int64_t* lens,
                                     DecTiming* tm, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  const bool chain_addr_ok = (int64_t)ch * len * 4 < (int64_t)INT32_MAX;
  auto count = [&](int n, double planes) {
    if (tm && tm->on) tm->launches += n;
    mrf_count(m, n, planes, ch, len, B, 4);
  };
  // which chains run as ONE launch (the whole ResBlock1 on the chain kernel)
  bool whole[WETTS_MAX_RB_KERNELS];
  const float* rx[WETTS_MAX_RB_KERNELS];
  for (int j = 0; j < nk; ++j) {
    const RB& rb = m->rbs[i * nk + j];
    rx[j] = xu;
    whole[j] = false;
    if (!m->dec_unfused && nd <= RESCHAIN32_MAX_PAIRS && ch <= m->chain_whole_maxc && m->chain_whole_waste_pct > 0 &&
        len % 4 == 0 && chain_addr_ok &&
        resblock_chain32_supported(rb.c1.data(), rb.c2.data(), nd, m->fuse32_lds / 2, m->chain_whole_waste_pct)) {
      int dl[RESCHAIN32_MAX_PAIRS];
      for (int d = 0; d < nd; ++d) dl[d] = rb.c1[d].dil;
      whole[j] = cdiv(len, resblock_chain32_nto(ch, rb.c1[0].ktaps, dl, nd)) * B >= m->fuse_min_blocks;
    }
  }
  enum Kind { K_CHAIN1, K_FUSE32, K_SINGLES };
  for (int d = 0; d < nd; ++d) {
    const bool last_d = d == nd - 1;
    Kind kind[WETTS_MAX_RB_KERNELS];
    float* outp[WETTS_MAX_RB_KERNELS];
    const PackedConv* g1[WETTS_MAX_RB_KERNELS];
    ConvParams p1s[WETTS_MAX_RB_KERNELS];
    int n1 = 0;
    for (int j = 0; j < nk; ++j) {
      if (whole[j]) continue;
      const RB& rb = m->rbs[i * nk + j];
      float *fa = chain_buf[j][0], *fb = chain_buf[j][1], *ft = chain_buf[j][2];
      outp[j] = last_d ? xsum : ((rx[j] == fa) ? fb : fa);
      const int pair_tiles = cdiv(len, pair_nto(ch, rb.c1[d].ktaps)) * B;
      const bool chain1 = !m->dec_unfused && len % 4 == 0 && chain_addr_ok &&
                          (ch <= m->chain_pair_maxc || rb.c1[d].ktaps <= m->chain_pair_kmax) &&
                          resblock_chain32_supported(&rb.c1[d], &rb.c2[d], 1, m->fuse32_lds / 2, 100) &&
                          pair_tiles >= m->fuse_min_blocks;
      const bool fuse32 = !chain1 && !m->dec_unfused && !lens &&
                          resblock_pair32_supported(rb.c1[d], rb.c2[d], m->fuse32_lds) &&
                          !(ch >= 128 && rb.c1[d].ktaps >= m->fuse32_kmax128) &&
                          (ch <= m->fuse32_maxc || rb.c1[d].ktaps <= m->fuse32_kwide) &&
                          rb.c1[d].ktaps <= m->fuse32_kmax && pair_tiles >= m->fuse_min_blocks;
      kind[j] = chain1 ? K_CHAIN1 : fuse32 ? K_FUSE32 : K_SINGLES;
      if (kind[j] == K_SINGLES) {  // xt = c1(lrelu(x)): every chain's c1 of this step in one launch
        ConvParams p = conv_io(rx[j], ch, len, ft, ch, B);
        p.in_act = IN_LRELU;
        p.in_slope = 0.1f;
        p.tag = 1;
        p.lens = lens;
        p.len_mul = spf;
        g1[n1] = &rb.c1[d];
        p1s[n1++] = p;
      }
    }
    if (n1 > 0) {
      if (m->dec_unfused) {  // the diagnostic form: every conv its own launch
        for (int q = 0; q < n1; ++q) WETTS_TRY(launch_conv(*g1[q], p1s[q], s));
        count(n1, 2.0 * n1);
      } else {
        int nl = 0;
        WETTS_TRY(launch_conv_group(g1, p1s, n1, s, &nl));
        count(nl, (d == 0 ? 1 : n1) + n1);  // step 0: every chain's c1 reads the one upsampled x
      }
    }
    // the second half of the step.  Not the last dilation: the chains are still independent -- fused pairs go out as
    // they are, the single c2 convs as one group.  Last dilation: every chain's last launch adds the running sum, so
    // those run one after the other in chain order (whole-chain launches take their place in that order).
    const PackedConv* g2[WETTS_MAX_RB_KERNELS];
    ConvParams p2s[WETTS_MAX_RB_KERNELS];
    int n2 = 0;
    for (int j = 0; j < nk; ++j) {
      const RB& rb = m->rbs[i * nk + j];
      const int accum = (last_d && j > 0) ? 1 : 0;
      const float odiv = (last_d && j == nk - 1) ? (float)nk : 1.f;  // x = xs / self.num_kernels
      if (whole[j]) {
        if (!last_d) continue;
        ResChain32Params cp;
        memset(&cp, 0, sizeof(cp));
        cp.x = xu;
        cp.out = xsum;
        cp.T = len;
        cp.B = B;
        cp.accum = (j > 0) ? 1 : 0;
        cp.out_div = odiv;
        cp.slope = 0.1f;
        cp.lens = lens;
        cp.len_mul = spf;
        WETTS_TRY(launch_resblock_chain32(rb.c1.data(), rb.c2.data(), nd, cp, s));
        count(1, 2 + (j > 0 ? 1 : 0));
        continue;
      }
      if (kind[j] == K_CHAIN1) {
        ResChain32Params cp;
        memset(&cp, 0, sizeof(cp));
        cp.x = rx[j];
        cp.out = outp[j];
        cp.T = len;
        cp.B = B;
        cp.accum = accum;
        cp.out_div = odiv;
        cp.slope = 0.1f;
        cp.lens = lens;
        cp.len_mul = spf;
        WETTS_TRY(launch_resblock_chain32(&rb.c1[d], &rb.c2[d], 1, cp, s));
        count(1, 2 + accum);
      } else if (kind[j] == K_FUSE32) {
        ResPair32Params pp;
        memset(&pp, 0, sizeof(pp));
        pp.x = rx[j];
        pp.out = outp[j];
        pp.T = len;
        pp.B = B;
        pp.accum = accum;
        pp.out_div = odiv;
        pp.slope = 0.1f;
        WETTS_TRY(launch_resblock_pair32(rb.c1[d], rb.c2[d], pp, s));
        count(1, 2 + accum);
      } else {  // x = c2(lrelu(xt)) + x
        ConvParams p2 = conv_io(chain_buf[j][2], ch, len, outp[j], ch, B);
        p2.in_act = IN_LRELU;
        p2.in_slope = 0.1f;
        p2.res = rx[j];
        p2.r_bs = (int64_t)ch * len;
        p2.r_cs = len;
        p2.accum = accum;
        p2.out_div = odiv;
        p2.tag = 1;
        p2.lens = lens;
        p2.len_mul = spf;
        if (last_d || m->dec_unfused) {
          WETTS_TRY(launch_conv(rb.c2[d], p2, s));
          count(1, 3 + accum);
        } else {
          g2[n2] = &rb.c2[d];
          p2s[n2++] = p2;
        }
      }
      rx[j] = outp[j];
    }
    if (n2 > 0) {
      int nl = 0;
      WETTS_TRY(launch_conv_group(g2, p2s, n2, s, &nl));
      count(nl, 3.0 * n2);
    }
  }
  return WETTS_OK;
}

// lens != null: ragged batch (wetts_hifigan_ragged) -- utterance b is decoded over its own lens[b] frames, as if
// it were alone in the call; ResBlock1 models at f32 only (the caller checks)
static int32_t run_hifigan(const wetts_model* m, const float* z, int64_t z_bs, int64_t z_cs,
                           const float* y_mask, int64_t mask_stride, const float* g, int B, int L,
                           float* audio, void* workspace, int64_t workspace_bytes, hipStream_t s,
                           DecTiming* tm, const int64_t* lens = nullptr) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, C0 = c->upsample_initial_channel;
  const int64_t mx = dec_max_elems(c, B, L);
  Bump ws(workspace, workspace_bytes);
  float* bx = ws.take<float>(mx);
  float* bt = ws.take<float>(mx);
  float* bs = ws.take<float>(mx);
  float* chain_buf[WETTS_MAX_RB_KERNELS][3];
  for (int j = 0; j < c->n_resblock_kernels; ++j)
    for (int q = 0; q < 3; ++q) chain_buf[j][q] = ws.take<float>(mx);
  float* cond = ws.take<float>((int64_t)B * C0);
  if (!ws.ok) {
    set_error("hifigan: workspace too small (need %lld bytes)",
              (long long)ws_decoder(c, B, L));
    return WETTS_E_WORKSPACE;
  }
  // x = conv_pre(z [* y_mask]) + cond(g)
  {
    ConvParams p = conv_io(z, I, L, bx, C0, B);
    p.x_bs = z_bs;
    p.x_cs = z_cs;
    if (y_mask) {
      p.in_mask = y_mask;
      p.in_mask_stride = mask_stride;
    }
    if (has_g(c) && g) {
      WETTS_TRY(k_cond_linear(g, m->dec_cond_w, m->dec_cond_b, B, C0, c->gin_channels, cond, s));
      p.bias_b = cond;
      p.bias_b_stride = C0;
    }
    p.lens = lens;
    p.len_mul = 1;
    WETTS_TRY(launch_conv(m->conv_pre, p, s));
  }
  int ch = C0, len = L;
  int spf = 1;  // samples per input frame at the current stage (ragged batches: lens[b] * spf samples)
  float* x = bx;   // stage input / MRF output
  float* xs = bs;  // MRF accumulator
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  for (int i = 0; i < c->n_upsamples; ++i) {
    const int u = c->upsample_rates[i];
    // x = ups[i](leaky_relu(x, 0.1))
    {
      ConvParams p = conv_io(x, ch, len, bt, ch / 2, B);
      p.in_act = IN_LRELU;
      p.in_slope = 0.1f;
      p.Tout = len * u;  // (len-1)*u - 2*pad + k with pad=(k-u)/2
      p.o_bs = (int64_t)(ch / 2) * len * u;
      p.o_cs = (int64_t)len * u;
      p.lens = lens;
      p.len_mul = spf;
      WETTS_TRY(launch_conv(m->ups[i], p, s));
    }
    ch /= 2;
    len *= u;
    spf *= u;
    float* xu = bt;                     // upsampled x, input of every resblock of this stage
    float* xsum = (x == bx) ? bs : bx;  // MRF accumulator for this stage
    (void)xs;
    if (tm && tm->on) WETTS_HIP_CHECK(hipEventRecord(tm->e0, s));
    hipEvent_t lv0 = nullptr, lv1 = nullptr;
    if (m->mrf_timing) {
      WETTS_HIP_CHECK(hipEventCreate(&lv0));
      WETTS_HIP_CHECK(hipEventCreate(&lv1));
      WETTS_HIP_CHECK(hipEventRecord(lv0, s));
    }
    // streams for the k = 3 / 7 / 11 chains of this stage: mrf_streams (opt-in, any size), or -- small_fork -- all
    // chains concurrently when the stage's convs are launches of a few blocks (a streaming window): three
    // independent 10-25 us kernels then share the chip instead of queueing behind each other
    const bool small_stage = !lens && (int64_t)cdiv(ch, 64) * cdiv(len, 64) * B <= m->small_max_tiles;
    const int nstreams = !m->fork_ok ? 1 : (c->resblock == 1 && ((m->small_fork && small_stage) || (ch <= m->mrf_fork_maxc && !m->dec_serial))) ? nk : m->mrf_streams;
    const bool forked = nstreams > 1;
    if (forked) WETTS_HIP_CHECK(hipEventRecord(m->ev_fork, s));
    // the chain kernel addresses one utterance's [C][T] plane with 32-bit byte offsets and buffer descriptors: a
    // plane of 2 GiB or more (one utterance above ~16.7 M samples at C = 32) takes the conv-by-conv path instead
    const bool chain_addr_ok = (int64_t)ch * len * 4 < (int64_t)INT32_MAX;
    const bool grouped = c->resblock == 1 && !forked && m->conv_groups;
    if (grouped) WETTS_TRY(run_stage_rb1_grouped(m, i, xu, xsum, chain_buf, ch, len, spf, B, lens, tm, s));
    for (int j = 0; j < (grouped ? 0 : nk); ++j) {
      const RB& rb = m->rbs[i * nk + j];
      hipStream_t sj = (forked && j > 0 && j < nstreams) ? m->aux_stream[j] : s;
      if (sj != s) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_fork, 0));
      float* fa = chain_buf[j][0];
      float* fb = chain_buf[j][1];
      float* ft = chain_buf[j][2];
      const float* rx = xu;  // current resblock x
      // a whole ResBlock1 in one launch (resblock_chain32.hip): x read once, the MRF sum written once
      if (c->resblock == 1 && !m->dec_unfused && nd <= RESCHAIN32_MAX_PAIRS &&
          ch <= m->chain_whole_maxc && m->chain_whole_waste_pct > 0 && len % 4 == 0 && chain_addr_ok &&
          resblock_chain32_supported(rb.c1.data(), rb.c2.data(), nd, m->fuse32_lds / 2,
                                     m->chain_whole_waste_pct)) {
        int dl[RESCHAIN32_MAX_PAIRS];
        for (int d = 0; d < nd; ++d) dl[d] = rb.c1[d].dil;
        if (cdiv(len, resblock_chain32_nto(ch, rb.c1[0].ktaps, dl, nd)) * B >= m->fuse_min_blocks) {
          ResChain32Params cp;
          memset(&cp, 0, sizeof(cp));
          cp.x = rx;
          cp.out = xsum;
          cp.T = len;
          cp.B = B;
          cp.accum = (j > 0) ? 1 : 0;
          cp.out_div = (j == nk - 1) ? (float)nk : 1.f;  // x = xs / self.num_kernels
          cp.slope = 0.1f;
          cp.lens = lens;
          cp.len_mul = spf;
          // (forked: the launch adds the running sum, which is ordered chain j - 1 -> chain j)
          if (forked && j > 0) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          WETTS_TRY(launch_resblock_chain32(rb.c1.data(), rb.c2.data(), nd, cp, sj));
          if (forked) WETTS_HIP_CHECK(hipEventRecord(m->ev_chain[j], sj));
          if (tm && tm->on) tm->launches += 1;
          mrf_count(m, 1, 2 + (j > 0 ? 1 : 0), ch, len, B, 4);
          continue;
        }
      }
      for (int d = 0; d < nd; ++d) {
        const bool last_d = (d == nd - 1);
        float* outp;
        int accum = 0;
        float odiv = 1.f;
        if (last_d) {
          outp = xsum;
          accum = (j > 0) ? 1 : 0;
          odiv = (j == nk - 1) ? (float)nk : 1.f;  // x = xs / self.num_kernels
        } else {
          outp = (rx == fa) ? fb : fa;
        }
        // ResBlock2 (two residual convs): both dilations in one launch where the second conv's
        // halo wastes little of the tile; then the d loop is done
        if (c->resblock == 2 && nd == 2 && d == 0 && !m->dec_unfused && !forked &&
            ch <= m->fuse2_maxc &&
            resblock2_chain32_supported(rb.c1[0], rb.c1[1], m->fuse32_lds,
                                        ch <= 32 ? m->fuse2_waste_pct : (m->fuse2_waste_pct + 1) / 2) &&
            cdiv(len, pair_nto(ch, 1) - (rb.c1[1].ktaps - 1) * rb.c1[1].dil) * B >= m->fuse_min_blocks) {
          ResPair32Params pp;
          memset(&pp, 0, sizeof(pp));
          pp.x = rx;
          pp.out = xsum;
          pp.T = len;
          pp.B = B;
          pp.accum = (j > 0) ? 1 : 0;
          pp.out_div = (j == nk - 1) ? (float)nk : 1.f;
          pp.slope = 0.1f;
          WETTS_TRY(launch_resblock2_chain32(rb.c1[0], rb.c1[1], pp, sj));
          if (tm && tm->on) tm->launches += 1;
          mrf_count(m, 1, 2 + (j > 0 ? 1 : 0), ch, len, B, 4);
          rx = xsum;
          break;
        }
        // fused wherever it measures faster (profiles/r01_conv32_fused_pair.txt): everything but
        // the MFMA-bound C=128, k=11 pairs, whose 2*(k-1)/2 discarded columns per 128 outweigh
        // the gain
        // ... and only when the launch has enough time tiles to occupy the chip: a streaming
        // window (50-60 frames) would be 25 blocks of 150 us each; the small unfused tiles spread
        // it over 4x as many CUs (bench.py --stream: 3.4 -> 2.8 ms per window)
        const int pair_tiles = cdiv(len, pair_nto(ch, rb.c1[d].ktaps)) * B;
        // one (c1, c2) pair on the chain kernel: every pair of the C = 32 stage, k = 3 pairs at any width
        const bool chain1 = c->resblock == 1 && !m->dec_unfused && len % 4 == 0 && chain_addr_ok &&
                            (ch <= m->chain_pair_maxc || rb.c1[d].ktaps <= m->chain_pair_kmax) &&
                            resblock_chain32_supported(&rb.c1[d], &rb.c2[d], 1, m->fuse32_lds / 2, 100) &&
                            pair_tiles >= m->fuse_min_blocks;
        if (chain1) {
          if (forked && last_d && j > 0) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          ResChain32Params cp;
          memset(&cp, 0, sizeof(cp));
          cp.x = rx;
          cp.out = outp;
          cp.T = len;
          cp.B = B;
          cp.accum = accum;
          cp.out_div = odiv;
          cp.slope = 0.1f;
          cp.lens = lens;
          cp.len_mul = spf;
          WETTS_TRY(launch_resblock_chain32(&rb.c1[d], &rb.c2[d], 1, cp, sj));
          if (tm && tm->on) tm->launches += 1;
          mrf_count(m, 1, 2 + accum, ch, len, B, 4);
          rx = outp;
          continue;
        }
        const bool fuse32 = c->resblock == 1 && !m->dec_unfused && !lens &&
                            resblock_pair32_supported(rb.c1[d], rb.c2[d], m->fuse32_lds) &&
                            !(ch >= 128 && rb.c1[d].ktaps >= m->fuse32_kmax128) &&
                            (ch <= m->fuse32_maxc || rb.c1[d].ktaps <= m->fuse32_kwide) &&
                            rb.c1[d].ktaps <= m->fuse32_kmax &&
                            pair_tiles >= m->fuse_min_blocks;
        if (fuse32) {
          // x = x + c2(lrelu(c1(lrelu(x)))) in one kernel (intermediate in LDS)
          if (forked && last_d && j > 0) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          ResPair32Params pp;
          memset(&pp, 0, sizeof(pp));
          pp.x = rx;
          pp.out = outp;
          pp.T = len;
          pp.B = B;
          pp.accum = accum;
          pp.out_div = odiv;
          pp.slope = 0.1f;
          WETTS_TRY(launch_resblock_pair32(rb.c1[d], rb.c2[d], pp, sj));
          if (tm && tm->on) tm->launches += 1;
          mrf_count(m, 1, 2 + accum, ch, len, B, 4);
        } else if (c->resblock == 1) {
          // xt = c1(lrelu(x)); x = c2(lrelu(xt)) + x
          ConvParams p1 = conv_io(rx, ch, len, ft, ch, B);
          p1.in_act = IN_LRELU;
          p1.in_slope = 0.1f;
          p1.tag = 1;
          p1.lens = lens;
          p1.len_mul = spf;
          WETTS_TRY(launch_conv(rb.c1[d], p1, sj));
          // the running sum is ordered chain j-1 -> chain j
          if (forked && last_d && j > 0) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          ConvParams p2 = conv_io(ft, ch, len, outp, ch, B);
          p2.in_act = IN_LRELU;
          p2.in_slope = 0.1f;
          p2.res = rx;
          p2.r_bs = (int64_t)ch * len;
          p2.r_cs = len;
          p2.accum = accum;
          p2.out_div = odiv;
          p2.tag = 1;
          p2.lens = lens;
          p2.len_mul = spf;
          WETTS_TRY(launch_conv(rb.c2[d], p2, sj));
          if (tm && tm->on) tm->launches += 2;
          mrf_count(m, 2, 5 + accum, ch, len, B, 4);
        } else {
          // x = c(lrelu(x)) + x
          if (forked && last_d && j > 0) WETTS_HIP_CHECK(hipStreamWaitEvent(sj, m->ev_chain[j - 1], 0));
          ConvParams p1 = conv_io(rx, ch, len, outp, ch, B);
          p1.in_act = IN_LRELU;
          p1.in_slope = 0.1f;
          p1.res = rx;
          p1.r_bs = (int64_t)ch * len;
          p1.r_cs = len;
          p1.accum = accum;
          p1.out_div = odiv;
          p1.tag = 1;
          WETTS_TRY(launch_conv(rb.c1[d], p1, sj));
          if (tm && tm->on) tm->launches += 1;
          mrf_count(m, 1, 2 + accum, ch, len, B, 4);  // (input and residual are one tensor)
        }
        rx = outp;
      }
      if (forked) WETTS_HIP_CHECK(hipEventRecord(m->ev_chain[j], sj));
    }
    // join: the last chain's final conv already waited for all the others through the sum order
    if (forked) WETTS_HIP_CHECK(hipStreamWaitEvent(s, m->ev_chain[nk - 1], 0));
    if (m->mrf_timing) {
      WETTS_HIP_CHECK(hipEventRecord(lv1, s));
      m->mrf_events.emplace_back(lv0, lv1);
    }
    if (tm && tm->on) {
      WETTS_HIP_CHECK(hipEventRecord(tm->e1, s));
      WETTS_HIP_CHECK(hipEventSynchronize(tm->e1));
      float ms = 0.f;
      WETTS_HIP_CHECK(hipEventElapsedTime(&ms, tm->e0, tm->e1));
      tm->mrf_ms += ms;
    }
    x = xsum;
  }
  // x = tanh(conv_post(leaky_relu(x)))   (default slope 0.01, decoders.py:78)
  WETTS_TRY(k_conv_post_tanh(x, m->conv_post_w, 7, B, ch, len, audio, s, lens, spf));
  if (m->mrf_timing) m->mrf_calls += 1;
  return WETTS_OK;
}
}  // namespace wetts

namespace wetts {
static void free_decoder_bf16(const wetts_model* m) {
  for (auto& pc : m->b_ups) free_packed_bf16(&pc);
  for (auto& v : m->b_c1) for (auto& pc : v) free_packed_bf16(&pc);
  for (auto& v : m->b_c2) for (auto& pc : v) free_packed_bf16(&pc);
  free_packed_bf16(&m->b_post);
  free_packed_bf16(&m->b_pre);
  m->b_ups.clear();
  m->b_c1.clear();
  m->b_c2.clear();
  m->dec_packed_prec = 0;
}

static int32_t pack_decoder_bf16_layers(const wetts_model* m, int f16, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  m->b_ups.resize(c->n_upsamples);
  m->b_c1.assign(c->n_upsamples * nk, std::vector<PackedConvB>(nd));
  m->b_c2.assign(c->n_upsamples * nk, std::vector<PackedConvB>(c->resblock == 1 ? nd : 0));
  int ch = c->upsample_initial_channel;
  WETTS_TRY(pack_conv_weight_bf16(m->T("dec.conv_pre.weight"), m->T("dec.conv_pre.bias"), ch, c->inter_channels, 7, 1, 3,
                                  0, 0, f16, s, &m->b_pre));
  for (int i = 0; i < c->n_upsamples; ++i) {
    const int u = c->upsample_rates[i], uk = c->upsample_kernel_sizes[i];
    WETTS_TRY(pack_conv_weight_bf16(m->T(S("dec.ups.%d.weight", i)), m->T(S("dec.ups.%d.bias", i)),
                                    ch / 2, ch, uk, 1, (uk - u) / 2, 1, u, f16, s, &m->b_ups[i]));
    ch /= 2;
    for (int j = 0; j < nk; ++j) {
      const int n = i * nk + j, k = c->resblock_kernel_sizes[j];
      for (int d = 0; d < nd; ++d) {
        const int dil = c->resblock_dilation_sizes[j][d];
        if (c->resblock == 1) {
          WETTS_TRY(pack_conv_weight_bf16(m->T(S("dec.resblocks.%d.convs1.%d.weight", n, d)),
                                          m->T(S("dec.resblocks.%d.convs1.%d.bias", n, d)), ch, ch,
                                          k, dil, (k * dil - dil) / 2, 0, 0, f16, s, &m->b_c1[n][d]));
          WETTS_TRY(pack_conv_weight_bf16(m->T(S("dec.resblocks.%d.convs2.%d.weight", n, d)),
                                          m->T(S("dec.resblocks.%d.convs2.%d.bias", n, d)), ch, ch,
                                          k, 1, (k - 1) / 2, 0, 0, f16, s, &m->b_c2[n][d]));
        } else {
          WETTS_TRY(pack_conv_weight_bf16(m->T(S("dec.resblocks.%d.convs.%d.weight", n, d)),
                                          m->T(S("dec.resblocks.%d.convs.%d.bias", n, d)), ch, ch,
                                          k, dil, (k * dil - dil) / 2, 0, 0, f16, s, &m->b_c1[n][d]));
        }
      }
    }
  }
  if (ch == 32) {  // conv_post on the matrix cores (k_conv_post_mfma16): rows 1..31 of the weight are zero
    const size_t n = (size_t)32 * ch * 7;
    if (!m->b_post_wpad) WETTS_HIP_CHECK(hipMalloc((void**)&m->b_post_wpad, n * sizeof(float)));
    WETTS_HIP_CHECK(hipMemsetAsync(m->b_post_wpad, 0, n * sizeof(float), s));
    WETTS_HIP_CHECK(hipMemcpyAsync(m->b_post_wpad, m->conv_post_w, (size_t)ch * 7 * sizeof(float),
                                   hipMemcpyDeviceToDevice, s));
    WETTS_TRY(pack_conv_weight_bf16(m->b_post_wpad, nullptr, 32, ch, 7, 1, 3, 0, 0, f16, s, &m->b_post));
  }
  return WETTS_OK;
}

// all or nothing, like pack_flow_bf16
static int32_t pack_decoder_bf16(const wetts_model* m, hipStream_t s) {
  const int want = m->dec_precision;
  if ((want != 1 && want != 2) || m->dec_packed_prec == want) return WETTS_OK;
  free_decoder_bf16(m);
  const int32_t rc = pack_decoder_bf16_layers(m, want == 2 ? 1 : 0, s);
  if (rc != WETTS_OK) {
    free_decoder_bf16(m);
    return rc;
  }
  m->dec_packed_prec = want;
  return WETTS_OK;
}

static ConvBParams convb_io(const unsigned short* x, int Cin, int T, unsigned short* out, int Cout,
                            int Tout, int B) {
  ConvBParams p;
  memset(&p, 0, sizeof(p));
  p.x = x;
  p.x_bs = (int64_t)Cin * T;
  p.Cin = Cin;
  p.Tin = T;
  p.in_act = IN_LRELU;
  p.in_slope = 0.1f;
  p.out = out;
  p.o_bs = (int64_t)Cout * Tout;
  p.cout = Cout;
  p.Tout = Tout;
  p.out_div = 1.f;
  p.B = B;
  return p;
}

// Generator.forward with 16-bit channel-last activations between the convs (f32 accumulate).  conv_pre too (round 6):
// (z * y_mask) is rounded to 16 bit channel-last and conv_pre (+ cond as a per-utterance bias) runs on the 16-bit kernel
// -- it was the one conv of the mode left on the f32 kernel, 0.41 ms of the 8.5 ms configs[2] step for a conv that is
// 60 us at 16 bit; conv_post's tanh is f32.
static int32_t run_hifigan_bf16(const wetts_model* m, const float* z, int64_t z_bs, int64_t z_cs,
                                const float* y_mask, int64_t mask_stride, const float* g, int B,
                                int L, float* audio, void* workspace, int64_t workspace_bytes,
                                hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, C0 = c->upsample_initial_channel;
  WETTS_TRY(pack_decoder_bf16(m, s));
  const int64_t mx = dec_max_elems(c, B, L);
  Bump ws(workspace, workspace_bytes);
  unsigned short* z16 = ws.take<unsigned short>((int64_t)B * I * L);
  unsigned short* bx = ws.take<unsigned short>(mx);
  unsigned short* bt = ws.take<unsigned short>(mx);
  unsigned short* bs = ws.take<unsigned short>(mx);
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  unsigned short* fa = ws.take<unsigned short>(mx);
  unsigned short* fb = ws.take<unsigned short>(mx);
  unsigned short* ft = ws.take<unsigned short>(mx);
  float* cond = ws.take<float>((int64_t)B * C0);
  if (!ws.ok) {
    set_error("hifigan(bf16): workspace too small");
    return WETTS_E_WORKSPACE;
  }
  {
    const int f16 = m->dec_precision == 2 ? 1 : 0;
    WETTS_TRY(k_cf32_to_cl16_strided(z, z_bs, z_cs, y_mask, mask_stride, z16, B, I, L, f16, s));
    ConvBParams p = convb_io(z16, I, L, bx, C0, L, B);
    p.in_act = IN_NONE;
    if (has_g(c) && g) {
      WETTS_TRY(k_cond_linear(g, m->dec_cond_w, m->dec_cond_b, B, C0, c->gin_channels, cond, s));
      p.bias_b = cond;
      p.bias_b_stride = C0;
    }
    WETTS_TRY(launch_conv_bf16(m->b_pre, p, s));
  }
  int ch = C0, len = L;
  unsigned short* x = bx;
  for (int i = 0; i < c->n_upsamples; ++i) {
    const int u = c->upsample_rates[i];
    {
      ConvBParams p = convb_io(x, ch, len, bt, ch / 2, len * u, B);
      WETTS_TRY(launch_conv_bf16(m->b_ups[i], p, s));
    }
    ch /= 2;
    len *= u;
    unsigned short* xsum = (x == bx) ? bs : bx;
    hipEvent_t lv0 = nullptr, lv1 = nullptr;
    if (m->mrf_timing) {
      WETTS_HIP_CHECK(hipEventCreate(&lv0));
      WETTS_HIP_CHECK(hipEventCreate(&lv1));
      WETTS_HIP_CHECK(hipEventRecord(lv0, s));
    }
    bool stage_done = false;
    for (int j = 0; j < nk; ++j) {
      const int n = i * nk + j;
      hipStream_t sj = s;
      const unsigned short* rx = bt;
      // a whole stage of ResBlock2 blocks in one launch (resblock2_stage16.hip): x read once, the sum written once
      if (j == 0 && c->resblock == 2 && nd == 2 && nk <= RESSTAGE2_MAX_CHAINS && !m->dec_unfused && m->stage2_pct > 0) {
        const PackedConvB *c1s[RESSTAGE2_MAX_CHAINS], *c2s[RESSTAGE2_MAX_CHAINS];
        for (int q = 0; q < nk; ++q) {
          c1s[q] = &m->b_c1[i * nk + q][0];
          c2s[q] = &m->b_c1[i * nk + q][1];
        }
        const int nto = resblock2_stage16_nto(c1s, c2s, nk, m->stage2_pct);
        if (nto > 0 && cdiv(len, nto) * B >= m->fuse_min_blocks) {
          ResStage2Params sp;
          memset(&sp, 0, sizeof(sp));
          sp.x = bt;
          sp.out = xsum;
          sp.T = len;
          sp.B = B;
          sp.out_div = (float)nk;
          sp.slope = 0.1f;
          WETTS_TRY(launch_resblock2_stage16(c1s, c2s, nk, sp, s));
          mrf_count(m, 1, 2, ch, len, B, 2);
          stage_done = true;
          break;  // every chain of the stage is done (on the caller's stream: nothing to join)
        }
      }
      for (int d = 0; d < nd; ++d) {
        const bool last_d = (d == nd - 1);
        unsigned short* outp = last_d ? xsum : ((rx == fa) ? fb : fa);
        const int accum = (last_d && j > 0) ? 1 : 0;
        const float odiv = (last_d && j == nk - 1) ? (float)nk : 1.f;
        if (c->resblock == 2 && nd == 2 && d == 0 && !m->dec_unfused &&
            resblock2_chain16_supported(m->b_c1[n][0], m->b_c1[n][1],
                                        ch <= 32 ? m->fuse2_waste_pct : (m->fuse2_waste_pct + 1) / 2) &&
            cdiv(len, resblock_pair16_ntc(ch, true) - (m->b_c1[n][1].ktaps - 1) * m->b_c1[n][1].dil) * B >=
                m->fuse_min_blocks) {
          // a whole ResBlock2 in one launch (see resblock16.hip, RB2)
          ResPairParams pp;
          memset(&pp, 0, sizeof(pp));
          pp.x = rx;
          pp.out = xsum;
          pp.T = len;
          pp.B = B;
          pp.accum = (j > 0) ? 1 : 0;
          pp.out_div = (j == nk - 1) ? (float)nk : 1.f;
          pp.slope = 0.1f;
          WETTS_TRY(launch_resblock2_chain16(m->b_c1[n][0], m->b_c1[n][1], pp, s));
          mrf_count(m, 1, 2 + (j > 0 ? 1 : 0), ch, len, B, 2);
          rx = xsum;
          break;
        }
        if (c->resblock == 1 && !m->dec_unfused &&
            resblock_pair16_supported(m->b_c1[n][d], m->b_c2[n][d]) &&
            cdiv(len, pair_nto(ch, m->b_c1[n][d].ktaps)) * B >= m->fuse_min_blocks) {
          ResPairParams pp;
          memset(&pp, 0, sizeof(pp));
          pp.x = rx;
          pp.out = outp;
          pp.T = len;
          pp.B = B;
          pp.accum = accum;
          pp.out_div = odiv;
          pp.slope = 0.1f;
          WETTS_TRY(launch_resblock_pair16(m->b_c1[n][d], m->b_c2[n][d], pp, sj));
          mrf_count(m, 1, 2 + accum, ch, len, B, 2);
        } else if (c->resblock == 1) {
          ConvBParams p1 = convb_io(rx, ch, len, ft, ch, len, B);
          p1.basic = m->dec_unfused;
          p1.tag = 1;
          WETTS_TRY(launch_conv_bf16(m->b_c1[n][d], p1, sj));
          ConvBParams p2 = convb_io(ft, ch, len, outp, ch, len, B);
          p2.basic = m->dec_unfused;
          p2.tag = 1;
          p2.res = rx;
          p2.r_bs = (int64_t)ch * len;
          p2.accum = accum;
          p2.out_div = odiv;
          WETTS_TRY(launch_conv_bf16(m->b_c2[n][d], p2, sj));
          mrf_count(m, 2, 5 + accum, ch, len, B, 2);
        } else {
          ConvBParams p1 = convb_io(rx, ch, len, outp, ch, len, B);
          p1.res = rx;
          p1.r_bs = (int64_t)ch * len;
          p1.accum = accum;
          p1.out_div = odiv;
          p1.tag = 1;
          WETTS_TRY(launch_conv_bf16(m->b_c1[n][d], p1, s));
          mrf_count(m, 1, 2 + accum, ch, len, B, 2);
        }
        rx = outp;
      }
    }
    (void)stage_done;
    if (m->mrf_timing) {
      WETTS_HIP_CHECK(hipEventRecord(lv1, s));
      m->mrf_events.emplace_back(lv0, lv1);
    }
    x = xsum;
  }
  if (m->mrf_timing) m->mrf_calls += 1;
  if (m->b_post.wpk && ch == 32) return k_conv_post_mfma16(m->b_post, x, B, ch, len, audio, s);
  return k_conv_post_bf16(x, m->conv_post_w, 7, B, ch, len, audio, m->dec_precision == 2 ? 1 : 0, s);
}
}  // namespace wetts

namespace wetts {
static void free_decoder_u8(const wetts_model* m) {
  free_packed_qconv(&m->q_pre);
  free_packed_qconv(&m->q_cond);
  free_packed_qconv(&m->q_post);
  for (auto& v : m->q_c1) for (auto& pc : v) free_packed_qconv(&pc);
  for (auto& v : m->q_c2) for (auto& pc : v) free_packed_qconv(&pc);
  m->q_c1.clear();
  m->q_c2.clear();
  m->q_packed = false;
}

static int32_t pack_decoder_u8_layers(const wetts_model* m, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, C0 = c->upsample_initial_channel;
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  WETTS_TRY(pack_qconv_weight(m->T("dec.conv_pre.weight"), m->T("dec.conv_pre.bias"), C0, I, 7, 1, 3, s,
                              &m->q_pre));
  if (has_g(c))
    WETTS_TRY(pack_qconv_weight(m->T("dec.cond.weight"), m->T("dec.cond.bias"), C0, c->gin_channels, 1,
                                1, 0, s, &m->q_cond));
  m->q_c1.assign(c->n_upsamples * nk, std::vector<PackedQConv>(nd));
  m->q_c2.assign(c->n_upsamples * nk, std::vector<PackedQConv>(c->resblock == 1 ? nd : 0));
  int ch = C0;
  for (int i = 0; i < c->n_upsamples; ++i) {
    ch /= 2;
    for (int j = 0; j < nk; ++j) {
      const int n = i * nk + j, k = c->resblock_kernel_sizes[j];
      for (int d = 0; d < nd; ++d) {
        const int dil = c->resblock_dilation_sizes[j][d];
        if (c->resblock == 1) {
          WETTS_TRY(pack_qconv_weight(m->T(S("dec.resblocks.%d.convs1.%d.weight", n, d)),
                                      m->T(S("dec.resblocks.%d.convs1.%d.bias", n, d)), ch, ch, k, dil,
                                      (k * dil - dil) / 2, s, &m->q_c1[n][d]));
          WETTS_TRY(pack_qconv_weight(m->T(S("dec.resblocks.%d.convs2.%d.weight", n, d)),
                                      m->T(S("dec.resblocks.%d.convs2.%d.bias", n, d)), ch, ch, k, 1,
                                      (k - 1) / 2, s, &m->q_c2[n][d]));
        } else {
          WETTS_TRY(pack_qconv_weight(m->T(S("dec.resblocks.%d.convs.%d.weight", n, d)),
                                      m->T(S("dec.resblocks.%d.convs.%d.bias", n, d)), ch, ch, k, dil,
                                      (k * dil - dil) / 2, s, &m->q_c1[n][d]));
        }
      }
    }
  }
  WETTS_TRY(pack_qconv_weight(m->T("dec.conv_post.weight"), nullptr, 1, ch, 7, 1, 3, s, &m->q_post));
  return WETTS_OK;
}

// all or nothing: a failure part-way frees the buffers already created (a retry does not leak them)
static int32_t pack_decoder_u8(const wetts_model* m, hipStream_t s) {
  if (m->q_packed) return WETTS_OK;
  free_decoder_u8(m);
  const int32_t rc = pack_decoder_u8_layers(m, s);
  if (rc != WETTS_OK) {
    free_decoder_u8(m);
    return rc;
  }
  m->q_packed = true;
  return WETTS_OK;
}

static QConvIO qio(const float* x, int Cin, int T, float* out, int Cout, int B) {
  QConvIO io;
  memset(&io, 0, sizeof(io));
  io.x = x; io.x_bs = (int64_t)Cin * T; io.x_cs = T;
  io.out = out; io.o_bs = (int64_t)Cout * T; io.o_cs = T;
  io.out_div = 1.f; io.B = B; io.T = T;
  return io;
}

// Generator.forward as the graph `export_onnx.py --quant` leaves behind (export_onnx.py:149-157):
// every Conv1d dynamically quantised to uint8 (qconv_u8.hip), ConvTranspose1d and the element-wise
// ops in float32.
static int32_t run_hifigan_u8(const wetts_model* m, const float* z, int64_t z_bs, int64_t z_cs,
                              const float* y_mask, int64_t mask_stride, const float* g, int B, int L,
                              float* audio, void* workspace, int64_t workspace_bytes, hipStream_t s) {
  const wetts_config_t* c = &m->cfg;
  const int I = c->inter_channels, C0 = c->upsample_initial_channel;
  WETTS_TRY(pack_decoder_u8(m, s));
  const int64_t mx = dec_max_elems(c, B, L);
  int64_t lenmax = L;
  for (int i = 0; i < c->n_upsamples; ++i) lenmax *= c->upsample_rates[i];
  Bump ws(workspace, workspace_bytes);
  float* bx = ws.take<float>(mx);
  float* bt = ws.take<float>(mx);
  float* bs = ws.take<float>(mx);
  float* fa = ws.take<float>(mx);
  float* fb = ws.take<float>(mx);
  float* ft = ws.take<float>(mx);
  float* cond = ws.take<float>((int64_t)B * C0);
  const int64_t qbytes = align_up(mx + 32 * (int64_t)B * lenmax, 256) + align_up(4 * (int64_t)B * lenmax, 256) + 1024 +
                         align_up((int64_t)B * (lenmax / 128 + 1) * 64, 256);  // + the per-block range records
  char* qs = ws.take<char>(qbytes);
  // one range slot per tensor a quantised conv consumes (qconv_u8.h): the producing conv fills it in its epilogue
  const int nslots = c->n_upsamples * (1 + c->n_resblock_kernels * c->n_resblock_dilations * 2) + 2;
  QuantStats* slots = ws.take<QuantStats>(nslots);
  if (!ws.ok) {
    set_error("hifigan(uint8): workspace too small");
    return WETTS_E_WORKSPACE;
  }
  WETTS_TRY(k_qstats_reset(slots, nslots, s));
  const bool use_g = has_g(c) && g;
  if (use_g) {  // cond(g): a Conv node like the others
    QConvIO io = qio(g, c->gin_channels, 1, cond, C0, B);
    WETTS_TRY(launch_qconv(m->q_cond, io, qs, qbytes, s));
  }
  {
    QConvIO io = qio(z, I, L, bx, C0, B);
    io.x_bs = z_bs;
    io.x_cs = z_cs;
    io.mask = y_mask;
    io.mask_stride = mask_stride;
    if (use_g) {
      io.bias_b = cond;
      io.bias_b_stride = C0;
    }
    WETTS_TRY(launch_qconv(m->q_pre, io, qs, qbytes, s));
  }
  int ch = C0, len = L;
  float* x = bx;
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  for (int i = 0; i < c->n_upsamples; ++i) {
    const int u = c->upsample_rates[i];
    {  // ConvTranspose1d stays float32 (dynamic quantisation does not touch it)
      ConvParams p = conv_io(x, ch, len, bt, ch / 2, B);
      p.in_act = IN_LRELU;
      p.in_slope = 0.1f;
      p.Tout = len * u;
      p.o_bs = (int64_t)(ch / 2) * len * u;
      p.o_cs = (int64_t)len * u;
      WETTS_TRY(launch_conv(m->ups[i], p, s));
    }
    ch /= 2;
    len *= u;
    float* xsum = (x == bx) ? bs : bx;
    hipEvent_t lv0 = nullptr, lv1 = nullptr;
    if (m->mrf_timing) {  // the MRF ResBlock class of this mode: every quantised conv of the stage (its kernels)
      WETTS_HIP_CHECK(hipEventCreate(&lv0));
      WETTS_HIP_CHECK(hipEventCreate(&lv1));
      WETTS_HIP_CHECK(hipEventRecord(lv0, s));
    }
    // range slots of this stage: [0] the upsampled input (no quantised conv produced it: one pass here, shared by
    // the first conv of every chain), then per (chain, dilation) the c1 output and the c2 output; the very last
    // slot is the last stage's output, which conv_post reads
    QuantStats* sl = slots + (size_t)i * (1 + nk * nd * 2);
    QuantStats* sl_in = sl;
    WETTS_TRY(k_qminmax(bt, B, ch, len, sl_in, s));
    QuantStats* sl_stage_out = (i == c->n_upsamples - 1) ? slots + nslots - 1 : nullptr;
    for (int j = 0; j < nk; ++j) {
      const int n = i * nk + j;
      const float* rx = bt;
      const QuantStats* rx_stats = sl_in;
      for (int d = 0; d < nd; ++d) {
        const bool last_d = (d == nd - 1);
        float* outp = last_d ? xsum : ((rx == fa) ? fb : fa);
        const float* cin = rx;
        const QuantStats* cin_stats = rx_stats;
        QuantStats* s_ft = sl + 1 + (size_t)(j * nd + d) * 2;
        QuantStats* s_out = s_ft + 1;
        mrf_count(m, (c->resblock == 1) ? 2 : 1, ((c->resblock == 1) ? 5 : 2) + ((last_d && j > 0) ? 1 : 0), ch, len, B, 4);  // f32 tensors between the nodes
        if (c->resblock == 1) {
          QConvIO i1 = qio(rx, ch, len, ft, ch, B);
          i1.in_act = 1;
          i1.in_slope = 0.1f;
          i1.in_stats = rx_stats;
          i1.out_stats = s_ft;
          WETTS_TRY(launch_qconv(m->q_c1[n][d], i1, qs, qbytes, s));
          cin = ft;
          cin_stats = s_ft;
        }
        QConvIO i2 = qio(cin, ch, len, outp, ch, B);
        i2.in_act = 1;
        i2.in_slope = 0.1f;
        i2.res = rx;
        i2.r_bs = (int64_t)ch * len;
        i2.r_cs = len;
        i2.accum = (last_d && j > 0) ? 1 : 0;
        i2.out_div = (last_d && j == nk - 1) ? (float)nk : 1.f;
        i2.in_stats = cin_stats;
        // what this conv writes is read by the chain's next conv, or -- the last launch of the last stage -- by conv_post
        i2.out_stats = !last_d ? s_out : (j == nk - 1 ? sl_stage_out : nullptr);
        WETTS_TRY(launch_qconv(c->resblock == 1 ? m->q_c2[n][d] : m->q_c1[n][d], i2, qs, qbytes, s));
        rx = outp;
        rx_stats = s_out;
      }
    }
    if (m->mrf_timing) {
      WETTS_HIP_CHECK(hipEventRecord(lv1, s));
      m->mrf_events.emplace_back(lv0, lv1);
    }
    x = xsum;
  }
  if (m->mrf_timing) m->mrf_calls += 1;
  {  // leaky_relu (default slope 0.01) -> conv_post (no bias) -> tanh
    QConvIO io = qio(x, ch, len, audio, 1, B);
    io.in_act = 1;
    io.in_slope = 0.01f;
    io.in_stats = c->n_upsamples > 0 ? slots + nslots - 1 : nullptr;
    WETTS_TRY(launch_qconv(m->q_post, io, qs, qbytes, s));
    WETTS_TRY(k_tanh_inplace(audio, (int64_t)B * len, s));
  }
  return WETTS_OK;
}
}  // namespace wetts

int32_t wetts_dynamic_quant_conv1d(const float* x, const float* w, const float* bias, int32_t B,
                                   int32_t Cin, int32_t Cout, int32_t k, int32_t dilation, int32_t padding,
                                   int32_t T, float* out, void* stream) {
  WETTS_REQUIRE(x && w && out && B > 0 && Cin > 0 && Cout > 0 && k > 0 && T > 0, "bad argument");
  WETTS_REQUIRE(2 * padding == (k - 1) * dilation, "only 'same' padding (output length T)");
  hipStream_t s = (hipStream_t)stream;
  PackedQConv pc;
  int32_t rc = pack_qconv_weight(w, bias, Cout, Cin, k, dilation, padding, s, &pc);
  void* scratch = nullptr;
  const int64_t nb = qconv_scratch_bytes(B, Cin, T);
  if (rc == WETTS_OK && hipMalloc(&scratch, (size_t)nb) != hipSuccess) {
    set_error("dynamic_quant_conv1d: hipMalloc(%lld) failed", (long long)nb);
    rc = WETTS_E_HIP;
  }
  if (rc == WETTS_OK) {
    QConvIO io = qio(x, Cin, T, out, Cout, B);
    rc = launch_qconv(pc, io, scratch, nb, s);
  }
  if (rc == WETTS_OK && hipStreamSynchronize(s) != hipSuccess) {
    set_error("dynamic_quant_conv1d: kernel failed");
    rc = WETTS_E_HIP;
  }
  if (scratch) (void)hipFree(scratch);
  free_packed_qconv(&pc);
  return rc;
}

int32_t wetts_set_decoder_precision(const wetts_model_t* m, int32_t precision) {
  WETTS_REQUIRE(m != nullptr, "null model");
  const int unfused = (precision & WETTS_DECODER_UNFUSED) ? 1 : 0;
  const int serial = (precision & WETTS_DECODER_SERIAL) ? 1 : 0;
  precision &= ~(WETTS_DECODER_UNFUSED | WETTS_DECODER_SERIAL);
  WETTS_REQUIRE(precision >= 0 && precision <= 3,
                "precision must be 0 (f32), 1 (bf16), 2 (f16) or 3 (uint8 dynamic quantisation)");
  m->dec_serial = serial;  // (only a validated request changes the model)
  m->dec_unfused = unfused;
  WETTS_REQUIRE(precision == 0 || m->cfg.vocoder_type == 0,
                "the 16-bit decoder mode covers the HiFi-GAN generator only");
  if (precision == 1 || precision == 2) {
    const wetts_config_t* c = &m->cfg;
    WETTS_REQUIRE((c->upsample_initial_channel >> c->n_upsamples) % 32 == 0,
                  "bf16 decoder needs every stage width to be a multiple of 32 channels");
  }
  m->dec_precision = precision;
  // packed here, not on first use: the stage calls never allocate (header contract)
  if (precision == 1 || precision == 2) WETTS_TRY(pack_decoder_bf16(m, nullptr));
  if (precision == 3) WETTS_TRY(pack_decoder_u8(m, nullptr));
  if (precision) WETTS_HIP_CHECK(hipStreamSynchronize(nullptr));
  return WETTS_OK;
}

int32_t wetts_hifigan(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                      int64_t z_channel_stride, const float* y_mask, int64_t mask_stride,
                      const float* g, int32_t B, int32_t L, float* audio, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && z && audio, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  if (B == 0 || L == 0) return WETTS_OK;
  if (m->cfg.vocoder_type == 1)
    return run_vocos(m, z, z_batch_stride, z_channel_stride, y_mask, mask_stride, g, B, L, audio,
                     workspace, workspace_bytes, (hipStream_t)stream);
  if (m->dec_precision == 3)
    return run_hifigan_u8(m, z, z_batch_stride, z_channel_stride, y_mask, mask_stride, g, B, L, audio,
                          workspace, workspace_bytes, (hipStream_t)stream);
  if (m->dec_precision >= 1)
    return run_hifigan_bf16(m, z, z_batch_stride, z_channel_stride, y_mask, mask_stride, g, B, L,
                            audio, workspace, workspace_bytes, (hipStream_t)stream);
  return run_hifigan(m, z, z_batch_stride, z_channel_stride, y_mask, mask_stride, g, B, L, audio,
                     workspace, workspace_bytes, (hipStream_t)stream, nullptr);
}

int32_t wetts_hifigan_ragged_supported(const wetts_model_t* m) {
  if (!m) return 0;
  const wetts_config_t* c = &m->cfg;
  if (c->vocoder_type != 0 || c->resblock != 1 || m->dec_precision != 0) return 0;
  for (int i = 0; i < c->n_upsamples; ++i)
    if (c->upsample_rates[i] <= 0) return 0;
  return 1;
}

int32_t wetts_hifigan_ragged(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                             int64_t z_channel_stride, const int64_t* y_lengths, const float* g,
                             int32_t B, int32_t L, float* audio, void* workspace, int64_t workspace_bytes,
                             void* stream) {
  WETTS_REQUIRE(m && z && audio && y_lengths, "null argument");
  WETTS_REQUIRE(wetts_hifigan_ragged_supported(m),
                "ragged decode covers the float32 HiFi-GAN generator with ResBlock1 (set_decoder_precision 0)");
  SmallConvScope small_scope(0);  // the small-launch schedule has no per-utterance extents
  if (B == 0 || L == 0) return WETTS_OK;
  return run_hifigan(m, z, z_batch_stride, z_channel_stride, nullptr, 0, g, B, L, audio, workspace,
                     workspace_bytes, (hipStream_t)stream, nullptr, y_lengths);
}

int32_t wetts_profile_hifigan(const wetts_model_t* m, const float* z, int64_t z_batch_stride,
                              int64_t z_channel_stride, const float* g, int32_t B, int32_t L,
                              float* audio, void* workspace, int64_t workspace_bytes, void* stream,
                              double* mrf_ms, double* total_ms, int32_t* mrf_launches) {
  WETTS_REQUIRE(m && z && audio && mrf_ms && total_ms && mrf_launches, "null argument");
  SmallConvScope small_scope(m->small_max_tiles);
  WETTS_REQUIRE(m->cfg.vocoder_type == 0, "wetts_profile_hifigan: HiFi-GAN models only");
  hipStream_t s = (hipStream_t)stream;
  DecTiming tm;
  tm.on = true;
  hipEvent_t t0, t1;
  WETTS_HIP_CHECK(hipEventCreate(&tm.e0));
  WETTS_HIP_CHECK(hipEventCreate(&tm.e1));
  WETTS_HIP_CHECK(hipEventCreate(&t0));
  WETTS_HIP_CHECK(hipEventCreate(&t1));
  WETTS_HIP_CHECK(hipEventRecord(t0, s));
  int32_t r = run_hifigan(m, z, z_batch_stride, z_channel_stride, nullptr, 0, g, B, L, audio,
                          workspace, workspace_bytes, s, &tm);
  if (r == WETTS_OK) {
    WETTS_HIP_CHECK(hipEventRecord(t1, s));
    WETTS_HIP_CHECK(hipEventSynchronize(t1));
    float ms = 0.f;
    WETTS_HIP_CHECK(hipEventElapsedTime(&ms, t0, t1));
    *total_ms = ms;
    *mrf_ms = tm.mrf_ms;
    *mrf_launches = tm.launches;
  }
  (void)hipEventDestroy(tm.e0);
  (void)hipEventDestroy(tm.e1);
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  return r;
}

int32_t wetts_set_mrf_timing(const wetts_model_t* m, int32_t enable) {
  WETTS_REQUIRE(m != nullptr, "null model");
  for (auto& pr : m->mrf_events) {
    (void)hipEventDestroy(pr.first);
    (void)hipEventDestroy(pr.second);
  }
  m->mrf_events.clear();
  m->mrf_launches = 0;
  m->mrf_calls = 0;
  m->mrf_bytes = 0;
  m->mrf_timing = enable != 0;
  return WETTS_OK;
}

int32_t wetts_read_mrf_timing(const wetts_model_t* m, double* mrf_ms, int64_t* conv_launches,
                              int32_t* hifigan_calls) {
  WETTS_REQUIRE(m && mrf_ms && conv_launches && hifigan_calls, "null argument");
  double tot = 0;
  for (auto& pr : m->mrf_events) {
    WETTS_HIP_CHECK(hipEventSynchronize(pr.second));
    float ms = 0.f;
    WETTS_HIP_CHECK(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
  }
  *mrf_ms = tot;
  *conv_launches = m->mrf_launches;
  *hifigan_calls = m->mrf_calls;
  return WETTS_OK;
}

int32_t wetts_read_mrf_bytes(const wetts_model_t* m, double* launched_bytes) {
  WETTS_REQUIRE(m && launched_bytes, "null argument");
  *launched_bytes = m->mrf_bytes;
  return WETTS_OK;
}

int32_t wetts_mas(const float* neg_cent, const int32_t* t_ys, const int32_t* t_xs, int32_t B,
                  int32_t Ty, int32_t Tx, int32_t* path, void* workspace, int64_t workspace_bytes,
                  void* stream) {
  WETTS_REQUIRE(neg_cent && t_ys && t_xs && path, "null argument");
  if (B == 0 || Ty == 0 || Tx == 0) return WETTS_OK;
  if (workspace == nullptr || workspace_bytes < (int64_t)B * Ty * Tx * 4) {
    set_error("mas: workspace too small (need %lld bytes)", (long long)B * Ty * Tx * 4);
    return WETTS_E_WORKSPACE;
  }
  return k_mas(neg_cent, t_ys, t_xs, B, Ty, Tx, path, (float*)workspace, (hipStream_t)stream);
}

int32_t wetts_audio_to_int16(const float* audio, const int64_t* lengths_samples, int32_t B,
                             int64_t L, int16_t* pcm, void* stream) {
  WETTS_REQUIRE(audio && pcm, "null argument");
  return k_audio_to_int16(audio, lengths_samples, B, L, pcm, (hipStream_t)stream);
}

int32_t wetts_hifigan_cost(const wetts_config_t* c, double* flops_per_frame,
                           double* bytes_per_frame_perconv, double* mrf_flops_per_frame,
                           double* mrf_bytes_per_frame_perconv) {
  WETTS_TRY(validate_config(c));
  if (c->vocoder_type == 1) {  // VocosGenerator: all work is per frame; "mrf" = the ConvNeXt stack
    const double I = c->inter_channels, VC = c->vocos_channels, VH = c->vocos_h_channels;
    const double NF = c->istft_n_fft, VO = NF + 2, NL = c->vocos_num_layers;
    const double stack_mac = NL * (3 * VC + 2 * VC * VH);
    const double stack_el = NL * (2 * VC + 2 * VC + (VC + VH) + (VH + 2 * VC));  // dw, LN, pw1, pw2+res
    const double mac = I * VC + stack_mac + VC * VO + VO * NF;
    const double el = (I + VC) + 2 * VC + stack_el + 2 * VC + (VC + VO) + 2 * VO + (VO + NF) +
                      (NF + c->istft_hop_length);
    if (flops_per_frame) *flops_per_frame = 2 * mac;
    if (bytes_per_frame_perconv) *bytes_per_frame_perconv = 4 * el;
    if (mrf_flops_per_frame) *mrf_flops_per_frame = 2 * stack_mac;
    if (mrf_bytes_per_frame_perconv) *mrf_bytes_per_frame_perconv = 4 * stack_el;
    return WETTS_OK;
  }
  // SURVEY.md §8(d) formulas (per input frame; L = samples per frame at the current stage)
  double mac = 0, elems = 0, mrf_mac = 0, mrf_elems = 0;
  const double C0 = c->upsample_initial_channel, I = c->inter_channels;
  mac += I * C0 * 7;
  elems += I + C0;
  double ch = C0, Ls = 1;
  const int nk = c->n_resblock_kernels, nd = c->n_resblock_dilations;
  const int nconv = (c->resblock == 1 ? 2 : 1) * nd, nres = nd;
  for (int i = 0; i < c->n_upsamples; ++i) {
    double co = ch / 2, Lo = Ls * c->upsample_rates[i];
    mac += ch * co * c->upsample_kernel_sizes[i] * Ls;
    elems += ch * Ls + co * Lo;
    ch = co;
    Ls = Lo;
    for (int j = 0; j < nk; ++j) {
      double m1 = (double)nconv * ch * ch * c->resblock_kernel_sizes[j] * Ls;
      double e1 = (double)nconv * 2 * ch * Ls + (double)nres * ch * Ls;
      mrf_mac += m1;
      mrf_elems += e1;
    }
    mrf_elems += (double)(nk + 1) * ch * Ls;
  }
  mac += mrf_mac + ch * 7 * Ls;
  elems += mrf_elems + ch * Ls + Ls;
  if (flops_per_frame) *flops_per_frame = 2 * mac;
  if (bytes_per_frame_perconv) *bytes_per_frame_perconv = 4 * elems;
  if (mrf_flops_per_frame) *mrf_flops_per_frame = 2 * mrf_mac;
  if (mrf_bytes_per_frame_perconv) *mrf_bytes_per_frame_perconv = 4 * mrf_elems;
  return WETTS_OK;
}

int64_t wetts_infer_workspace_bytes(const wetts_model_t* m, int32_t B, int32_t Tx,
                                    int32_t max_frames) {
  if (!m || B < 0 || Tx < 0 || max_frames < 0) return WETTS_E_INVALID;
  const wetts_config_t* c = &m->cfg;
  const int64_t H = c->hidden_channels, I = c->inter_channels;
  const int64_t gin = c->gin_channels > 0 ? c->gin_channels : 1;
  int64_t n = A256(B * gin) + A256(B * H * Tx) + A256(B * 2 * I * Tx) + 4 * A256((int64_t)B * Tx) +
              align_up((int64_t)(B + 1) * 8, 256) + 2 * A256((int64_t)B * max_frames) +
              2 * A256(B * I * (int64_t)max_frames) +
              A256((int64_t)B * 2 * Tx) + A256(B * I * (int64_t)max_frames);  // internal eps_w / eps_z
  return n + wetts_workspace_bytes(m, B, Tx, max_frames);
}

int32_t wetts_infer(const wetts_model_t* m, const int64_t* x, const int64_t* x_lengths,
                    const int64_t* sid, const float* eps_w, const float* eps_z,
                    float noise_scale, float length_scale, float noise_scale_w, int32_t B,
                    int32_t Tx, int32_t max_frames, float* audio, int64_t* y_lengths_host,
                    int32_t* frames_out, void* workspace, int64_t workspace_bytes, void* stream) {
  WETTS_REQUIRE(m && x && x_lengths && audio && y_lengths_host && frames_out, "null argument");
  WETTS_REQUIRE(B > 0 && Tx > 0 && max_frames > 0, "empty batch");
  hipStream_t s = (hipStream_t)stream;
  const wetts_config_t* c = &m->cfg;
  const int H = c->hidden_channels, I = c->inter_channels;
  const int gin = c->gin_channels > 0 ? c->gin_channels : 1;
  // persistent region at the head of the workspace, stage scratch after it
  Bump ws(workspace, workspace_bytes);
  float* g = ws.take<float>((int64_t)B * gin);
  float* x_enc = ws.take<float>((int64_t)B * H * Tx);
  float* stats = ws.take<float>((int64_t)B * 2 * I * Tx);
  float* x_mask = ws.take<float>((int64_t)B * Tx);
  float* logw = ws.take<float>((int64_t)B * Tx);
  float* w_ceil = ws.take<float>((int64_t)B * Tx);
  float* cum = ws.take<float>((int64_t)B * Tx);
  int64_t* ylen = ws.take<int64_t>(B + 1);  // [B] lengths + the status word: ONE D2H, one sync
  int32_t* status = reinterpret_cast<int32_t*>(ylen + B);
  float* own_eps_w = ws.take<float>((int64_t)B * 2 * Tx);
  float* own_eps_z = ws.take<float>((int64_t)B * I * max_frames);
  int32_t* f2p = ws.take<int32_t>((int64_t)B * max_frames);
  float* y_mask = ws.take<float>((int64_t)B * max_frames);
  float* z_p = ws.take<float>((int64_t)B * I * max_frames);
  float* z = ws.take<float>((int64_t)B * I * max_frames);
  if (!ws.ok) {
    set_error("infer: workspace too small");
    return WETTS_E_WORKSPACE;
  }
  void* scratch = (char*)workspace + ws.off;
  const int64_t scratch_bytes = workspace_bytes - ws.off;
  // errors the reference raises from inside its modules are collected in a device word for the
  // duration of this call and read back with y_lengths
  struct StatusScope {
    const wetts_model* m;
    int32_t* saved;
    ~StatusScope() { m->status_word = saved; }
  } scope{m, m->status_word};
  WETTS_HIP_CHECK(hipMemsetAsync(ylen + B, 0, sizeof(int64_t), s));
  m->status_word = status;
  // the two torch.randn draws of the reference (duration_predictors.py:257, models.py:267)
  if (c->use_sdp && !eps_w) {
    WETTS_TRY(k_randn(own_eps_w, (int64_t)B * 2 * Tx, m->rng_seed, m->rng_offset, s));
    m->rng_offset += ((uint64_t)B * 2 * Tx + 3) / 4;
    eps_w = own_eps_w;
  }
  WETTS_TRY(wetts_speaker_embedding(m, sid, B, g, stream));
  const float* gp = has_g(c) ? g : nullptr;
  WETTS_TRY(wetts_text_encoder(m, x, x_lengths, gp, B, Tx, x_enc, stats, x_mask, scratch,
                               scratch_bytes, stream));
  if (c->use_sdp) {
    WETTS_TRY(wetts_duration_sdp(m, x_enc, x_mask, gp, eps_w, noise_scale_w, B, Tx, logw, status,
                                 scratch, scratch_bytes, stream));
  } else {
    WETTS_TRY(wetts_duration_dp(m, x_enc, x_mask, gp, B, Tx, logw, scratch, scratch_bytes, stream));
  }
  WETTS_TRY(wetts_durations_to_lengths(logw, x_mask, length_scale, B, Tx, w_ceil, cum, ylen,
                                       status, stream));
  std::vector<int64_t> host((size_t)B + 1);
  WETTS_HIP_CHECK(hipMemcpyAsync(host.data(), ylen, (size_t)(B + 1) * sizeof(int64_t),
                                 hipMemcpyDeviceToHost, s));
  WETTS_HIP_CHECK(hipStreamSynchronize(s));  // commons.py:114-115 `length.max()`
  memcpy(y_lengths_host, host.data(), (size_t)B * sizeof(int64_t));
  const int32_t st = (int32_t)(host[B] & 0xffffffff);
  if (st & (WETTS_STATUS_PHONE_ID_RANGE | WETTS_STATUS_SPEAKER_ID_RANGE)) {
    set_error("infer: %s id outside its embedding table (IndexError in the reference, %s)",
              (st & WETTS_STATUS_PHONE_ID_RANGE) ? "phoneme" : "speaker",
              (st & WETTS_STATUS_PHONE_ID_RANGE) ? "encoders.py:48" : "models.py:239");
    return WETTS_E_INVALID;
  }
  if (st & (WETTS_STATUS_SPLINE_DOMAIN | WETTS_STATUS_DURATION_NONFINITE)) {
    set_error("infer: %s", (st & WETTS_STATUS_SPLINE_DOMAIN)
                               ? "spline inverse: negative discriminant (transforms.py:171 asserts)"
                               : "non-finite predicted durations");
    return WETTS_E_DOMAIN;
  }
  int64_t Ty = 1;
  for (int b = 0; b < B; ++b)
    if (y_lengths_host[b] > Ty) Ty = y_lengths_host[b];
  *frames_out = (int32_t)Ty;
  if (Ty > max_frames) {
    set_error("infer: predicted %lld frames > capacity %d", (long long)Ty, max_frames);
    return WETTS_E_WORKSPACE;
  }
  // models.py:267 `torch.randn_like(m_p)`: drawn now that Ty is known, packed [B, I, Ty] -- the same draw the
  // Python and C++ hosts make, so a seed gives the same audio whatever capacity the caller passed
  int64_t eps_bs = (int64_t)I * max_frames, eps_cs = max_frames;  // a caller's eps_z: [B, I, max_frames]
  if (!eps_z) {
    WETTS_TRY(k_randn(own_eps_z, (int64_t)B * I * Ty, m->rng_seed, m->rng_offset, s));
    m->rng_offset += ((uint64_t)B * I * Ty + 3) / 4;
    eps_z = own_eps_z;
    eps_bs = (int64_t)I * Ty;
    eps_cs = Ty;
  }
  WETTS_TRY(wetts_length_regulate(m, stats, cum, x_mask, ylen, eps_z, eps_bs, eps_cs, noise_scale, B, Tx,
                                  (int)Ty, f2p, y_mask, nullptr, nullptr, nullptr, z_p, stream));
  WETTS_TRY(wetts_flow_reverse(m, z_p, y_mask, gp, B, (int)Ty, z, scratch, scratch_bytes, stream));
  // o = dec((z * y_mask), g); audio rows are packed with stride Ty*hop
  WETTS_TRY(wetts_hifigan(m, z, (int64_t)I * Ty, Ty, y_mask, Ty, gp, B, (int)Ty, audio, scratch,
                          scratch_bytes, stream));
  return WETTS_OK;
}

}  // extern "C"
