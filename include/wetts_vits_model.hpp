// wetts_vits_model.hpp -- native C++ twin of the reference's `wetts::VitsModel`
// (runtime/core/model/vits_model.h:30-66, vits_model.cc:27-153) on top of the C ABI in
// wetts_hip.h: same call shape -- Forward(phonemes, sid, &audio), SetInput / StreamDecode with the
// chunk + overlap-discard protocol (SplitToChunks :96-111, Depadding :114-126) -- but the
// arithmetic runs in libwetts_hip.so on the MI355X instead of two ONNX-Runtime sessions.
// Header-only; needs the HIP runtime for device buffers (compile with hipcc or g++ -lamdhip64).
//
// Like the reference class it is NOT thread-safe (per-object streaming state).
#ifndef WETTS_VITS_MODEL_HPP_
#define WETTS_VITS_MODEL_HPP_

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "wetts_hip.h"

namespace wetts_hip {

class DeviceBuffer {
 public:
  DeviceBuffer() = default;
  ~DeviceBuffer() { if (p_) (void)hipFree(p_); }
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  // grow-only, with 25 % headroom: sizes follow the (sampled) frame count, and a request that sets a new
  // maximum should not cost a hipFree + hipMalloc every time
  void* Reserve(size_t bytes) {
    if (bytes > cap_) {
      if (p_) (void)hipFree(p_);
      p_ = nullptr;
      const size_t cap = bytes + bytes / 4;
      if (hipMalloc(&p_, cap) != hipSuccess) throw std::runtime_error("hipMalloc failed");
      cap_ = cap;
    }
    return p_;
  }
  template <typename T> T* As() { return reinterpret_cast<T*>(p_); }
 private:
  void* p_ = nullptr;
  size_t cap_ = 0;
};

class VitsModel {
 public:
  // `blob` is the folded float32 weight blob in wetts_blob_tensor_info() order (what
  // wetts_amd.checkpoint.pack_blob writes); scales follow vits_model.cc:44 {0.667, 1.0, 0.8}.
  VitsModel(const wetts_config_t& cfg, const std::vector<float>& blob, int chunk_size = 40,
            int pad_size = 10, uint64_t seed = 0)
      : cfg_(cfg), chunk_size_(chunk_size), pad_size_(pad_size), seed_(seed) {
    if ((int64_t)blob.size() != wetts_blob_numel(&cfg_))
      throw std::runtime_error(std::string("blob size mismatch: ") + wetts_last_error());
    DeviceBuffer tmp;
    float* d = static_cast<float*>(tmp.Reserve(blob.size() * sizeof(float)));
    Check(hipMemcpy(d, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    Check(wetts_create(&cfg_, d, (int64_t)blob.size(), nullptr, &model_), "wetts_create");
    hop_ = wetts_hop_length(model_);
    // The reference class runs the two graphs export_onnx.py wrote, and that script builds its model with
    // hps.model.is_onnx = True (export_onnx.py:59): a Vocos decoder graph ends in OnnxSTFT.inverse
    // (utils/stft.py:325-340), not torch.istft.  This twin stands where those sessions stood, so it computes the
    // same head whatever the config says (HiFi-GAN models have no such switch and ignore the call).
    Check(wetts_set_istft_mode(model_, WETTS_ISTFT_ONNX), "wetts_set_istft_mode");
  }
  ~VitsModel() { wetts_destroy(model_); }
  VitsModel(const VitsModel&) = delete;
  VitsModel& operator=(const VitsModel&) = delete;

  int hop_length() const { return hop_; }
  // WETTS_ISTFT_TORCH: the iSTFT of SynthesizerTrn.infer() in the PyTorch CLI (inference.py) instead of the exported graphs'
  void set_istft_mode(int mode) { Check(wetts_set_istft_mode(model_, mode), "wetts_set_istft_mode"); }
  void set_scales(float noise, float length, float noise_w) {
    noise_scale_ = noise; length_scale_ = length; noise_scale_w_ = noise_w;
  }

  // Non-stream call: text -> audio (float, x32767 like vits_model.cc:84-86).
  void Forward(const std::vector<int64_t>& phonemes, int sid, std::vector<float>* audio) {
    Encode(phonemes, sid);
    audio->resize((size_t)frames_ * hop_);
    Decode(0, frames_, audio->data());
  }

  // Stream call: encode once, then decode chunk by chunk.
  void SetInput(const std::vector<int64_t>& phonemes, int sid) {
    Encode(phonemes, sid);
    cur_ = 0;
    num_chunks_ = (frames_ + chunk_size_ - 1) / chunk_size_;
  }
  // Returns true when all chunks are done (same contract as vits_model.cc:128-153).
  bool StreamDecode(std::vector<float>* audio) {
    audio->clear();
    if (cur_ >= num_chunks_) return true;
    const int start = std::max(0, cur_ * chunk_size_ - pad_size_);
    const int end = std::min((cur_ + 1) * chunk_size_ + pad_size_, frames_);
    std::vector<float> win((size_t)(end - start) * hop_);
    Decode(start, end - start, win.data());
    // Depadding
    const int front = std::min(cur_ * chunk_size_, pad_size_);
    size_t a, b;
    if (cur_ == 0) { a = 0; b = std::min(win.size(), (size_t)chunk_size_ * hop_); }
    else if (cur_ == num_chunks_ - 1) { a = (size_t)front * hop_; b = win.size(); }
    else { a = (size_t)front * hop_; b = (size_t)(front + chunk_size_) * hop_; }
    audio->assign(win.begin() + a, win.begin() + b);
    ++cur_;
    return cur_ >= num_chunks_;
  }
  int frames() const { return frames_; }

 private:
  static void Check(hipError_t e) {
    if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e));
  }
  static void Check(int32_t rc, const char* what) {
    if (rc != WETTS_OK) throw std::runtime_error(std::string(what) + ": " + wetts_last_error());
  }
  // infer_encoder (models.py:282-331): everything up to z = flow^-1(z_p) * y_mask
  void Encode(const std::vector<int64_t>& phonemes, int sid) {
    const int Tx = (int)phonemes.size();
    if (Tx == 0) throw std::runtime_error("empty phoneme sequence");
    const int H = cfg_.hidden_channels, I = cfg_.inter_channels;
    const int gin = cfg_.gin_channels > 0 ? cfg_.gin_channels : 1;
    const int64_t xl = Tx, s64 = sid;
    int64_t* d_x = static_cast<int64_t*>(ids_.Reserve((size_t)(Tx + 2) * 8));
    Check(hipMemcpy(d_x, phonemes.data(), (size_t)Tx * 8, hipMemcpyHostToDevice));
    Check(hipMemcpy(d_x + Tx, &xl, 8, hipMemcpyHostToDevice));
    Check(hipMemcpy(d_x + Tx + 1, &s64, 8, hipMemcpyHostToDevice));
    float* enc = static_cast<float*>(enc_.Reserve(
        sizeof(float) * ((size_t)gin + (size_t)H * Tx + (size_t)2 * I * Tx + 5 * (size_t)Tx) + 64));
    g_ = enc;
    float* x_enc = g_ + gin;
    float* stats = x_enc + (size_t)H * Tx;
    float* x_mask = stats + (size_t)2 * I * Tx;
    float* logw = x_mask + Tx;
    float* w_ceil = logw + Tx;
    float* cum = w_ceil + Tx;
    int64_t wsb = wetts_workspace_bytes(model_, 1, Tx, 0);
    void* ws = ws_.Reserve((size_t)wsb);
    // [0] y_length, [1] low word = WETTS_STATUS_* bits the stage kernels OR in
    int64_t* d_ylen = static_cast<int64_t*>(ylen_.Reserve(16));
    int32_t* d_status = reinterpret_cast<int32_t*>(d_ylen + 1);
    Check(hipMemset(d_ylen, 0, 16));
    Check(wetts_set_status_word(model_, d_status, nullptr), "set_status_word");
    Check(wetts_speaker_embedding(model_, cfg_.n_speakers > 0 ? d_x + Tx + 1 : nullptr, 1, g_,
                                  nullptr), "speaker_embedding");
    const float* gp = cfg_.n_speakers > 0 ? g_ : nullptr;
    Check(wetts_text_encoder(model_, d_x, d_x + Tx, gp, 1, Tx, x_enc, stats, x_mask, ws, wsb,
                             nullptr), "text_encoder");
    if (cfg_.use_sdp) {
      // torch.randn(b, 2, t) of duration_predictors.py:257, drawn on the device
      float* d_eps = static_cast<float*>(epsw_.Reserve((size_t)2 * Tx * 4));
      Check(wetts_randn(d_eps, (int64_t)2 * Tx, seed_, rng_offset_, nullptr), "randn");
      rng_offset_ += ((uint64_t)2 * Tx + 3) / 4;
      Check(wetts_duration_sdp(model_, x_enc, x_mask, gp, d_eps, noise_scale_w_, 1, Tx, logw,
                               d_status, ws, wsb, nullptr), "duration_sdp");
    } else {
      Check(wetts_duration_dp(model_, x_enc, x_mask, gp, 1, Tx, logw, ws, wsb, nullptr),
            "duration_dp");
    }
    Check(wetts_durations_to_lengths(logw, x_mask, length_scale_, 1, Tx, w_ceil, cum, d_ylen,
                                     d_status, nullptr), "durations_to_lengths");
    int64_t back[2] = {0, 0};  // y_length + the status word: one D2H, the one host sync
    Check(hipMemcpy(back, d_ylen, 16, hipMemcpyDeviceToHost));
    Check(wetts_set_status_word(model_, nullptr, nullptr), "set_status_word");
    const int32_t st = (int32_t)(back[1] & 0xffffffff);
    // the reference aborts here too: IndexError from nn.Embedding / glog CHECK in vits_model.cc,
    // `assert (discriminant >= 0).all()` in transforms.py:171
    if (st & (WETTS_STATUS_PHONE_ID_RANGE | WETTS_STATUS_SPEAKER_ID_RANGE))
      throw std::out_of_range("phoneme / speaker id outside the model's embedding tables");
    if (st & (WETTS_STATUS_SPLINE_DOMAIN | WETTS_STATUS_DURATION_NONFINITE))
      throw std::domain_error("duration predictor left its domain (spline discriminant < 0)");
    frames_ = (int)back[0];
    const int Ty = frames_;
    // torch.randn_like(m_p) of models.py:267, drawn on the device
    float* d_epsz = static_cast<float*>(epsz_.Reserve((size_t)I * Ty * 4));
    Check(wetts_randn(d_epsz, (int64_t)I * Ty, seed_, rng_offset_, nullptr), "randn");
    rng_offset_ += ((uint64_t)I * Ty + 3) / 4;
    float* zb = static_cast<float*>(z_.Reserve(sizeof(float) * ((size_t)2 * I * Ty + 2 * (size_t)Ty) + 64));
    zp_ = zb;
    zz_ = zp_ + (size_t)I * Ty;
    y_mask_ = zz_ + (size_t)I * Ty;
    int32_t* f2p = reinterpret_cast<int32_t*>(y_mask_ + Ty);
    Check(wetts_length_regulate(model_, stats, cum, x_mask, d_ylen, d_epsz, (int64_t)I * Ty, Ty,
                                noise_scale_, 1, Tx, Ty, f2p, y_mask_, nullptr, nullptr, nullptr,
                                zp_, nullptr), "length_regulate");
    wsb = wetts_workspace_bytes(model_, 1, Tx, Ty);
    ws = ws_.Reserve((size_t)wsb);
    Check(wetts_flow_reverse(model_, zp_, y_mask_, gp, 1, Ty, zz_, ws, wsb, nullptr),
          "flow_reverse");
    wsb_ = wsb;
  }

  // Generator on frames [start, start+len) of z*y_mask through strides (no copy)
  void Decode(int start, int len, float* host_out) {
    const int Ty = frames_;
    float* d_audio = static_cast<float*>(audio_.Reserve((size_t)len * hop_ * 4));
    const float* gp = cfg_.n_speakers > 0 ? g_ : nullptr;
    Check(wetts_hifigan(model_, zz_ + start, (int64_t)cfg_.inter_channels * Ty, Ty,
                        y_mask_ + start, Ty, gp, 1, len, d_audio, ws_.As<void>(), wsb_, nullptr),
          "hifigan");
    Check(hipMemcpy(host_out, d_audio, (size_t)len * hop_ * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)len * hop_; ++i) host_out[i] *= 32767.f;  // vits_model.cc:84-86
  }

  wetts_config_t cfg_;
  wetts_model_t* model_ = nullptr;
  int hop_ = 256, chunk_size_, pad_size_;
  float noise_scale_ = 0.667f, length_scale_ = 1.0f, noise_scale_w_ = 0.8f;
  uint64_t seed_ = 0, rng_offset_ = 0;  // Philox stream of the two standard-normal draws
  DeviceBuffer ids_, enc_, epsw_, epsz_, z_, ylen_, ws_, audio_;
  float *g_ = nullptr, *zp_ = nullptr, *zz_ = nullptr, *y_mask_ = nullptr;
  int64_t wsb_ = 0;
  int frames_ = 0, cur_ = 0, num_chunks_ = 0;
};

}  // namespace wetts_hip

#endif  // WETTS_VITS_MODEL_HPP_
