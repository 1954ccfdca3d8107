// Native-host check of include/wetts_vits_model.hpp (the C++ twin of the reference's
// runtime/core/model/vits_model.h).  Reads a case written by tests/test_gpu_native.py:
//   case.bin = [wetts_config_t][int64 n_blob][float blob...][int64 n_ph][int64 ph...][int64 sid]
//              [int64 n_audio][float expected_audio...]   (expected = the REFERENCE's golden audio for
//              this utterance, tests/golden/tiny_sdp_b1_nonoise.npz; noise scales 0 so the run is deterministic)
//   argv: case.bin chunk pad rf   (rf = the generator's receptive field in frames, one side)
// Gates: Forward() vs the golden (RMS < 1e-4); the streamed concatenation has the one-shot's length; every
// streamed sample whose window reaches rf frames to both sides (or ends at the utterance's own edge) equals the
// one-shot decode and the golden to 1e-4 -- the overlap-discard protocol of vits_model.cc:96-153.
// Prints "OK ..." and exits 0 on success.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "wetts_vits_model.hpp"

template <typename T>
static void rd(std::ifstream& f, T* p, size_t n) {
  f.read(reinterpret_cast<char*>(p), sizeof(T) * n);
  if (!f) { std::fprintf(stderr, "short read\n"); std::exit(2); }
}

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s case.bin\n", argv[0]); return 2; }
  std::ifstream f(argv[1], std::ios::binary);
  wetts_config_t cfg;
  rd(f, &cfg, 1);
  int64_t n;
  rd(f, &n, 1);
  std::vector<float> blob(n);
  rd(f, blob.data(), n);
  rd(f, &n, 1);
  std::vector<int64_t> ph(n);
  rd(f, ph.data(), n);
  int64_t sid;
  rd(f, &sid, 1);
  rd(f, &n, 1);
  std::vector<float> expect(n);
  rd(f, expect.data(), n);
  const int chunk = argc > 2 ? std::atoi(argv[2]) : 16, pad = argc > 3 ? std::atoi(argv[3]) : 12;
  const int rf = argc > 4 ? std::atoi(argv[4]) : 20;
  try {
    wetts_hip::VitsModel model(cfg, blob, chunk, pad);
    model.set_scales(0.f, 1.f, 0.f);
    std::vector<float> audio;
    model.Forward(ph, (int)sid, &audio);
    if (audio.size() != expect.size()) {
      std::printf("FAIL size %zu vs %zu\n", audio.size(), expect.size());
      return 1;
    }
    double se = 0;
    for (size_t i = 0; i < audio.size(); ++i) {
      double d = audio[i] / 32767.0 - expect[i];
      se += d * d;
    }
    const double rms = std::sqrt(se / audio.size());
    // streaming: SetInput once, StreamDecode until done (vits_model.cc:128-153)
    model.SetInput(ph, (int)sid);
    std::vector<float> cat, piece;
    bool done = false;
    int calls = 0;
    while (!done && calls < 1000) { done = model.StreamDecode(&piece); cat.insert(cat.end(), piece.begin(), piece.end()); ++calls; }
    const bool stream_ok = cat.size() == audio.size();
    const int frames = model.frames(), hop = model.hop_length();
    const int nchunks = (frames + chunk - 1) / chunk;
    double worst_in = 0, worst_in_gold = 0, worst_all = 0;
    long n_in = 0;
    if (stream_ok) {
      for (int t = 0; t < frames; ++t) {
        const int c = t / chunk;  // the chunk that emits frame t
        const int ws = std::max(0, c * chunk - pad), we = std::min((c + 1) * chunk + pad, frames);
        const bool interior = (ws == 0 || t - ws >= rf) && (we == frames || we - 1 - t >= rf);
        for (int k = 0; k < hop; ++k) {
          const size_t i = (size_t)t * hop + k;
          const double d = std::fabs((double)cat[i] - audio[i]) / 32767.0;
          worst_all = std::max(worst_all, d);
          if (interior) {
            worst_in = std::max(worst_in, d);
            worst_in_gold = std::max(worst_in_gold, std::fabs((double)cat[i] / 32767.0 - expect[i]));
            ++n_in;
          }
        }
      }
    }
    const bool ok = rms < 1e-4 && stream_ok && calls == nchunks && n_in > 0 && worst_in < 1e-4 && worst_in_gold < 1e-3;
    std::printf("%s rms_vs_reference=%.3e frames=%d stream_calls=%d stream_ok=%d interior_samples=%ld of %zu "
                "interior_worst=%.3e interior_worst_vs_reference=%.3e all_worst=%.3e\n",
                ok ? "OK" : "FAIL", rms, frames, calls, (int)stream_ok, n_in, cat.size(), worst_in, worst_in_gold,
                worst_all);
    return ok ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 1;
  }
}
