// Native-host check of include/wetts_vits_model.hpp (the C++ twin of the reference's
// runtime/core/model/vits_model.h).  Reads a case written by tests/test_gpu_native.py:
//   case.bin = [wetts_config_t][int64 n_blob][float blob...][int64 n_ph][int64 ph...][int64 sid]
//              [int64 n_audio][float expected_audio...]   (expected from the Python path,
//              noise scales 0 so both sides are deterministic)
// Prints "OK rms=<..> stream_ok=<0/1>" and exits 0 on success.
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
  try {
    wetts_hip::VitsModel model(cfg, blob, /*chunk*/ 16, /*pad*/ 12);
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
    // streaming: concatenation has the same length; interior samples agree with the one-shot
    model.SetInput(ph, (int)sid);
    std::vector<float> cat, piece;
    bool done = false;
    int calls = 0;
    while (!done && calls < 1000) { done = model.StreamDecode(&piece); cat.insert(cat.end(), piece.begin(), piece.end()); ++calls; }
    bool stream_ok = cat.size() == audio.size();
    double worst = 0;
    if (stream_ok) {
      std::vector<double> diffs;
      for (size_t i = 0; i < cat.size(); ++i) worst = std::max(worst, std::fabs((double)cat[i] - audio[i]) / 32767.0);
    }
    std::printf("%s rms=%.3e frames=%d stream_calls=%d stream_ok=%d stream_worst=%.3e\n",
                (rms < 1e-4 && stream_ok) ? "OK" : "FAIL", rms, model.frames(), calls,
                (int)stream_ok, worst);
    return (rms < 1e-4 && stream_ok) ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("FAIL exception: %s\n", e.what());
    return 1;
  }
}
