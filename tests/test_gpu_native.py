"""GPU tier: a native C++ host (include/wetts_vits_model.hpp, the twin of the reference's
runtime/core/model/vits_model.h) links libwetts_hip.so without Python / torch and reproduces the
REFERENCE's waveform for the utterance (golden tiny_sdp_b1_nonoise: B = 1, noise scales 0 -- the native host
draws its own noise, so only a noise-free case can be compared with a reference run)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# receptive field of the "tiny" generator, one side, in input frames (decoders.py:17-82 with config.py "tiny":
# conv_pre k7 -> 3; per stage (ups k/stride, then the widest ResBlock1: k = 7, dilations 1, 3, 5 -> 3 * (3 + 9 + 15) / ...):
#   stage 1 (x4): ConvT k8 s4 ~1 frame, ResBlock k7: sum_d [(k-1)d/2 + (k-1)/2] = 3*(1+3+5) + 9 = 36 samples = 9 frames
#   stage 2 (x8): ConvT k4 s2 ~0.5,   ResBlock k7: 36 samples = 4.5 frames;  conv_post k7: 3 samples
# => 3 + 1 + 9 + 0.5 + 4.5 + 0.4 < 19 frames
TINY_RF = 19


# Vocos "tiny_vocos" head: 2 ConvNeXt layers (depthwise k3: 1 frame each), the reflection pad's shift (1), the iSTFT's
# overlap (n_fft / hop / 2 = 2 frames per side) => < 8 frames
TINY_VOCOS_RF = 8


@pytest.mark.parametrize("cname,rf,chunk,pad", [("tiny_sdp_b1_nonoise", TINY_RF, 16, 20), ("tiny_sdp_b1_nonoise", TINY_RF, 16, 12),
                                                ("tiny_sdp_b1_nonoise", TINY_RF, 40, 10),
                                                # a Vocos model: the class stands where the EXPORTED graphs stood, so it must
                                                # reproduce the is_onnx=True reference (OnnxSTFT.inverse) from a plain config
                                                ("tiny_vocos_onnx_b1_nonoise", TINY_VOCOS_RF, 16, 8),
                                                ("tiny_vocos_onnx_b1_nonoise", TINY_VOCOS_RF, 40, 10)])
def test_native_vits_model_forward_and_stream_vs_reference_golden(tmp_path, cname, rf, chunk, pad):
    """pad >= the receptive field: EVERY streamed sample is interior (the streamed waveform equals the one-shot
    decode and the reference golden); pad < RF (incl. the reference's defaults 40 / 10): the samples whose window
    covers their receptive field."""
    from wetts_amd import build
    exe = os.path.join(build.LIBDIR, "vits_model_main")
    assert os.path.exists(exe), "native test host not built (run __graft_entry__.build())"
    case = util.load_case(cname)
    assert case["x"].shape[0] == 1 and tuple(case["scales"]) == (0.0, 1.0, 0.0)
    cfg, sd, W, blob = util.case_model(case)
    cfg.is_onnx = 0  # the native class selects the exported graphs' iSTFT itself (wetts_vits_model.hpp ctor)
    n = int(case["x_lengths"][0])
    ph = case["x"][0, :n].astype(np.int64)
    sid = int(case["sid"][0])
    expect = case["audio"][0, 0].astype(np.float32)  # the reference's infer() output
    path = tmp_path / "case.bin"
    with open(path, "wb") as f:
        f.write(bytes(cfg))
        f.write(np.int64(blob.numel()).tobytes())
        f.write(blob.numpy().astype(np.float32).tobytes())
        f.write(np.int64(n).tobytes())
        f.write(ph.tobytes())
        f.write(np.int64(sid).tobytes())
        f.write(np.int64(expect.size).tobytes())
        f.write(expect.tobytes())
    env = dict(os.environ, LD_LIBRARY_PATH=build.LIBDIR + ":/opt/rocm/lib:" +
               os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe, str(path), str(chunk), str(pad), str(rf)], capture_output=True, text=True,
                       env=env, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    if pad >= rf:  # all samples interior
        fields = dict(kv.split("=") for kv in r.stdout.split() if "=" in kv)
        assert float(fields["all_worst"]) < 1e-4
