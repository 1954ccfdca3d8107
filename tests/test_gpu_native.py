"""GPU tier: a native C++ host (include/wetts_vits_model.hpp, the twin of the reference's
runtime/core/model/vits_model.h) links libwetts_hip.so without Python / torch and reproduces the
Python path's waveform."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_vits_model_forward_and_stream(tmp_path):
    from wetts_amd import SynthesizerTrn, build, config
    exe = os.path.join(build.LIBDIR, "vits_model_main")
    assert os.path.exists(exe), "native test host not built (run __graft_entry__.build())"
    case = util.load_case("tiny_sdp_nonoise")
    cfg, sd, W, blob = util.case_model(case)
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]),
                         **config.MODEL_CONFIGS[str(case["model"])])
    net.load_state_dict(sd).to("cuda")
    n = int(case["x_lengths"][0])
    ph = case["x"][0, :n].astype(np.int64)
    sid = int(case["sid"][0])
    o, *_ = net.infer(torch.from_numpy(ph)[None].cuda(), torch.tensor([n]).cuda(),
                      sid=torch.tensor([sid]).cuda(), noise_scale=0.0, length_scale=1.0,
                      noise_scale_w=0.0)
    expect = o[0, 0].cpu().numpy().astype(np.float32)
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
    r = subprocess.run([exe, str(path)], capture_output=True, text=True, env=env, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
