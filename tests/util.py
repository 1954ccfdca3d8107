"""Shared helpers for the test-suite (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from wetts_amd import checkpoint, config, synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INFER_CASES = ["tiny_sdp_b3", "tiny_dp_b2", "tiny_sdp_nonoise", "tiny_sdp_single", "v1_b2", "v3_b2",
               "tiny_vocos_b2", "vocos_b2",  # VocosGenerator (decoders.py:251-308)
               "tiny_vits2_vocos_b2", "vits2_vocos_b2",  # + VITS2 pre_conv flows (flows.py:95-177)
               "tiny_preconv2_spk_b3",  # pre_conv2 flows (flows.py:16-92) + speaker-conditioned encoder
               # mono-layer flows (flows.py:242-324,391-425): post-residual = the reference's default type, inter
               "tiny_mono_post_b2", "tiny_mono_inter_b3",
               "tiny_sdp_b1_nonoise",  # B = 1, noise-free: the native C++ host's call shape
               # BASELINE-size phoneme counts (make_golden.py BIG_CASES): MFMA / flash attention, the
               # 128x128 and 64x256 conv tiles and fused ResBlock launches with >= 128 time tiles
               "v1_b4x128", "v3_b3x128", "vits2_vocos_b2x64",
               "aishell3_b4x128",  # configs[3]: 218-row speaker table, ragged, sids at both ends
               "tiny_mono_post_b2x64",  # mono-layer flows with the flash attention kernel (~400 frames)
               # examples/*/configs/v2.json (ResBlock1 stages of 64 / 32 / 16 / 8 channels) at 8 and at 128 phonemes
               "v2_b2", "v2_b4x128",
               "vits2_v1_b2",  # examples/baker/configs/vits2_v1.json: pre_conv flows + HiFi-GAN v1
               "stress48k_b2",  # BASELINE configs[4] generator (hop 512, [8,8,4,2]) at f32 vs the live reference
               # is_onnx=True models -- what export_onnx.py builds (export_onnx.py:59): the Vocos head ends in
               # OnnxSTFT.inverse (utils/stft.py:325-340), not torch.istft
               "vocos_onnx_b2", "tiny_vocos_onnx_b2", "tiny_vocos_onnx_b1_nonoise", "vits2_vocos_onnx_b2x64"]
# every committed golden is a GPU parity case (round 5): nothing is held against the oracle only
ORACLE_ONLY_CASES = []
# sub-sampled full-batch fixtures (make_golden.py: every 16th audio sample, every 8th frame of z): own tests
# BASELINE configs[1] / [2] / [4] at their benched batch: 16 x 128 (v1), 64 x 128 two speakers (v3), 16 x 128 (stress48k)
STRIDED_CASES = ["v1_b16x128", "v3_b64x128", "stress48k_b16x128"]
BIG_CASES = ["v1_b4x128", "v3_b3x128", "vits2_vocos_b2x64", "aishell3_b4x128", "tiny_mono_post_b2x64", "v2_b4x128",
             "stress48k_b2", "vits2_vocos_onnx_b2x64"]


def big_case_noise(seed, shape, which):
    """The injected standard-normal draws of the BIG cases: numpy RandomState (a frozen stream),
    exactly as tests/golden/make_golden.py:big_noise fed them to the reference."""
    rs = np.random.RandomState(int(seed) + (0 if which == "w" else 1))
    return rs.standard_normal(shape).astype(np.float32)


def load_case(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    c = {k: d[k] for k in d.files}
    if "noise" in c and str(c["noise"]) == "randomstate":  # compact full-size fixture
        B, Tx = c["x"].shape
        I, Ty = (c["z"].shape[1], c["z"].shape[2]) if "z" in c else (int(c["z_shape"][1]), int(c["z_shape"][2]))
        c["eps_w"] = big_case_noise(c["noise_seed"], (B, 2, Tx), "w")
        c["eps_z"] = big_case_noise(c["noise_seed"], (B, I, Ty), "z")
        shp = tuple(int(v) for v in c["attn_shape"])
        c["attn"] = np.unpackbits(c["attn_bits"], axis=-1)[..., :shp[-1]].reshape(shp)
    return c


def model_dict(case):
    """The `hps.model` dict of a golden case: the named config plus the ctor overrides the fixture records
    (is_onnx=True: the model export_onnx.py builds, export_onnx.py:59)."""
    m = dict(config.MODEL_CONFIGS[str(case["model"])])
    if "is_onnx" in case and int(case["is_onnx"]):
        m["is_onnx"] = True
    return m


def case_model(case):
    """(cfg struct, reference-keyed state_dict, folded weights dict, blob) of a golden case."""
    mname = str(case["model"])
    cfg = config.make_config(model_dict(case), int(case["n_vocab"]), int(case["n_speakers"]))
    sd = synth.make_state_dict(cfg, int(case["weight_seed"]))
    blob = checkpoint.pack_blob(cfg, sd)
    assert abs(synth.blob_checksum(blob) - float(case["blob_checksum"])) <= \
        1e-6 * max(1.0, abs(float(case["blob_checksum"]))), \
        "synthetic weights differ from the ones the golden vectors were generated with"
    W = checkpoint.fold_weight_norm(sd)
    return cfg, sd, W, blob


def cfg_dict(cfg):
    """Plain-dict view of wetts_config_t in the shape oracle/vits_oracle.py expects."""
    nk, nd = cfg.n_resblock_kernels, cfg.n_resblock_dilations
    return dict(
        hidden_channels=cfg.hidden_channels, inter_channels=cfg.inter_channels,
        n_heads=cfg.n_heads, n_layers=cfg.n_layers, kernel_size=cfg.kernel_size,
        window_size=cfg.window_size, n_speakers=cfg.n_speakers, use_sdp=bool(cfg.use_sdp),
        sdp_n_flows=cfg.sdp_n_flows, flow_n_flows=cfg.flow_n_flows,
        flow_wn_layers=cfg.flow_wn_layers, flow_kernel_size=cfg.flow_kernel_size,
        resblock=cfg.resblock, vocoder_type=cfg.vocoder_type, vocos_num_layers=cfg.vocos_num_layers,
        istft_n_fft=cfg.istft_n_fft, istft_hop_length=cfg.istft_hop_length,
        istft_win_length=cfg.istft_win_length, transformer_flows=cfg.transformer_flows,
        use_spk_conditioned_encoder=cfg.use_spk_conditioned_encoder, is_onnx=int(cfg.is_onnx),
        resblock_kernel_sizes=[cfg.resblock_kernel_sizes[j] for j in range(nk)],
        resblock_dilation_sizes=[[cfg.resblock_dilation_sizes[j][i] for i in range(nd)]
                                 for j in range(nk)],
        upsample_rates=[cfg.upsample_rates[i] for i in range(cfg.n_upsamples)],
        upsample_kernel_sizes=[cfg.upsample_kernel_sizes[i] for i in range(cfg.n_upsamples)],
    )


def rms(a):
    a = np.asarray(a, dtype=np.float64)
    return float(np.sqrt(np.mean(a * a))) if a.size else 0.0


def rel_rms(a, ref):
    return rms(np.asarray(a, np.float64) - np.asarray(ref, np.float64)) / max(rms(ref), 1e-30)


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a))
    return x.to(dtype) if dtype is not None else x
