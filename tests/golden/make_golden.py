#!/usr/bin/env python3
"""Generates the committed golden vectors by running the REAL reference (wenet-e2e/wetts,
/root/reference, imported unmodified through oracle/ref_import.py) on seeded synthetic
checkpoints.  Run in the build container only:

    python tests/golden/make_golden.py

Fixtures (tests/golden/*.npz) hold inputs, injected noise and the reference's outputs at every
stage boundary of SynthesizerTrn.infer(); weights are NOT stored -- they are regenerated from
(config name, seed) by wetts_amd.synth.make_state_dict, and a checksum of the packed blob is
stored so RNG drift is detected instead of silently mis-pinning.
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_import  # noqa: E402
from wetts_amd import checkpoint, config, synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# name -> (model config, n_vocab, n_speakers, B, Tx, lengths, weight seed, noise seed, scales)
CASES = {
    "tiny_sdp_b3": ("tiny", 40, 3, 3, 12, [12, 7, 9], 11, 101, (0.667, 1.0, 0.8)),
    "tiny_dp_b2": ("tiny_dp", 40, 2, 2, 10, [10, 6], 12, 102, (0.667, 1.1, 0.8)),
    "tiny_sdp_nonoise": ("tiny", 40, 3, 2, 9, [9, 4], 13, 103, (0.0, 1.0, 0.0)),
    "tiny_sdp_single": ("tiny", 40, 1, 1, 1, [1], 14, 104, (0.667, 1.0, 0.8)),
    "v1_b2": ("v1", 64, 1, 2, 8, [8, 5], 21, 201, (0.667, 1.0, 0.8)),
    "v3_b2": ("v3", 64, 2, 2, 8, [8, 6], 22, 202, (0.667, 1.0, 0.8)),
    # examples/*/configs/v2.json: upsample_initial_channel 128 -> ResBlock stages of 64 / 32 / 16 / 8 channels
    "v2_b2": ("v2", 64, 1, 2, 8, [8, 5], 25, 205, (0.667, 1.0, 0.8)),
    # examples/baker/configs/vits2_v1.json: the VITS2 "pre_conv" flows in front of the HiFi-GAN v1 decoder
    "vits2_v1_b2": ("vits2_v1", 64, 1, 2, 8, [8, 6], 26, 206, (0.667, 1.0, 0.8)),
    # VocosGenerator (decoders.py:251-308); iSTFT through the documented torch.istft stand-in
    "tiny_vocos_b2": ("tiny_vocos", 40, 2, 2, 10, [10, 6], 15, 105, (0.667, 1.0, 0.8)),
    "vocos_b2": ("vocos", 64, 2, 2, 8, [8, 5], 23, 203, (0.667, 1.0, 0.8)),
    # VITS2 "pre_conv" transformer flows + SDP + Vocos (examples/baker/configs/vits2_vocos_v1.json)
    "tiny_vits2_vocos_b2": ("tiny_vits2_vocos", 40, 2, 2, 10, [10, 7], 16, 106, (0.667, 1.0, 0.8)),
    "vits2_vocos_b2": ("vits2_vocos_v1", 64, 1, 2, 8, [8, 6], 24, 204, (0.667, 1.0, 0.8)),
    # the two options no checked-in recipe enables: "pre_conv2" flows + speaker-conditioned encoder
    "tiny_preconv2_spk_b3": ("tiny_preconv2_spk", 40, 3, 3, 11, [11, 5, 8], 17, 107, (0.667, 1.0, 0.8)),
    # the mono-layer flow types (flows.py:242-324,391-425); "tiny_mono_post" carries no transformer_flow_type key,
    # i.e. it is the reference's DEFAULT type (models.py:74-75)
    "tiny_mono_post_b2": ("tiny_mono_post", 40, 2, 2, 10, [10, 6], 18, 108, (0.667, 1.0, 0.8)),
    "tiny_mono_inter_b3": ("tiny_mono_inter", 40, 3, 3, 12, [12, 5, 9], 19, 109, (0.667, 1.0, 0.8)),
    # B = 1, noise-free: the call shape of the native C++ host (vits_model.cc:37-87 feeds one utterance)
    "tiny_sdp_b1_nonoise": ("tiny", 40, 3, 1, 14, [14], 20, 110, (0.0, 1.0, 0.0)),
    # The model export_onnx.py builds (`hps['model']['is_onnx'] = True`, export_onnx.py:59): VocosGenerator ends in
    # OnnxSTFT.inverse (utils/stft.py:325-340) instead of torchaudio's InverseSpectrogram (decoders.py:279-283,300-304).
    # Same weights, inputs and noise as vocos_b2 / tiny_vocos_b2 (10th field: ctor overrides), so everything up to the
    # spectrogram is shared and the two heads can be held side by side.  main() also asserts that export_forward -- the
    # function the ONNX graph is traced from (export_onnx.py:82) -- returns this audio.
    "vocos_onnx_b2": ("vocos", 64, 2, 2, 8, [8, 5], 23, 203, (0.667, 1.0, 0.8), dict(is_onnx=True)),
    "tiny_vocos_onnx_b2": ("tiny_vocos", 40, 2, 2, 10, [10, 6], 15, 105, (0.667, 1.0, 0.8), dict(is_onnx=True)),
    # B = 1, noise-free, exported arithmetic: what the native C++ host (the twin of vits_model.cc, which runs the
    # exported graphs) must reproduce for a Vocos model
    "tiny_vocos_onnx_b1_nonoise": ("tiny_vocos", 40, 2, 1, 14, [14], 27, 111, (0.0, 1.0, 0.0), dict(is_onnx=True)),
}
# Full-size cases (BASELINE.json configs[1]/[2] phoneme counts): these switch on the kernels the tiny
# cases never reach -- MFMA text-encoder attention (Tx >= 64), the flash attention of the VITS2 flows,
# 128x128 / 64x256 conv tiles and fused ResBlock pairs with >= 128 time tiles.  To keep the fixtures
# small the two standard-normal draws are INJECTED: torch.randn / torch.randn_like are patched for the
# duration of the reference's infer() to return numpy RandomState(seed) draws (a frozen stream), so the
# tests regenerate eps_w / eps_z from the seed instead of loading them (tests/util.py:big_case_noise).
# Stored: inputs, logw, y_mask, bit-packed attn, z, audio.
BIG_CASES = {
    "v1_b4x128": ("v1", 256, 1, 4, 128, [128, 97, 113, 128], 31, 301, (0.667, 1.0, 0.8)),
    "v3_b3x128": ("v3", 256, 2, 3, 128, [128, 97, 64], 32, 302, (0.667, 1.0, 0.8)),
    "vits2_vocos_b2x64": ("vits2_vocos_v1", 128, 1, 2, 64, [64, 49], 34, 304, (0.667, 1.0, 0.8)),
    # BASELINE.json configs[3] as benched: AISHELL-3 v1 with the 218-row speaker table (the row count is
    # data derived, task.py:229-232), ragged lengths, speaker ids at both ends of the table
    "aishell3_b4x128": ("v1", 256, 218, 4, 128, [128, 57, 100, 33], 35, 305, (0.667, 1.0, 0.8), [0, 57, 217, 3]),
    # mono-layer flows at 64 phonemes (~380 frames): the flash attention kernel inside the flow's Encoder
    "tiny_mono_post_b2x64": ("tiny_mono_post", 64, 2, 2, 64, [64, 41], 36, 326, (0.667, 1.0, 0.8)),
    # examples/*/configs/v2.json at BASELINE phoneme counts: the 64 / 32 / 16 / 8-channel ResBlock1 stages on the
    # 128-row tile selection, the chain kernels and the grouped launches (v2_b2's 8 phonemes never reach those)
    "v2_b4x128": ("v2", 256, 1, 4, 128, [128, 97, 113, 128], 37, 307, (0.667, 1.0, 0.8)),
    # BASELINE.json configs[4] (SURVEY 8d "cfg 5": hop 512 = [8,8,4,2] / [16,16,8,4], 48 kHz): the f32 generator of
    # the stress config against the live reference (the 16-bit modes are held against this f32 path)
    "stress48k_b2": ("stress48k", 256, 1, 2, 48, [48, 31], 38, 308, (0.667, 1.0, 0.8)),
    # BASELINE.json configs[1] AT ITS BENCHED BATCH: 16 x 128 phonemes, all full length (bench.py's shape), through the
    # live reference.  12.6 MB of audio + 9.5 MB of z would not be a small fixture, so the fixture keeps every 16th
    # audio sample and every 8th frame of z (10th / 11th field: strides) beside the full logw / y_mask / bit-packed attn
    "v1_b16x128": ("v1", 256, 1, 16, 128, [128] * 16, 39, 310, (0.667, 0.92, 0.8), None, (16, 8)),
    # the reference's published-metric config (vits2_vocos_v1, runtime/cpu_triton_stream/README.md) AS EXPORTED
    # (is_onnx=True, 12th field: ctor overrides); same weights / inputs / noise as vits2_vocos_b2x64
    "vits2_vocos_onnx_b2x64": ("vits2_vocos_v1", 128, 1, 2, 64, [64, 49], 34, 304, (0.667, 1.0, 0.8), None, None,
                               dict(is_onnx=True)),
    # BASELINE.json configs[2] AT ITS BENCHED BATCH: multilingual v3, 64 x 128 phonemes, two speakers with sid
    # alternating 0 / 1 (SURVEY 8d cfg 3), through the live reference at f32 -- the absolute anchor the bf16 line is
    # held to.  Sub-sampled like v1_b16x128 (every 32nd audio sample, every 16th frame of z).
    # 13th field: the length_scale is calibrated (a scan of +0.0005 steps above the nominal value, recorded in the
    # fixture's `scales`) so that no duration w = exp(logw) * length_scale of the 8192 / 2048 phonemes sits within
    # 1e-4 of an integer: ceil(w) is then the same on any f32 implementation (SURVEY 8d lets the builder pin the
    # length scale), and alignment EQUALITY is a fair demand of the GPU path at this batch.
    "v3_b64x128": ("v3", 256, 2, 64, 128, [128] * 64, 40, 311, (0.667, 1.0, 0.8), [0, 1] * 32, (32, 16), None, True),
    # BASELINE.json configs[4] AT ITS BENCHED BATCH: the 48 kHz stress generator, 16 x 128 phonemes, f32 reference
    "stress48k_b16x128": ("stress48k", 256, 1, 16, 128, [128] * 16, 41, 312, (0.667, 0.92, 0.8), None, (32, 8), None, True),
}
ONLY = os.environ.get("WETTS_GOLDEN_ONLY")  # comma-separated case names (default: all)


def build_reference(model_name, n_vocab, n_speakers, sd, overrides=None):
    SynthesizerTrn, _, _, _ = ref_import.import_reference()
    model = dict(config.MODEL_CONFIGS[model_name], **(overrides or {}))
    with contextlib.redirect_stdout(io.StringIO()):
        net = SynthesizerTrn(n_vocab, 513, 32, n_speakers=n_speakers, **model).eval()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    # everything the synthetic checkpoint does not carry must be outside the infer() path
    # (post_transformer: built by ResidualCouplingTransformersLayer but its use is commented out,
    # flows.py:152-154)
    # dec.stft.*: the two constant buffers OnnxSTFT registers (utils/stft.py:289-290), built by its ctor
    bad = [k for k in missing if not (k.startswith("enc_q.") or k.startswith("dp.post_") or k.startswith("dec.stft.")
                                      or k.startswith("dp.flows.1.") or ".post_transformer." in k)]
    assert not bad, bad
    return net


def run_reference(net, x, x_len, sid, scales, noise_seed):
    """Reference infer() with its two torch.randn draws reproduced from `noise_seed`."""
    ns, ls, nsw = scales
    torch.manual_seed(noise_seed)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        o, attn, y_mask, (z, z_p, m_p, logs_p) = net.infer(
            x, x_len, sid=sid, noise_scale=ns, length_scale=ls, noise_scale_w=nsw)
    return o, attn, y_mask, z, z_p, m_p, logs_p


def stage_probe(net, x, x_len, sid, scales, noise_seed):
    """Re-runs the reference's sub-modules to capture stage boundaries infer() does not return."""
    ns, ls, nsw = scales
    torch.manual_seed(noise_seed)
    with torch.no_grad():
        g = net.emb_g(sid).unsqueeze(-1) if net.n_speakers > 0 else None
        xe, m_p, logs_p, x_mask = net.enc_p(x, x_len, g=g)
        if net.use_sdp:
            logw = net.dp(xe, x_mask, g=g, reverse=True, noise_scale=nsw)
        else:
            logw = net.dp(xe, x_mask, g=g)
    return xe, m_p, logs_p, x_mask, logw


def big_noise(seed, shape, which):
    """The frozen noise stream of the full-size cases (numpy RandomState never changes)."""
    rs = np.random.RandomState(seed + (0 if which == "w" else 1))
    return torch.from_numpy(rs.standard_normal(shape).astype(np.float32))


def run_big_cases():
    import unittest.mock as mock
    for name, spec in BIG_CASES.items():
        mname, n_vocab, n_spk, B, Tx, lens, wseed, nseed, scales = spec[:9]
        if ONLY and name not in ONLY.split(","):
            continue
        ov = spec[11] if len(spec) > 11 else None  # ctor overrides (is_onnx=True)
        cfg = config.make_config(dict(config.MODEL_CONFIGS[mname], **(ov or {})), n_vocab, n_spk)
        sd = synth.make_state_dict(cfg, wseed)
        blob = checkpoint.pack_blob(cfg, sd)
        net = build_reference(mname, n_vocab, n_spk, sd, ov)
        gi = torch.Generator().manual_seed(nseed + 7)
        x = torch.randint(0, n_vocab, (B, Tx), generator=gi)
        x_len = torch.tensor(lens, dtype=torch.long)
        sid = torch.randint(0, n_spk, (B,), generator=gi)
        if len(spec) > 9 and spec[9] is not None:  # explicit speaker ids
            sid = torch.tensor(spec[9], dtype=torch.long)
        strides = spec[10] if len(spec) > 10 else None  # (audio stride, z frame stride) of a sub-sampled fixture
        ns, ls, nsw = scales
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def fake_randn(*size, **kw):
            size = tuple(size[0]) if len(size) == 1 and not isinstance(size[0], int) else size
            assert size == (B, 2, Tx), size  # duration_predictors.py:257
            return big_noise(nseed, size, "w")

        def fake_randn_like(t, **kw):
            assert t.dim() == 3 and t.shape[0] == B and t.shape[1] == cfg.inter_channels  # models.py:267
            return big_noise(nseed, tuple(t.shape), "z")

        if len(spec) > 12 and spec[12]:  # calibrate length_scale for the widest ceil() margin (see BIG_CASES)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), \
                    mock.patch.object(torch, "randn", fake_randn):
                g0 = net.emb_g(sid).unsqueeze(-1) if net.n_speakers > 0 else None
                xe0, _, _, xm0 = net.enc_p(x, x_len, g=g0)
                lw0 = net.dp(xe0, xm0, g=g0, reverse=True, noise_scale=nsw) if net.use_sdp else net.dp(xe0, xm0, g=g0)
            best = None
            for k in range(41):
                cand = float(np.float32(ls + 0.0005 * k))
                w0 = torch.exp(lw0) * xm0 * cand
                f0 = (torch.ceil(w0) - w0)[xm0 > 0]
                m0 = float(torch.minimum(f0, 1 - f0).min())
                if best is None or m0 > best[0]:
                    best = (m0, cand)
            ls = best[1]
            scales = (ns, ls, nsw)
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), \
                mock.patch.object(torch, "randn", fake_randn), \
                mock.patch.object(torch, "randn_like", fake_randn_like):
            o, attn, y_mask, (z, z_p, m_pe, logs_pe) = net.infer(
                x, x_len, sid=sid, noise_scale=ns, length_scale=ls, noise_scale_w=nsw)
            g = net.emb_g(sid).unsqueeze(-1) if net.n_speakers > 0 else None
            xe, m_p, logs_p, x_mask = net.enc_p(x, x_len, g=g)
            logw = net.dp(xe, x_mask, g=g, reverse=True, noise_scale=nsw) if net.use_sdp \
                else net.dp(xe, x_mask, g=g)
        assert torch.randn is real_randn and torch.randn_like is real_randn_like
        w = torch.exp(logw) * x_mask * ls
        frac = (torch.ceil(w) - w)[x_mask > 0]
        margin = float(torch.minimum(frac, 1 - frac).min())
        Ty = z.shape[2]
        full = dict(z=z.numpy(), audio=o.numpy())
        if strides:  # sub-sampled fixture: strided views + the full tensors' shapes and float64 sums
            sa, sz = strides
            full = dict(audio_sub=o.numpy()[..., ::sa].copy(), z_sub=z.numpy()[..., ::sz].copy(),
                        sub_strides=np.array([sa, sz]), audio_shape=np.array(o.shape), z_shape=np.array(z.shape),
                        audio_sum=float(o.double().sum()), audio_sqsum=float(o.double().pow(2).sum()))
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            model=mname, n_vocab=n_vocab, n_speakers=n_spk, weight_seed=wseed, noise_seed=nseed,
            noise="randomstate", scales=np.array(scales, np.float64), is_onnx=int(cfg.is_onnx),
            blob_checksum=synth.blob_checksum(blob),
            x=x.numpy(), x_lengths=x_len.numpy(), sid=sid.numpy(), x_mask=x_mask.numpy(),
            # (the encoder's stage boundaries are pinned at this phoneme count by v1_b4x128; a sub-sampled fixture
            # leaves them out to stay small)
            **({} if strides else dict(x_enc=xe.numpy(), m_p=m_p.numpy(), logs_p=logs_p.numpy())),
            logw=logw.numpy(), ceil_margin=margin,
            attn_bits=np.packbits(attn.numpy().astype(np.uint8), axis=-1),
            attn_shape=np.array(attn.shape), y_mask=y_mask.numpy(), **full)
        print(f"{name}: Ty={Ty} audio={tuple(o.shape)} rms={float(o.pow(2).mean().sqrt()):.4f} "
              f"ceil_margin={margin:.2e} frames/phone={float(y_mask.sum() / x_mask.sum()):.2f}")


def main():
    if not ref_import.available():
        raise SystemExit("reference not present; golden vectors can only be generated in the "
                         "build container")
    torch.set_num_threads(1)
    for name, spec in CASES.items():
        mname, n_vocab, n_spk, B, Tx, lens, wseed, nseed, scales = spec[:9]
        ov = spec[9] if len(spec) > 9 else None  # ctor overrides (is_onnx=True)
        if ONLY and name not in ONLY.split(","):
            continue
        cfg = config.make_config(dict(config.MODEL_CONFIGS[mname], **(ov or {})), n_vocab, n_spk)
        sd = synth.make_state_dict(cfg, wseed)
        blob = checkpoint.pack_blob(cfg, sd)
        net = build_reference(mname, n_vocab, n_spk, sd, ov)
        gi = torch.Generator().manual_seed(nseed + 7)
        x = torch.randint(0, n_vocab, (B, Tx), generator=gi)
        x_len = torch.tensor(lens, dtype=torch.long)
        sid = torch.randint(0, n_spk, (B,), generator=gi)
        # the noise the reference will draw: randn(B,2,Tx) then randn_like(m_p) [B,192,Ty]
        torch.manual_seed(nseed)
        eps_w = torch.randn(B, 2, Tx) if cfg.use_sdp else torch.zeros(B, 2, Tx)
        o, attn, y_mask, z, z_p, m_pe, logs_pe = run_reference(net, x, x_len, sid, scales, nseed)
        if ov and ov.get("is_onnx"):
            # the function the exported graph is traced from (export_onnx.py:82): same draws, same audio
            torch.manual_seed(nseed)
            with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
                oe = net.export_forward(x, x_len, torch.tensor([list(scales)] * B), sid)
            assert torch.equal(oe, o), "export_forward differs from infer() on the is_onnx model"
        Ty = z.shape[2]
        torch.manual_seed(nseed)
        if cfg.use_sdp:
            torch.randn(B, 2, Tx)
        # randn_like(m_p): m_p is a transposed VIEW of a [B,Ty,C] matmul result (models.py:262-267)
        # and torch's CPU normal_ consumes the generator differently for strided tensors, so the
        # draw must be reproduced on a tensor with the same strides.
        eps_z = torch.randn_like(torch.empty(B, Ty, cfg.inter_channels).transpose(1, 2)).contiguous()
        xe, m_p, logs_p, x_mask, logw = stage_probe(net, x, x_len, sid, scales, nseed)
        # margin of ceil(): distance of w to the nearest integer from below
        w = torch.exp(logw) * x_mask * scales[1]
        frac = (torch.ceil(w) - w)[x_mask > 0]
        margin = float(torch.minimum(frac, 1 - frac).min()) if frac.numel() else 1.0
        # folded weight-norm check against the reference's own remove_weight_norm
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"),
            model=mname, n_vocab=n_vocab, n_speakers=n_spk, weight_seed=wseed, noise_seed=nseed,
            scales=np.array(scales, np.float64), blob_checksum=synth.blob_checksum(blob), is_onnx=int(cfg.is_onnx),
            x=x.numpy(), x_lengths=x_len.numpy(), sid=sid.numpy(),
            eps_w=eps_w.numpy(), eps_z=eps_z.numpy(),
            x_enc=xe.numpy(), m_p=m_p.numpy(), logs_p=logs_p.numpy(), x_mask=x_mask.numpy(),
            logw=logw.numpy(), ceil_margin=margin,
            attn=attn.numpy().astype(np.uint8), y_mask=y_mask.numpy(), z=z.numpy(),
            z_p=z_p.numpy(), m_p_exp=m_pe.numpy(), logs_p_exp=logs_pe.numpy(),
            audio=o.numpy())
        rms = float(o.pow(2).mean().sqrt())
        print(f"{name}: Ty={Ty} audio={tuple(o.shape)} rms={rms:.4f} ceil_margin={margin:.2e} "
              f"frames/phone={float(y_mask.sum() / x_mask.sum()):.2f}")

    torch.set_num_threads(8)
    run_big_cases()
    if ONLY:
        if "vocos_onnx_stream_kat" in ONLY.split(","):
            vocos_onnx_stream_kat()
        return
    # MAS known-answer vectors from the reference's maximum_path (numba stub => plain Python)
    _, _, _, mas = ref_import.import_reference()
    g = torch.Generator().manual_seed(5)
    cases = []
    for (b, ty, tx) in [(3, 9, 5), (2, 16, 16), (4, 23, 7), (1, 1, 1), (2, 12, 1)]:
        neg = torch.randn(b, ty, tx, generator=g)
        t_y = torch.randint(max(1, ty // 2), ty + 1, (b,), generator=g)
        t_x = torch.minimum(torch.randint(1, tx + 1, (b,), generator=g), t_y)
        mask = (torch.arange(ty).view(1, ty, 1) < t_y.view(b, 1, 1)) & \
               (torch.arange(tx).view(1, 1, tx) < t_x.view(b, 1, 1))
        path = mas.maximum_path(neg, mask.float())
        cases.append((neg.numpy(), t_y.numpy().astype(np.int32), t_x.numpy().astype(np.int32),
                      path.numpy().astype(np.int8)))
    # tie case: constant scores exercise the strict `<` in the backtrack
    neg = torch.zeros(1, 6, 3)
    mask = torch.ones(1, 6, 3)
    cases.append((neg.numpy(), np.array([6], np.int32), np.array([3], np.int32),
                  mas.maximum_path(neg, mask).numpy().astype(np.int8)))
    np.savez_compressed(os.path.join(OUT, "mas_kat.npz"), n=len(cases),
                        **{f"neg{i}": c[0] for i, c in enumerate(cases)},
                        **{f"ty{i}": c[1] for i, c in enumerate(cases)},
                        **{f"tx{i}": c[2] for i, c in enumerate(cases)},
                        **{f"path{i}": c[3] for i, c in enumerate(cases)})
    print("mas_kat:", len(cases), "cases")

    # generate_path edge cases (commons.py:120-136), incl. zero durations and single phoneme
    _, _, commons, _ = ref_import.import_reference()
    durs = [[2, 0, 3, 1], [0, 0, 0, 0], [1, 1, 1, 1], [5, 0, 0, 0]]
    d = torch.tensor(durs, dtype=torch.float32).unsqueeze(1)
    ylen = torch.clamp_min(d.sum([1, 2]), 1).long()
    y_mask = commons.sequence_mask(ylen, None).unsqueeze(1).float()
    x_mask = torch.ones(4, 1, 4)
    attn = commons.generate_path(d, x_mask.unsqueeze(2) * y_mask.unsqueeze(-1))
    np.savez_compressed(os.path.join(OUT, "generate_path_kat.npz"), durations=d.numpy(),
                        y_lengths=ylen.numpy(), attn=attn.numpy())
    print("generate_path_kat: ok")
    chunk_kat()
    chunk_kat_triton()
    vocos_onnx_stream_kat()


def vocos_onnx_stream_kat():
    """The reference's streaming client on an EXPORTED Vocos model, end to end: the is_onnx=True reference module stands
    for the two graphs (encoder = export_encoder_forward, decoder = export_decoder_forward, export_onnx.py:94,127),
    and the chunk loop is inference_onnx.py:146-158 with its own get_chunks / depadding lifted from the file's AST.
    Windows through OnnxSTFT.inverse have edges unlike torch.istft's (no envelope division), so what the
    overlap-discard margins leave of them is part of the answer.  Stored: z [1,L,192] of one utterance, sid, the
    streamed audio for (block, pad) = (40, 10) and (16, 4), and the one-window decode."""
    import ast
    import math
    src = open(os.path.join(ref_import.REF_VITS, "inference_onnx.py")).read()
    fns = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in ("get_chunks", "depadding")]
    ns = {"math": math, "np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "inference_onnx.py", "exec"), ns)
    mname, n_vocab, n_spk, wseed, nseed = "vits2_vocos_v1", 64, 1, 42, 313
    ov = dict(is_onnx=True)
    cfg = config.make_config(dict(config.MODEL_CONFIGS[mname], **ov), n_vocab, n_spk)
    sd = synth.make_state_dict(cfg, wseed)
    blob = checkpoint.pack_blob(cfg, sd)
    net = build_reference(mname, n_vocab, n_spk, sd, ov)
    gi = torch.Generator().manual_seed(nseed + 7)
    x = torch.randint(0, n_vocab, (1, 17), generator=gi)
    x_len = torch.tensor([17])
    sid = torch.zeros(1, dtype=torch.long)
    scales = torch.tensor([[0.667, 1.0, 0.8]])
    torch.manual_seed(nseed)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        z = net.export_encoder_forward(x, x_len, scales, sid)  # [1, L, 192]
        whole = net.export_decoder_forward(z, sid)
        hop = cfg.istft_hop_length
        streams = {}
        for block, pad in ((40, 10), (16, 4)):
            chunks = ns["get_chunks"](z.numpy(), block, pad)
            pieces = []
            for i, ch in enumerate(chunks):
                a = net.export_decoder_forward(torch.from_numpy(np.ascontiguousarray(ch)), sid)[0].numpy()
                pieces.append(ns["depadding"](a, len(chunks), i, block, pad, hop))
            streams[f"stream_{block}_{pad}"] = np.concatenate(pieces, axis=1)
    L = z.shape[1]
    assert all(v.shape[1] == L * hop for v in streams.values())
    np.savez_compressed(os.path.join(OUT, "vocos_onnx_stream_kat.npz"), model=mname, n_vocab=n_vocab, n_speakers=n_spk,
                        weight_seed=wseed, blob_checksum=synth.blob_checksum(blob), is_onnx=1, z=z.numpy(),
                        sid=sid.numpy(), whole=whole.numpy(), **streams)
    print(f"vocos_onnx_stream_kat: L={L} frames, {L * hop} samples, whole-vs-stream rms "
          + ", ".join(f"{k}: {float(np.sqrt(np.mean((v - whole[0].numpy()) ** 2))):.2e}" for k, v in streams.items()))


def chunk_kat():
    """Known answers of the reference's streaming helpers get_chunks / depadding
    (wetts/vits/inference_onnx.py:37-76): the two function definitions are lifted out of the file's
    AST and executed as they are (the module itself imports onnxruntime, which is absent)."""
    import ast
    import math
    src = open(os.path.join(ref_import.REF_VITS, "inference_onnx.py")).read()
    tree = ast.parse(src)
    fns = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("get_chunks", "depadding")]
    assert len(fns) == 2
    ns = {"math": math, "np": np}
    exec(compile(ast.Module(body=fns, type_ignores=[]), "inference_onnx.py", "exec"), ns)
    rows = []
    for L in (1, 7, 39, 40, 41, 50, 79, 80, 81, 100, 120, 333):
        for block, pad in ((40, 10), (16, 4), (25, 0), (40, 60), (-1, 10)):
            hop = 256 if block != 16 else 64
            mel = np.arange(L, dtype=np.int64).reshape(1, L, 1)
            with contextlib.redirect_stdout(io.StringIO()):
                chunks = ns["get_chunks"](mel, block, pad)
            for i, ch in enumerate(chunks):
                a, b = int(ch[0, 0, 0]), int(ch[0, -1, 0]) + 1
                n = (b - a) * hop
                audio = np.arange(n, dtype=np.int64).reshape(1, n)
                if block == -1:  # one window, nothing to discard (inference_onnx.py:152-158)
                    lo, hi = 0, n
                else:
                    kept = ns["depadding"](audio, len(chunks), i, block, pad, hop)
                    lo, hi = (int(kept[0, 0]), int(kept[0, -1]) + 1) if kept.shape[1] else (0, 0)
                rows.append((L, block, pad, hop, len(chunks), i, a, b, lo, hi))
    np.savez_compressed(os.path.join(OUT, "chunk_kat.npz"), rows=np.array(rows, np.int64))
    print("chunk_kat:", len(rows), "windows")


def chunk_kat_triton():
    """Known answers of the Triton streaming twin's get_chunks / depadding
    (runtime/cpu_triton_stream/model_repo/stream_tts/1/model.py:58-111, MIN_CHUNK = 65 and the reflect padding of
    a short last window), lifted out of the file's AST and executed as they are (the module imports
    triton_python_backend_utils / pypinyin / tn, all absent).  One row per window:
    (L, block, pad, hop, n_chunks, i, win_start, win_len_with_padding, pad_end or -1, index_checksum, lo, hi,
    raised) -- `index_checksum` = sum((pos + 1) * frame_index) over the window incl. its reflected frames, `raised`
    = 1 where the reference's depadding raises TypeError (`-None`, model.py:105)."""
    import ast
    path = os.path.join(ref_import.REF_VITS, "..", "..", "runtime", "cpu_triton_stream", "model_repo", "stream_tts",
                        "1", "model.py")
    tree = ast.parse(open(path).read())
    keep = [n for n in tree.body
            if (isinstance(n, ast.FunctionDef) and n.name in ("get_chunks", "depadding"))
            or (isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") in
                ("MIN_CHUNK", "VOC_BLOCK_SIZE", "VOC_PAD_SIZE", "UPSAMPLE_SIZE"))]
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), "stream_tts/1/model.py", "exec"), ns)
    assert (ns["MIN_CHUNK"], ns["VOC_BLOCK_SIZE"], ns["VOC_PAD_SIZE"], ns["UPSAMPLE_SIZE"]) == (65, 70, 10, 256)
    rows = []
    for L in (2, 30, 64, 65, 66, 69, 70, 71, 80, 124, 125, 139, 140, 141, 150, 200, 210, 333):
        for block, pad in ((70, 10), (40, 10), (16, 4), (100, 20)):
            hop = 256 if block != 16 else 64
            mel = np.arange(L, dtype=np.int64).reshape(1, 1, L)  # (B, freq, Frame): the Triton twin's z layout
            chunks, pad_end = ns["get_chunks"](mel, block, pad)
            for i, ch in enumerate(chunks):
                idx = ch[0, 0]
                n = idx.shape[0] * hop
                audio = np.arange(n, dtype=np.int64).reshape(1, n)
                raised, lo, hi = 0, 0, 0
                try:
                    kept = ns["depadding"](audio, len(chunks), i, block, pad, hop, pad_end)
                    lo, hi = (int(kept[0, 0]), int(kept[0, -1]) + 1) if kept.shape[1] else (0, 0)
                except TypeError:
                    raised = 1
                cks = int(((np.arange(idx.shape[0]) + 1) * idx).sum())
                rows.append((L, block, pad, hop, len(chunks), i, int(idx[0]), idx.shape[0],
                             -1 if pad_end is None else int(pad_end), cks, lo, hi, raised))
    np.savez_compressed(os.path.join(OUT, "chunk_kat_triton.npz"), rows=np.array(rows, np.int64))
    print("chunk_kat_triton:", len(rows), "windows,", sum(r[-1] for r in rows), "where the reference raises")


if __name__ == "__main__":
    main()
