"""Pins oracle/vits_oracle.py against the golden vectors the REAL reference produced
(tests/golden/make_golden.py) -- CPU tier.  Tolerances are float32 round-off: both sides run the
same ATen CPU kernels, only the relative-attention / generate_path / spline formulations differ."""
import numpy as np
import pytest
import torch

from oracle import vits_oracle as vo
from tests import util


@pytest.mark.parametrize("name", util.INFER_CASES)
def test_infer_matches_reference(name):
    case = util.load_case(name)
    cfg, _, W, _ = util.case_model(case)
    cd = util.cfg_dict(cfg)
    ns, ls, nsw = [float(v) for v in case["scales"]]
    st = vo.infer(W, cd, util.t(case["x"]), util.t(case["x_lengths"]), util.t(case["sid"]),
                  noise_scale=ns, length_scale=ls, noise_scale_w=nsw,
                  eps_w=util.t(case["eps_w"]), eps_z=util.t(case["eps_z"]), return_stages=True)
    assert util.rel_rms(st["x"].numpy(), case["x_enc"]) < 2e-5
    assert util.rel_rms(st["m_p"].numpy(), case["m_p"]) < 2e-5
    dlw = float(np.abs(st["logw"].numpy() - case["logw"]).max())
    assert dlw < 2e-4
    # ceil() margin stored with the fixture: the distance of w = exp(logw)*mask*length_scale to the
    # nearest integer; equality of y_mask / attn is only a fair demand while the observed error in w
    # stays an order of magnitude inside it
    w_err = float(np.abs(np.exp(st["logw"].numpy()) - np.exp(case["logw"])).max()) * ls
    assert float(case["ceil_margin"]) > 10 * w_err, (float(case["ceil_margin"]), w_err)
    assert np.array_equal(st["y_mask"].numpy(), case["y_mask"])
    assert np.array_equal(st["attn"].numpy().astype(np.uint8), case["attn"])
    if "z_p" in case:  # the compact full-size fixtures keep z and audio only
        assert util.rel_rms(st["z_p"].numpy(), case["z_p"]) < 2e-5
    assert util.rel_rms(st["z"].numpy(), case["z"]) < 5e-5
    # the north-star gate is 1e-3 absolute RMS; the oracle sits ~100x inside it
    assert util.rms(st["o"].numpy() - case["audio"]) < 2e-5
    assert util.rel_rms(st["o"].numpy(), case["audio"]) < 2e-4


@pytest.mark.parametrize("name", util.STRIDED_CASES)
def test_infer_matches_reference_at_the_benched_batch(name):
    """BASELINE configs[1] / [2] / [4] at their benched batch (the sub-sampled fixtures v1_b16x128, v3_b64x128,
    stress48k_b16x128): the oracle against the live reference's durations, alignment and the strided z / audio views.
    configs[1] decodes in full (~25 s of CPU); the two others -- whose decoders are pinned by their small-batch goldens --
    run every stage up to z in full and the decoder on the first 96 frames (audio compared over the first 48, in front of
    the cut's receptive field), which keeps the CPU tier to minutes."""
    case = util.load_case(name)
    cfg, _, W, _ = util.case_model(case)
    ns, ls, nsw = [float(v) for v in case["scales"]]
    torch.set_num_threads(8)
    full = name == "v1_b16x128"
    st = vo.infer(W, util.cfg_dict(cfg), util.t(case["x"]), util.t(case["x_lengths"]), util.t(case["sid"]),
                  noise_scale=ns, length_scale=ls, noise_scale_w=nsw, eps_w=util.t(case["eps_w"]),
                  eps_z=util.t(case["eps_z"]), return_stages=True, max_len=None if full else 96)
    sa, sz = (int(v) for v in case["sub_strides"])
    w_err = float(np.abs(np.exp(st["logw"].numpy()) - np.exp(case["logw"])).max()) * ls
    assert float(case["ceil_margin"]) > 10 * w_err
    assert np.array_equal(st["y_mask"].numpy(), case["y_mask"])
    assert np.array_equal(st["attn"].numpy().astype(np.uint8), case["attn"])
    assert util.rel_rms(st["z"].numpy()[..., ::sz], case["z_sub"]) < 5e-5
    if full:
        assert util.rms(st["o"].numpy()[..., ::sa] - case["audio_sub"]) < 2e-5
        assert abs(float(st["o"].double().pow(2).sum()) / float(case["audio_sqsum"]) - 1.0) < 1e-5
    else:
        hop = int(case["audio_shape"][-1]) // int(case["z_shape"][-1])
        n = 48 * hop // sa  # strided samples of the first 48 frames
        assert util.rms(st["o"].numpy()[..., ::sa][..., :n] - case["audio_sub"][..., :n]) < 2e-5


def test_onnx_stft_head_is_0375_of_torch_istft_in_the_interior_and_differs_at_the_edges():
    """The two Vocos heads side by side on the SAME model and noise (vocos_b2 / vocos_onnx_b2 share weights, inputs and
    draws): OnnxSTFT.inverse has no window-envelope division, so at hop = n_fft / 4 (sum of hann^2 = 1.5, scale = 4) its
    interior is 0.375 x the torch.istft audio; the first / last n_fft/2 samples differ in shape."""
    a, b = util.load_case("vocos_b2"), util.load_case("vocos_onnx_b2")
    assert np.array_equal(a["z"], b["z"]) and int(b["is_onnx"]) == 1
    nf = 1024
    ta, tb = a["audio"][0, 0], b["audio"][0, 0]
    assert util.rel_rms(tb[nf:-nf], 0.375 * ta[nf:-nf]) < 1e-5
    assert util.rel_rms(tb[:nf // 2], 0.375 * ta[:nf // 2]) > 1e-2


def test_oracle_streams_the_exported_vocos_model_like_the_references_client():
    """vocos_onnx_stream_kat.npz (the reference's chunk loop over export_decoder_forward of an is_onnx module): the
    oracle's decoder on the windows of session.get_chunks, cut by session.depad_bounds, reproduces the stream."""
    from wetts_amd import session
    case = util.load_case("vocos_onnx_stream_kat")
    cfg, _, W, _ = util.case_model(case)
    cd = util.cfg_dict(cfg)
    assert cd["is_onnx"] == 1
    z = util.t(case["z"]).transpose(1, 2)
    g = torch.nn.functional.embedding(util.t(case["sid"]), W["emb_g.weight"]).unsqueeze(-1)
    hop = cfg.istft_hop_length
    with torch.no_grad():
        assert util.rms(vo.decoder(W, cd, z, g).numpy() - case["whole"]) < 2e-6
        for block, pad in ((40, 10), (16, 4)):
            wins = session.get_chunks(z.shape[2], block, pad)
            pieces = []
            for i, (ws, we) in enumerate(wins):
                a = vo.decoder(W, cd, z[:, :, ws:we], g).numpy().reshape(1, -1)
                lo, hi = session.depad_bounds(len(wins), i, block, pad, hop, a.shape[1])
                pieces.append(a[:, lo:hi])
            assert util.rms(np.concatenate(pieces, axis=1) - case[f"stream_{block}_{pad}"]) < 2e-6


def test_mas_known_answers():
    d = np.load(util.GOLDEN + "/mas_kat.npz")
    for i in range(int(d["n"])):
        p = vo.maximum_path_numpy(d[f"neg{i}"], d[f"ty{i}"], d[f"tx{i}"])
        assert np.array_equal(p.astype(np.int8), d[f"path{i}"]), f"case {i}"
        # exactly one 1 per valid frame row, monotone non-decreasing column
        for b in range(p.shape[0]):
            ty = int(d[f"ty{i}"][b])
            assert (p[b, :ty].sum(1) == 1).all() and p[b, ty:].sum() == 0
            cols = p[b, :ty].argmax(1)
            assert (np.diff(cols) >= 0).all() and (np.diff(cols) <= 1).all()


def test_generate_path_known_answers():
    d = np.load(util.GOLDEN + "/generate_path_kat.npz")
    dur = torch.from_numpy(d["durations"])
    ylen = torch.from_numpy(d["y_lengths"])
    y_mask = vo.sequence_mask(ylen, None).unsqueeze(1).float()
    x_mask = torch.ones(dur.shape[0], 1, dur.shape[2])
    attn, f2p = vo.generate_path(dur, x_mask, y_mask)
    assert np.array_equal(attn.numpy(), d["attn"])
    # SURVEY §8 a11 example: durations [2,0,3,1] -> rows 110000 / 000000 / 001110 / 000001
    assert attn[0, 0, :, 0].tolist() == [1, 1, 0, 0, 0, 0]
    assert attn[0, 0, :, 2].tolist() == [0, 0, 1, 1, 1, 0]
    assert f2p[1, 0].item() == -1  # all-zero durations: y_length clamps to 1, no phoneme


def test_audio_to_int16_matches_cli_formula():
    rng = np.random.default_rng(0)
    a = (rng.standard_normal(1000) * 0.1).astype(np.float32)
    ref = a.copy()
    ref *= 32767 / max(0.01, np.max(np.abs(ref))) * 0.6  # inference.py:101
    ref = np.clip(ref, -32767.0, 32767.0).astype(np.int16)
    got = vo.audio_to_int16(a)
    assert np.abs(got.astype(np.int32) - ref.astype(np.int32)).max() <= 1


def _c_mas():
    import ctypes as C
    import os
    import subprocess
    odir = os.path.join(os.path.dirname(util.GOLDEN), "..", "oracle")
    subprocess.check_call(["make", "-s", "-C", odir])
    lib = C.CDLL(os.path.join(odir, "_build", "libmas_oracle.so"))
    lib.mas_oracle.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_int]
    lib.mas_oracle.restype = None

    def run(neg, t_ys, t_xs):
        values = np.array(neg, dtype=np.float32, copy=True)
        paths = np.zeros(values.shape, dtype=np.int32)
        t_ys = np.ascontiguousarray(t_ys, dtype=np.int32)
        t_xs = np.ascontiguousarray(t_xs, dtype=np.int32)
        b, ty, tx = values.shape
        lib.mas_oracle(paths.ctypes.data, values.ctypes.data, t_ys.ctypes.data, t_xs.ctypes.data,
                       b, ty, tx)
        return paths
    return run


def test_mas_c_oracle_known_answers_and_numpy_twin():
    run = _c_mas()
    d = np.load(util.GOLDEN + "/mas_kat.npz")
    for i in range(int(d["n"])):
        assert np.array_equal(run(d[f"neg{i}"], d[f"ty{i}"], d[f"tx{i}"]).astype(np.int8),
                              d[f"path{i}"]), f"case {i}"
    rng = np.random.default_rng(1)
    neg = rng.standard_normal((3, 60, 17)).astype(np.float32)
    t_y = np.array([60, 41, 33], np.int32)
    t_x = np.array([17, 9, 17], np.int32)
    assert np.array_equal(run(neg, t_y, t_x), vo.maximum_path_numpy(neg, t_y, t_x))


def test_philox_oracle_matches_random123_known_answers():
    """The counter RNG behind the library's noise kernel: the three Philox4x32-10 known-answer
    vectors published with Random123 (kat_vectors), through the numpy restatement the GPU test
    compares the kernel with."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, want in kat:
        got = vo.philox4x32_10(np.array([c], dtype=np.uint32), np.array([k], dtype=np.uint32))[0]
        assert [int(v) for v in got] == want
    x = vo.philox_randn(200000, 9, 3)
    assert abs(float(x.mean())) < 1e-2 and abs(float(x.std()) - 1) < 1e-2
    assert np.array_equal(vo.philox_randn(64, 9, 5), vo.philox_randn(72, 9, 3)[8:])


# ------------------------------------------------------------------------------------------------
# ONNX operator-specification known answers for the uint8 dynamic-quantisation variant
# (wetts/vits/export_onnx.py:149-157 -> onnxruntime quantize_dynamic; ORT itself is a third-party
# dependency absent from the reference tree, v1.13.1 per runtime/core/cmake/onnxruntime.cmake:3-11).
# The inputs below are the literal examples of the published operator definitions
# (onnx/docs/Operators.md, DynamicQuantizeLinear-11 and ConvInteger-10); ConvInteger's expected outputs are
# literal in the specification, DynamicQuantizeLinear's are the specification's own numpy reference
# formula evaluated on its inputs (= the contents of the backend test data test_dynamicquantizelinear*).
# ------------------------------------------------------------------------------------------------
ONNX_DQL_KATS = [  # (name, X, shape, Y, Y_Scale, Y_ZeroPoint)
    ("dynamicquantizelinear", [0, 2, -3, -2.5, 1.34, 0.5], (6,),
     [153, 255, 0, 26, 221, 179], 0.019607844, 153),
    ("dynamicquantizelinear_max_adjusted", [-1.0, -2.1, -1.3, -2.5, -3.34, -4.0], (6,),
     [191, 121, 172, 96, 42, 0], 0.015686275, 255),
    ("dynamicquantizelinear_min_adjusted", [1, 2.1, 1.3, 2.5, 3.34, 4.0, 1.5, 2.6, 3.9, 4.0, 3.0, 2.345], (3, 4),
     [64, 134, 83, 159, 213, 255, 96, 166, 249, 255, 191, 149], 0.015686275, 0),
]
# ConvInteger-10 example: x uint8 3x3 = 2..10 with x_zero_point 1, w uint8 2x2 of ones (w_zero_point 0)
ONNX_CONVINT_X = np.arange(2, 11, dtype=np.uint8).reshape(3, 3)
ONNX_CONVINT_XZP = 1
ONNX_CONVINT_Y = np.array([12, 16, 24, 28], np.int32).reshape(2, 2)                         # no padding
ONNX_CONVINT_Y_PADDED = np.array([1, 3, 5, 3, 5, 12, 16, 9, 11, 24, 28, 15, 7, 15, 17, 9],  # pads [1,1,1,1]
                                 np.int32).reshape(4, 4)


def onnx_spec_dql(X):
    """The specification's reference formula for DynamicQuantizeLinear, transcribed."""
    x_min = np.minimum(0, np.min(X))
    x_max = np.maximum(0, np.max(X))
    y_scale = np.float32((x_max - x_min) / (255 - 0))
    y_zp = np.clip(round((0 - x_min) / y_scale), 0, 255).astype(np.uint8)
    y = np.clip(np.round(X / y_scale) + y_zp, 0, 255).astype(np.uint8)
    return y, y_scale, y_zp


def onnx_convinteger_as_conv1d():
    """The ConvInteger example (2-D, 2x2 kernel) restated as the 1-D conv the decoder uses: output row i of the
    padded result convolves input rows (i-1, i) -- those two rows become the two input CHANNELS of batch item
    i, a padding row is a row of x_zero_point -- and the kernel row [1, 1] over columns (j-1, j) becomes a
    3-tap "same" kernel [1, 1, 0] over a row widened by one x_zero_point column.  Returns
    (xq uint8 [4,2,4], wq uint8 [1,2,3], expected int32 [4,1,4] = ONNX_CONVINT_Y_PADDED)."""
    zp = ONNX_CONVINT_XZP
    rows = np.full((5, 4), zp, np.uint8)  # rows -1 .. 3, columns 0 .. 3 (column 3 = padding)
    rows[1:4, 0:3] = ONNX_CONVINT_X
    xq = np.stack([np.stack([rows[i], rows[i + 1]]) for i in range(4)])
    wq = np.array([[[1, 1, 0], [1, 1, 0]]], np.uint8)
    return xq, wq, ONNX_CONVINT_Y_PADDED.reshape(4, 1, 4)


@pytest.mark.parametrize("name,X,shape,Y,scale,zp", ONNX_DQL_KATS)
def test_onnx_spec_dynamicquantizelinear_known_answers(name, X, shape, Y, scale, zp):
    X = np.array(X, np.float32).reshape(shape)
    y_ref, s_ref, z_ref = onnx_spec_dql(X)  # the transcription reproduces the literals ...
    assert y_ref.reshape(-1).tolist() == Y and float(s_ref) == float(np.float32(scale)) and int(z_ref) == zp
    y, s, z = vo.dynamic_quantize_linear(torch.from_numpy(X))  # ... and the oracle reproduces both
    assert y.dtype == torch.uint8 and y.numpy().reshape(-1).tolist() == Y
    assert float(s) == float(np.float32(scale)) and int(z) == zp


def test_onnx_spec_convinteger_known_answers():
    # the published literals, by the definition itself (direct 2-D sums) ...
    xs = ONNX_CONVINT_X.astype(np.int32) - ONNX_CONVINT_XZP
    direct = np.array([[xs[i:i + 2, j:j + 2].sum() for j in range(2)] for i in range(2)], np.int32)
    assert np.array_equal(direct, ONNX_CONVINT_Y)
    assert np.array_equal(ONNX_CONVINT_Y_PADDED[1:3, 1:3], ONNX_CONVINT_Y)
    # ... and through the oracle's ConvInteger in the 1-D form the decoder uses
    xq, wq, want = onnx_convinteger_as_conv1d()
    got = vo.conv_integer(torch.from_numpy(xq), torch.from_numpy(wq), ONNX_CONVINT_XZP, 0, padding=1)
    assert got.dtype == torch.int32 and np.array_equal(got.numpy(), want)


def test_dynamic_quant_conv1d_composes_the_spec_operators():
    """The float-in / float-out conv node = DQL -> ConvInteger -> Cast * (s_x s_w) + bias, on the spec inputs:
    a 1x1 conv whose weight tensor is [[1], [255]] quantises to w_q = [1, 255], s_w = 1, z_w = 0, so output
    channel 0 is exactly (Y - Y_ZeroPoint) * Y_Scale of the DynamicQuantizeLinear example."""
    for name, X, shape, Y, scale, zp in ONNX_DQL_KATS:
        x = torch.tensor(X, dtype=torch.float32).view(1, 1, -1)
        w = torch.tensor([[[1.0]], [[255.0]]])
        out = vo.dynamic_quant_conv1d(x, w, None)
        want = (np.array(Y, np.float32) - np.float32(zp)) * np.float32(scale)
        assert np.array_equal(out[0, 0].numpy(), want), name
