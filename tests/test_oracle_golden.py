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
