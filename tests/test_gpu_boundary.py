"""GPU tier: the drop-in boundary surfaces (ORT-shaped sessions, streaming chunk protocol, CLI,
native wetts_infer) on top of the same HIP kernels."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu


def _model(name):
    from wetts_amd import SynthesizerTrn, config
    case = util.load_case(name)
    cfg, sd, W, blob = util.case_model(case)
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]),
                         **util.model_dict(case))
    net.load_state_dict(sd).to("cuda")
    return net, case, cfg, sd, W


def test_session_run_matches_export_forward_shapes_and_noise_free_golden():
    """noise_scale = noise_scale_w = 0 makes infer() deterministic, so the ORT-shaped call can be
    compared with the reference golden audio without injecting noise."""
    from wetts_amd.session import InferenceSession
    net, case, *_ = _model("tiny_sdp_nonoise")
    sess = InferenceSession(net)
    B = case["x"].shape[0]
    scales = np.tile(np.array([[0.0, 1.0, 0.0]], np.float32), (B, 1))
    out = sess.run(None, {"input": case["x"], "input_lengths": case["x_lengths"],
                          "scales": scales, "sid": case["sid"]})
    assert isinstance(out, list) and out[0].dtype == np.float32
    assert out[0].shape == case["audio"].shape
    assert util.rms(out[0] - case["audio"]) < 1e-4
    assert [i.name for i in sess.get_inputs()] == ["input", "input_lengths", "scales", "sid"]
    with pytest.raises(ValueError):
        sess.run(["nope"], {})


def test_streaming_encoder_decoder_chunks_equal_full_decode_away_from_edges():
    """Chunked decoding with overlap-discard (inference_onnx.py:37-76): identical sample count and
    equal to the one-shot decode wherever the receptive field fits inside the padding."""
    from wetts_amd.session import DecoderSession, EncoderSession, get_chunks, stream_decode
    net, case, *_ = _model("tiny_sdp_nonoise")
    enc, dec = EncoderSession(net), DecoderSession(net)
    scales = np.array([[0.0, 1.0, 0.0]], np.float32)
    feeds = {"input": case["x"][:1], "input_lengths": case["x_lengths"][:1], "scales": scales,
             "sid": case["sid"][:1]}
    z = enc.run(None, feeds)[0]
    assert z.ndim == 3 and z.shape[2] == net.inter_channels
    L = z.shape[1]
    full = dec.run(None, {"z": z, "sid": case["sid"][:1]})[0][0, 0]
    assert full.shape[0] == L * net.hop_length
    pieces = list(stream_decode(dec, z, case["sid"][:1], chunk_size=16, pad_size=12))
    cat = np.concatenate(pieces)
    assert cat.shape == full.shape
    wins = get_chunks(L, 16, 12)
    assert wins[0][0] == 0 and wins[-1][1] == L
    # tiny config: receptive field of the generator is < 12 frames => interiors agree closely
    err = np.abs(cat - full)
    assert np.median(err) < 1e-5 and err.max() < 0.05


def test_native_wetts_infer_matches_python_composition():
    """The C entry point composing all stages (shape of VitsModel::Forward, vits_model.cc:89-93)
    gives the same waveform as the Python-orchestrated stage calls."""
    from wetts_amd import _lib
    net, case, *_ = _model("tiny_sdp_b3")
    lib = _lib.load()
    dev = net.device
    x = util.t(case["x"]).to(dev)
    xl = util.t(case["x_lengths"]).to(dev)
    sid = util.t(case["sid"]).to(dev)
    ns, ls, nsw = [float(v) for v in case["scales"]]
    B, Tx = x.shape
    Ty = case["eps_z"].shape[2]
    cap = Ty + 5
    eps_w = util.t(case["eps_w"]).to(dev).contiguous()
    eps_z = torch.zeros(B, net.inter_channels, cap, device=dev)
    eps_z[:, :, :Ty] = util.t(case["eps_z"]).to(dev)
    nws = lib.wetts_infer_workspace_bytes(net._handle, B, Tx, cap)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    audio = torch.zeros(B * cap * net.hop_length, dtype=torch.float32, device=dev)
    ylen = np.zeros(B, np.int64)
    frames = C.c_int32()
    rc = lib.wetts_infer(net._handle, _lib.ptr(x), _lib.ptr(xl), _lib.ptr(sid), _lib.ptr(eps_w),
                         _lib.ptr(eps_z), ns, ls, nsw, B, Tx, cap, _lib.ptr(audio),
                         ylen.ctypes.data_as(C.c_void_p), C.byref(frames), _lib.ptr(ws), nws,
                         _lib.current_stream_ptr())
    _lib.check(rc, "wetts_infer")
    torch.cuda.synchronize()
    assert frames.value == Ty
    got = audio[:B * Ty * net.hop_length].view(B, 1, -1).cpu().numpy()
    assert util.rms(got - case["audio"]) < 1e-4
    assert ylen.tolist() == case["y_mask"].sum(axis=(1, 2)).astype(np.int64).tolist()
    # capacity too small is reported, not overrun
    rc = lib.wetts_infer(net._handle, _lib.ptr(x), _lib.ptr(xl), _lib.ptr(sid), _lib.ptr(eps_w),
                         _lib.ptr(eps_z), ns, ls, nsw, B, Tx, 3, _lib.ptr(audio),
                         ylen.ctypes.data_as(C.c_void_p), C.byref(frames), _lib.ptr(ws), nws,
                         _lib.current_stream_ptr())
    assert rc == -3 and frames.value == Ty


def test_cli_writes_reference_format_wavs(tmp_path, capsys):
    from scipy.io import wavfile
    from wetts_amd import config, inference, synth
    mname, n_vocab, n_spk = "tiny", 12, 2
    cfg = config.make_config(config.MODEL_CONFIGS[mname], n_vocab, n_spk)
    sd = synth.make_state_dict(cfg, 3)
    ckpt = tmp_path / "G_100.pth"
    torch.save({"model": sd, "iteration": 100, "optimizer": {}, "learning_rate": 2e-4}, ckpt)
    import json
    (tmp_path / "cfg.json").write_text(json.dumps({
        "train": {"segment_size": 8192},
        "data": {"filter_length": 1024, "hop_length": 8, "sampling_rate": 22050},
        "model": config.MODEL_CONFIGS[mname]}))
    (tmp_path / "phones.txt").write_text("".join(f"p{i} {i}\n" for i in range(n_vocab)))
    (tmp_path / "speaker.txt").write_text("spk0 0\nspk1 1\n")
    (tmp_path / "test.txt").write_text("a/utt1.wav|spk0|p1 p2 p3 p4 p5\nb/utt2.wav|spk1|p6 p7 p8\n")
    out = tmp_path / "out"
    out.mkdir()
    inference.main(["--checkpoint", str(ckpt), "--cfg", str(tmp_path / "cfg.json"), "--outdir",
                    str(out), "--phone_table", str(tmp_path / "phones.txt"), "--speaker_table",
                    str(tmp_path / "speaker.txt"), "--test_file", str(tmp_path / "test.txt"),
                    "--gpu", "0", "--batch", "2", "--seed", "1"])
    printed = [l for l in capsys.readouterr().out.splitlines() if l.endswith(".wav")]
    assert printed == ["a/utt1.wav", "b/utt2.wav"]  # test-file order, whatever the bucket order was
    ragged = {n: wavfile.read(out / n)[1] for n in ("utt1.wav", "utt2.wav")}
    # --decode padded: the shorter utterance is decoded inside the padded batch (infer()'s semantics); same seed,
    # same frame counts, the longest utterance's audio is the same either way
    inference.main(["--checkpoint", str(ckpt), "--cfg", str(tmp_path / "cfg.json"), "--outdir",
                    str(out), "--phone_table", str(tmp_path / "phones.txt"), "--speaker_table",
                    str(tmp_path / "speaker.txt"), "--test_file", str(tmp_path / "test.txt"),
                    "--gpu", "0", "--batch", "2", "--seed", "1", "--decode", "padded", "--max_pad_frac", "0.9"])
    padded = {n: wavfile.read(out / n)[1] for n in ("utt1.wav", "utt2.wav")}
    assert all(padded[n].shape == ragged[n].shape for n in ragged)
    assert np.abs(padded["utt1.wav"].astype(np.int32) - ragged["utt1.wav"].astype(np.int32)).max() <= 1
    for name in ("utt1.wav", "utt2.wav"):
        sr, pcm = wavfile.read(out / name)
        assert sr == 22050 and pcm.dtype == np.int16 and pcm.size > 0
        # inference.py:101: peak-normalised to 0.6 full scale
        assert abs(int(np.abs(pcm).max()) - int(32767 * 0.6)) <= 2
    with pytest.raises(KeyError):  # unknown phone, like inference.py:85
        (tmp_path / "bad.txt").write_text("c/utt3.wav|spk0|p1 zz\n")
        inference.main(["--checkpoint", str(ckpt), "--cfg", str(tmp_path / "cfg.json"),
                        "--outdir", str(out), "--phone_table", str(tmp_path / "phones.txt"),
                        "--speaker_table", str(tmp_path / "speaker.txt"), "--test_file",
                        str(tmp_path / "bad.txt"), "--gpu", "0"])


def test_ragged_and_degenerate_batches():
    """Ragged lengths incl. a length-1 utterance; padded rows must not disturb valid ones
    (compared with the oracle run on the same padded batch, SURVEY §7 hard-part 3)."""
    from oracle import vits_oracle as vo
    net, case, cfg, sd, W = _model("tiny_sdp_b3")
    cd = util.cfg_dict(cfg)
    g = torch.Generator().manual_seed(9)
    B, Tx = 4, 11
    x = torch.randint(0, int(case["n_vocab"]), (B, Tx), generator=g)
    xl = torch.tensor([11, 1, 6, 3])
    sid = torch.tensor([0, 1, 2, 1])
    eps_w = torch.randn(B, 2, Tx, generator=g)
    st = vo.infer(W, cd, x, xl, sid, 0.667, 1.0, 0.8, eps_w=eps_w,
                  eps_z=None, return_stages=True) if False else None
    # draw eps_z after Ty is known from a first oracle pass with zeros
    torch.manual_seed(0)
    st0 = vo.infer(W, cd, x, xl, sid, 0.667, 1.0, 0.8, eps_w=eps_w,
                   eps_z=torch.zeros(B, cd["inter_channels"], 1).expand(B, -1, 1)
                   if False else None, return_stages=True)
    Ty = st0["y_mask"].shape[-1]
    eps_z = torch.randn(B, cd["inter_channels"], Ty, generator=g)
    ref = vo.infer(W, cd, x, xl, sid, 0.667, 1.0, 0.8, eps_w=eps_w, eps_z=eps_z,
                   return_stages=True)
    o, attn, y_mask, _ = net.infer(x.cuda(), xl.cuda(), sid=sid.cuda(), noise_scale=0.667,
                                   length_scale=1.0, noise_scale_w=0.8, eps_w=eps_w.cuda(),
                                   eps_z=eps_z.cuda())
    assert np.array_equal(y_mask.cpu().numpy(), ref["y_mask"].numpy())
    assert np.array_equal(attn.cpu().numpy(), ref["attn"].numpy())
    assert util.rms(o.cpu().numpy() - ref["o"].numpy()) < 1e-4


@pytest.mark.parametrize("cname", ["tiny_sdp_b3", "tiny_dp_b2", "tiny_vocos_b2", "tiny_vits2_vocos_b2",
                                   "tiny_mono_post_b2", "tiny_mono_inter_b3"])
def test_random_small_shapes_match_the_oracle(cname):
    """Odd batch sizes and text lengths (B = 1..5, Tx = 1..23, ragged, length-1 utterances) against the
    oracle on the same padded batch: these are the shapes the small-launch conv schedule, the scalar
    attention path and every partial tile see; alignment must be EQUAL, audio within 1e-4 RMS."""
    from oracle import vits_oracle as vo
    net, case, cfg, sd, W = _model(cname)
    cd = util.cfg_dict(cfg)
    g = torch.Generator().manual_seed(1234)
    n_vocab, n_spk = int(case["n_vocab"]), int(case["n_speakers"])
    worst = 0.0
    for B, Tx in [(1, 1), (1, 2), (1, 7), (2, 3), (3, 23), (5, 9), (4, 16), (2, 17)]:
        x = torch.randint(0, n_vocab, (B, Tx), generator=g)
        xl = torch.randint(1, Tx + 1, (B,), generator=g)
        xl[int(torch.randint(0, B, (1,), generator=g))] = Tx  # the batch's longest fills the padded width
        sid = torch.randint(0, max(1, n_spk), (B,), generator=g)
        eps_w = torch.randn(B, 2, Tx, generator=g)
        st0 = vo.infer(W, cd, x, xl, sid, 0.667, 1.0, 0.8, eps_w=eps_w, return_stages=True)
        Ty = st0["y_mask"].shape[-1]  # (durations do not depend on eps_z)
        if "vocos" in cname and Ty < 2:
            continue  # reflection pad needs two frames (the reference raises, tested separately)
        eps_z = torch.randn(B, cd["inter_channels"], Ty, generator=g)
        ref = vo.infer(W, cd, x, xl, sid, 0.667, 1.0, 0.8, eps_w=eps_w, eps_z=eps_z, return_stages=True)
        o, attn, y_mask, _ = net.infer(x.cuda(), xl.cuda(), sid=sid.cuda(), noise_scale=0.667,
                                       length_scale=1.0, noise_scale_w=0.8, eps_w=eps_w.cuda(),
                                       eps_z=eps_z.cuda())
        assert np.array_equal(y_mask.cpu().numpy(), ref["y_mask"].numpy()), (B, Tx)
        assert np.array_equal(attn.cpu().numpy(), ref["attn"].numpy()), (B, Tx)
        err = util.rms(o.cpu().numpy() - ref["o"].numpy())
        worst = max(worst, err)
        assert err < 1e-4, (B, Tx, err)
    print(cname, "worst audio rms error over the shape sweep", worst)


def test_graphed_stream_decoder_equals_plain():
    """HIP-graph replay of the decoder windows (session.GraphedDecoder) is the same kernels in the
    same order: every streamed piece must EQUAL the un-graphed one, and repeated replays of one
    window shape with different inputs must not leak state."""
    from wetts_amd.session import DecoderSession, stream_decode
    net, case, cfg, *_ = _model("tiny_sdp_b3")
    torch.manual_seed(3)
    z = torch.randn(1, 131, cfg.inter_channels).numpy()
    sid = np.array([min(1, int(case["n_speakers"]) - 1)], dtype=np.int64)
    plain, graphed = DecoderSession(net), DecoderSession(net, use_graph=True)
    for rep in range(2):
        zz = z * (1.0 + rep)
        a = np.concatenate(list(stream_decode(plain, zz, sid, chunk_size=40, pad_size=10)))
        b = np.concatenate(list(stream_decode(graphed, zz, sid, chunk_size=40, pad_size=10)))
        assert a.shape == b.shape == (131 * net.hop_length,)
        assert np.array_equal(a, b)


def test_vocos_needs_two_frames_like_reflection_pad():
    """nn.ReflectionPad1d([1, 0]) (decoders.py:265) raises for a single frame; the HIP path reports
    the same condition instead of reading out of bounds.  Two frames work."""
    from wetts_amd import _lib
    net, case, cfg, *_ = _model("tiny_vocos_b2")
    z = torch.randn(1, cfg.inter_channels, 1, device="cuda")
    with pytest.raises(_lib.WettsError, match="at least 2 frames"):
        net.hifigan(z, None if net.n_speakers == 0 else torch.zeros(1, net.gin_channels, device="cuda"))
    z2 = torch.randn(1, cfg.inter_channels, 2, device="cuda")
    out = net.hifigan(z2, None if net.n_speakers == 0 else torch.zeros(1, net.gin_channels, device="cuda"))
    assert out.shape == (1, 1, 2 * net.hop_length) and torch.isfinite(out).all()


def test_status_word_domain_and_id_range_errors():
    """Errors the reference raises from inside its modules come back through the device status
    word with the ONE host sync of infer(): a NaN in the spline parameters fails
    `assert (discriminant >= 0).all()` (transforms.py:171) => AssertionError in the Python twin,
    WETTS_E_DOMAIN (-4) from wetts_infer; an id outside the embedding table is nn.Embedding's
    IndexError => IndexError / WETTS_E_INVALID (-1)."""
    from wetts_amd import SynthesizerTrn, _lib, config
    case = util.load_case("tiny_sdp_b3")
    cfg, sd, W, _ = util.case_model(case)
    lib = _lib.load()

    def native(net, x, xl, sid):
        B, Tx = x.shape
        cap = 200
        nws = lib.wetts_infer_workspace_bytes(net._handle, B, Tx, cap)
        ws = torch.empty(nws, dtype=torch.uint8, device="cuda")
        audio = torch.zeros(B * cap * net.hop_length, dtype=torch.float32, device="cuda")
        ylen = np.zeros(B, np.int64)
        frames = C.c_int32()
        # eps_w / eps_z = NULL: drawn by the library's Philox stream
        rc = lib.wetts_infer(net._handle, _lib.ptr(x), _lib.ptr(xl), _lib.ptr(sid), None, None,
                             0.667, 1.0, 0.8, B, Tx, cap, _lib.ptr(audio),
                             ylen.ctypes.data_as(C.c_void_p), C.byref(frames), _lib.ptr(ws), nws,
                             _lib.current_stream_ptr())
        torch.cuda.synchronize()
        return rc, frames.value, ylen

    x = util.t(case["x"]).cuda()
    xl = util.t(case["x_lengths"]).cuda()
    sid = util.t(case["sid"]).cuda()
    mk = lambda state: SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]),
                                      **config.MODEL_CONFIGS[str(case["model"])]).load_state_dict(state).to("cuda")
    # healthy model: native call with internal noise succeeds, deterministic under the seed
    net = mk(sd)
    lib.wetts_set_seed(net._handle, 77)
    rc, fr, yl = native(net, x, xl, sid)
    assert rc == 0 and fr == int(yl.max()) and (yl >= 1).all()
    lib.wetts_set_seed(net._handle, 77)
    rc2, fr2, yl2 = native(net, x, xl, sid)
    assert rc2 == 0 and fr2 == fr and yl2.tolist() == yl.tolist()
    # phoneme id out of range
    xb = x.clone()
    xb[0, 1] = int(case["n_vocab"])
    with pytest.raises(IndexError):
        net.infer(xb, xl, sid=sid)
    rc, _, _ = native(net, xb, xl, sid)
    assert rc == -1 and "phoneme" in _lib.last_error()
    # speaker id out of range
    sb = sid.clone()
    sb[2] = int(case["n_speakers"]) + 3
    with pytest.raises(IndexError):
        net.infer(x, xl, sid=sb)
    rc, _, _ = native(net, x, xl, sb)
    assert rc == -1 and "speaker" in _lib.last_error()
    # the model is still usable afterwards
    o, *_ = net.infer(x, xl, sid=sid)
    assert torch.isfinite(o).all()
    # spline domain: NaN spline parameters (the last ConvFlow's projection bias)
    bad = dict(sd)
    key = [k for k in bad if k.startswith("dp.flows.") and k.endswith(".proj.bias")][-1]
    bad[key] = torch.full_like(bad[key], float("nan"))
    netb = mk(bad)
    with pytest.raises(AssertionError):
        netb.infer(x, xl, sid=sid)
    rc, _, _ = native(netb, x, xl, sid)
    assert rc == -4 and "discriminant" in _lib.last_error()


def test_randn_kernel_is_philox_box_muller():
    """wetts_randn against the CPU restatement (oracle.philox_randn, itself pinned to the Random123
    known-answer vectors in the CPU tier): same uint32 stream, float32 Box-Muller within libm
    round-off; offsets address the stream; moments of a large draw are standard normal."""
    from oracle import vits_oracle as vo
    from wetts_amd import _lib
    lib = _lib.load()
    n = 100003
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    _lib.check(lib.wetts_randn(_lib.ptr(out), n, 0x123456789ABCDEF, 5, None), "randn")
    torch.cuda.synchronize()
    ref = vo.philox_randn(n, 0x123456789ABCDEF, 5)
    got = out.cpu().numpy()
    assert np.abs(got - ref).max() < 2e-5
    # offset semantics: a draw at offset 5 + 10 equals the tail of the draw at offset 5
    out2 = torch.empty(1000, dtype=torch.float32, device="cuda")
    _lib.check(lib.wetts_randn(_lib.ptr(out2), 1000, 0x123456789ABCDEF, 15, None), "randn")
    assert torch.equal(out2.cpu(), out[40:1040].cpu())
    big = torch.empty(1 << 22, dtype=torch.float32, device="cuda")
    _lib.check(lib.wetts_randn(_lib.ptr(big), big.numel(), 42, 0, None), "randn")
    b = big.double()
    m, v = float(b.mean()), float(b.var())
    kurt = float(((b - m) ** 4).mean() / v ** 2)
    assert abs(m) < 3e-3 and abs(v - 1) < 5e-3 and abs(kurt - 3) < 3e-2 and torch.isfinite(big).all()


def test_manual_seed_rewinds_the_noise_stream():
    """torch.manual_seed(s) makes a run reproducible the way it does for the reference -- also when `s` is the
    seed already in use (`manual_seed(0); a = infer(); manual_seed(0); b = infer()` gives a == b), while two calls
    without a re-seed draw different noise.  The Philox (seed, offset) pair is the device generator's own."""
    net, case, *_ = _model("tiny_sdp_b3")
    x, xl, sid = (util.t(case[k]).cuda() for k in ("x", "x_lengths", "sid"))

    def run():
        o, _, ym, (z, z_p, _, _) = net.infer(x, xl, sid=sid, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8)
        return o.cpu(), z_p.cpu()
    torch.manual_seed(0)
    a = run()
    b = run()
    torch.manual_seed(0)
    c = run()
    d = run()
    torch.manual_seed(1)
    e = run()
    assert a[0].shape == c[0].shape and torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    assert b[0].shape == d[0].shape and torch.equal(b[0], d[0])  # the continuation is reproducible too
    assert a[1].shape != b[1].shape or not torch.equal(a[1], b[1])
    assert a[1].shape != e[1].shape or not torch.equal(a[1], e[1])
    # an ATen draw from the same generator advances the stream like any other consumer
    torch.manual_seed(0)
    torch.randn(8, device="cuda")
    f = run()
    assert a[1].shape != f[1].shape or not torch.equal(a[1], f[1])


def test_generate_path_known_answers_through_the_c_abi():
    """tests/golden/generate_path_kat.npz -- the reference's commons.generate_path on its edge cases (zero
    durations, an all-zero row whose y_length clamps to 1, a single long phoneme; make_golden.py) -- through
    wetts_durations_to_lengths + wetts_length_regulate: y_lengths, y_mask and attn EQUAL, frame2phone = -1 where no
    phoneme owns the frame (commons.py:120-136, models.py:254-259)."""
    from wetts_amd import _lib
    net, case, cfg, *_ = _model("tiny_sdp_b3")
    lib = _lib.load()
    d = np.load(os.path.join(util.GOLDEN, "generate_path_kat.npz"))
    dur = d["durations"][:, 0]  # [B, Tx] integer-valued w_ceil
    B, Tx = dur.shape
    dev = net.device
    # logw such that ceil(exp(logw) * 1 * 1) == duration: log(d) for d > 0, -inf -> exp = 0 for d = 0
    with np.errstate(divide="ignore"):
        logw = torch.from_numpy(np.log(dur).astype(np.float32)).to(dev)
    x_mask = torch.ones(B, Tx, device=dev)
    w_ceil = torch.empty(B, Tx, device=dev)
    cum = torch.empty(B, Tx, device=dev)
    meta = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    status = C.c_void_p(meta.data_ptr() + 8 * B)
    s = _lib.current_stream_ptr()
    _lib.check(lib.wetts_durations_to_lengths(_lib.ptr(logw), _lib.ptr(x_mask), 1.0, B, Tx, _lib.ptr(w_ceil),
                                              _lib.ptr(cum), _lib.ptr(meta), status, s), "durations_to_lengths")
    mh = meta.cpu().numpy()
    assert int(mh[B]) & 0xFFFFFFFF == 0
    assert np.array_equal(w_ceil.cpu().numpy(), dur)
    assert mh[:B].tolist() == d["y_lengths"].tolist() == [6, 1, 4, 5]
    Ty = int(mh[:B].max())
    I = cfg.inter_channels
    stats = torch.randn(B, 2 * I, Tx, device=dev)
    eps = torch.zeros(B, I, Ty, device=dev)
    f2p = torch.empty(B, Ty, dtype=torch.int32, device=dev)
    y_mask = torch.empty(B, Ty, device=dev)
    attn = torch.empty(B, Ty, Tx, device=dev)
    m_p = torch.empty(B, I, Ty, device=dev)
    logs_p = torch.empty(B, I, Ty, device=dev)
    z_p = torch.empty(B, I, Ty, device=dev)
    _lib.check(lib.wetts_length_regulate(net._handle, _lib.ptr(stats), _lib.ptr(cum), _lib.ptr(x_mask),
                                         _lib.ptr(meta), _lib.ptr(eps), I * Ty, Ty, 0.0, B, Tx, Ty, _lib.ptr(f2p),
                                         _lib.ptr(y_mask), _lib.ptr(attn), _lib.ptr(m_p), _lib.ptr(logs_p),
                                         _lib.ptr(z_p), s), "length_regulate")
    ref_attn = d["attn"]  # [B,1,Ty,Tx]
    assert np.array_equal(attn.cpu().numpy(), ref_attn[:, 0])
    ref_mask = (np.arange(Ty)[None] < d["y_lengths"][:, None]).astype(np.float32)
    assert np.array_equal(y_mask.cpu().numpy(), ref_mask)
    f = f2p.cpu().numpy()
    own = ref_attn[:, 0].argmax(-1)
    has = ref_attn[:, 0].sum(-1) > 0
    assert np.array_equal(f[has], own[has]) and (f[~has] == -1).all()
    assert f[1, 0] == -1  # all-zero durations: y_length clamps to 1 and no phoneme owns that frame
    # the prior expansion is the gather attn . m_p (models.py:262-265): frames without a phoneme get zeros
    ref_m = torch.matmul(torch.from_numpy(ref_attn[:, 0]).to(dev), stats[:, :I].transpose(1, 2)).transpose(1, 2)
    assert torch.equal(m_p, ref_m) and torch.equal(z_p, ref_m)  # noise_scale 0 => z_p = m_p


def test_state_dict_reads_back_the_device_weights():
    """state_dict() on a device model goes through wetts_get_blob: same tensors as the folded checkpoint, on the
    module's device, and a second model built from it synthesises the same waveform."""
    from wetts_amd import SynthesizerTrn, checkpoint, config
    net, case, cfg, sd, W = _model("tiny_mono_post_b2")
    out = net.state_dict()
    assert all(v.device.type == "cuda" for v in out.values())
    for name, *_ in checkpoint.blob_layout(cfg):
        assert torch.equal(out[name].cpu(), W[name].to(torch.float32)), name
    net2 = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]),
                          **config.MODEL_CONFIGS[str(case["model"])])
    net2.load_state_dict({k: v.cpu() for k, v in out.items()}).to("cuda")
    args = (util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda())
    kw = dict(sid=util.t(case["sid"]).cuda(), noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8,
              eps_w=util.t(case["eps_w"]).cuda(), eps_z=util.t(case["eps_z"]).cuda())
    assert torch.equal(net.infer(*args, **kw)[0], net2.infer(*args, **kw)[0])


def test_triton_min_chunk_streaming_protocol():
    """stream_decode(min_chunk=65, chunk 70, pad 10) = the Triton twin's protocol (stream_tts/1/model.py:58-111):
    L*hop samples; the reflect-padded last window contributes only audio of real frames; interior samples equal the
    one-shot decode (pad 10 < receptive field, so only away from the window edges)."""
    from wetts_amd.session import (DecoderSession, TRITON_BLOCK_SIZE, TRITON_MIN_CHUNK, TRITON_PAD_SIZE,
                                   get_chunks_min, stream_decode)
    net, case, cfg, *_ = _model("tiny_sdp_b3")
    dec = DecoderSession(net)
    sid = np.array([1], dtype=np.int64)
    hop = net.hop_length
    torch.manual_seed(5)
    for L in (30, 100, 150, 211):
        z = torch.randn(1, L, cfg.inter_channels).numpy()
        full = dec.run(None, {"z": z, "sid": sid})[0][0, 0]
        pieces = list(stream_decode(dec, z, sid, TRITON_BLOCK_SIZE, TRITON_PAD_SIZE, min_chunk=TRITON_MIN_CHUNK))
        wins, pad_end = get_chunks_min(L, TRITON_BLOCK_SIZE, TRITON_PAD_SIZE, TRITON_MIN_CHUNK)
        assert len(pieces) == len(wins)
        cat = np.concatenate(pieces)
        assert cat.shape == full.shape == (L * hop,)
        err = np.abs(cat - full)
        rf = 19  # tests/test_gpu_native.py: receptive field of the tiny generator in frames
        interior = np.zeros(L, bool)
        for i, (ws, we) in enumerate(wins):
            a, b = i * TRITON_BLOCK_SIZE, min((i + 1) * TRITON_BLOCK_SIZE, L)  # frames this window emits
            t = np.arange(a, b)
            # the last window's reflected tail replaces the utterance's real right edge: only frames rf away from it
            right_ok = (we - 1 - t >= rf) if (we < L or (pad_end and i == len(wins) - 1)) else np.ones_like(t, bool)
            interior[a:b] = ((ws == 0) | (t - ws >= rf)) & right_ok
        assert interior.any()
        assert err.reshape(L, hop)[interior].max() < 1e-5, (L, err.reshape(L, hop)[interior].max())


def test_graphed_encoder_matches_plain_and_keeps_the_noise_stream():
    """session.GraphedEncoder replays the encoder call from two captured HIP graphs (before / after the call's one
    host sync; the second per frame bucket, extra frames masked).  Same seed => same durations, same alignment (EQUAL)
    and z within round-off of the plain call (the flow's tile shapes follow the bucketed length); replays with other
    inputs of the same shape do not leak state; errors the plain path raises are raised."""
    from wetts_amd.session import EncoderSession
    net, case, cfg, sd, W = _model("tiny_sdp_b3")
    plain, graphed = EncoderSession(net), EncoderSession(net, use_graph=True, frame_bucket=16)
    scales = np.tile(np.array([[0.667, 1.0, 0.8]], np.float32), (case["x"].shape[0], 1))
    g = np.random.default_rng(0)
    for rep in range(4):
        x = case["x"] if rep == 0 else g.integers(0, int(case["n_vocab"]), size=case["x"].shape)
        feeds = {"input": x, "input_lengths": case["x_lengths"], "scales": scales, "sid": case["sid"]}
        torch.manual_seed(10 + rep)
        a = plain.run(None, feeds)[0]
        torch.manual_seed(10 + rep)
        b = graphed.run(None, feeds)[0]
        assert a.shape == b.shape, (a.shape, b.shape)  # same frame count: the durations saw the same noise
        assert util.rel_rms(b, a) < 1e-5, (rep, util.rel_rms(b, a))
        assert np.array_equal(b == 0, a == 0)  # the same frames are masked
    ge = graphed._graphed
    assert len(ge._pre) == 1 and 1 <= len(next(iter(ge._pre.values()))["post"]) <= 4
    bad = dict(feeds, input=np.full_like(case["x"], int(case["n_vocab"]) + 5))
    with pytest.raises(IndexError):
        graphed.run(None, bad)
    torch.manual_seed(3)
    ok = graphed.run(None, feeds)[0]  # and the entry still works afterwards
    assert np.isfinite(ok).all()


def test_graphed_encoder_caches_are_bounded_and_bucket_the_phoneme_count():
    """A serving loop sees a new phoneme count with almost every request.  The phoneme count is bucketed (ids padded with
    zeros, x_lengths masks them: a shorter utterance in a padded batch), so requests of 9 ... 16 phonemes replay ONE
    first-half graph; and both caches are LRU-bounded, so 40 distinct lengths leave at most `max_shapes` entries (each
    with at most `max_buckets` second-half graphs) instead of 40 graphs + 40 private workspaces.  Results stay those of
    the plain call (same seed => same durations, z to round-off), also for an entry that was evicted and re-captured."""
    from wetts_amd.session import EncoderSession
    net, case, cfg, sd, W = _model("tiny_sdp_b3")
    plain = EncoderSession(net)
    graphed = EncoderSession(net, use_graph=True, frame_bucket=16, phoneme_bucket=8, max_shapes=3, max_buckets=2)
    ge = graphed._graphed
    g = np.random.default_rng(1)

    def one(Tx, seed):
        x = g.integers(0, int(case["n_vocab"]), size=(2, Tx))
        feeds = {"input": x, "input_lengths": np.array([Tx, max(1, Tx - 3)]), "scales": np.tile(np.array([[0.667, 1.0, 0.8]], np.float32), (2, 1)),
                 "sid": np.array([0, 2])}
        torch.manual_seed(seed)
        a = plain.run(None, feeds)[0]
        torch.manual_seed(seed)
        b = graphed.run(None, feeds)[0]
        assert a.shape == b.shape, (Tx, a.shape, b.shape)
        assert util.rel_rms(b, a) < 1e-5 and np.array_equal(b == 0, a == 0), (Tx, util.rel_rms(b, a))

    for Tx in range(9, 17):  # one phoneme bucket (16): a single first-half graph
        one(Tx, 100 + Tx)
    assert len(ge._pre) == 1
    for Tx in range(1, 41):  # five buckets through a cache of three
        one(Tx, 200 + Tx)
    assert len(ge._pre) <= 3 and all(len(e["post"]) <= 2 for e in ge._pre.values())
    before = ge.captures
    one(40, 7)  # most recent bucket: a replay, nothing captured
    assert ge.captures <= before + 1  # (at most a new frame bucket)
    one(3, 8)  # evicted long ago: captured again, still right
    assert ge.captures > before and len(ge._pre) <= 3


def test_overlap_debug_mode_catches_an_unmaterialised_input():
    """WETTS_DEBUG_OVERLAP: with overlap on, a call whose ids are still being produced by a device op on the caller's
    stream is a race (the side stream does not wait for it); the debug mode turns it into an exception, and leaves a
    correct call alone."""
    net, case, cfg, sd, W = _model("tiny_sdp_b3")
    x, xl, sid = (util.t(case[k]).cuda() for k in ("x", "x_lengths", "sid"))
    net.set_overlap(True)
    net._debug_overlap = True
    try:
        torch.manual_seed(5)
        o1, *_ = net.infer(x, xl, sid=sid, noise_scale=0.667, noise_scale_w=0.8)  # materialised inputs: fine
        torch.cuda.synchronize()
        big = torch.zeros(64 << 20, device="cuda")
        x2 = torch.full_like(x, 1)
        for _ in range(100):  # keep the caller's stream busy, then produce the ids behind that work
            big.add_(1.0)
        x2.copy_(x, non_blocking=True)
        with pytest.raises(RuntimeError, match="overlap mode"):
            net.infer(x2, xl, sid=sid, noise_scale=0.667, noise_scale_w=0.8)
        torch.cuda.synchronize()
        torch.manual_seed(5)
        o3, *_ = net.infer(x2, xl, sid=sid, noise_scale=0.667, noise_scale_w=0.8)  # now materialised
        assert torch.equal(o1, o3)
    finally:
        net._debug_overlap = False
        net.set_overlap(False)


def test_overlap_mode_pipelines_calls_without_changing_results():
    """set_overlap(True): the encoder stages of a call run on a side stream beside the previous call's decoder.  A
    sequence of back-to-back calls of changing shapes (no synchronisation in between) must return exactly what the
    same calls return one at a time -- audio, alignment, z -- and switching modes mid-stream must be safe."""
    net, case, cfg, sd, W = _model("tiny_sdp_b3")
    g = torch.Generator().manual_seed(4)
    calls = []
    for B, Tx in [(3, 12), (1, 5), (4, 20), (2, 12), (3, 12), (5, 17)]:
        x = torch.randint(0, int(case["n_vocab"]), (B, Tx), generator=g).cuda()
        xl = torch.randint(1, Tx + 1, (B,), generator=g)
        xl[0] = Tx
        calls.append((x, xl.cuda(), torch.randint(0, int(case["n_speakers"]), (B,), generator=g).cuda()))
    torch.cuda.synchronize()

    def run_all(overlap, sync_each):
        net.set_overlap(overlap)
        torch.manual_seed(21)
        outs = []
        for x, xl, sid in calls:
            o, attn, ym, (z, *_r) = net.infer(x, xl, sid=sid, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8)
            outs.append((o, attn, z))
            if sync_each:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return [tuple(t.cpu() for t in r) for r in outs]

    ref = run_all(False, True)
    piped = run_all(True, False)
    again = run_all(False, False)
    for a, b, c in zip(ref, piped, again):
        for ta, tb, tc in zip(a, b, c):
            assert ta.shape == tb.shape == tc.shape and torch.equal(ta, tb) and torch.equal(ta, tc)
    net.set_overlap(False)


# ---- is_onnx: the iSTFT head of the EXPORTED Vocos graphs (export_onnx.py:59 -> decoders.py:300-301 -> utils/stft.py:325-340)
def _masked_z_time_major(case):
    return np.ascontiguousarray((case["z"] * case["y_mask"]).transpose(0, 2, 1))


@pytest.mark.parametrize("name", ["vocos_onnx_b2", "tiny_vocos_onnx_b2", "vits2_vocos_onnx_b2x64"])
def test_decoder_session_on_a_vocos_model_computes_the_exported_graphs_istft(name):
    """A DecoderSession stands for decoder_*.onnx, which export_onnx.py traces from an is_onnx=True module: on a Vocos
    model built WITHOUT the flag (the PyTorch CLI's module) the session must still return the OnnxSTFT.inverse audio
    -- the reference golden of the is_onnx model -- and leave the module's own arithmetic as it found it."""
    from wetts_amd import SynthesizerTrn
    from wetts_amd.session import DecoderSession
    case = util.load_case(name)
    cfg, sd, W, _ = util.case_model(case)
    plain = dict(util.model_dict(case))
    plain.pop("is_onnx")
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]), **plain)
    net.load_state_dict(sd).to("cuda")
    assert net.is_onnx is False
    z = _masked_z_time_major(case)
    for use_graph in (False, True):
        got = DecoderSession(net, use_graph=use_graph).run(None, {"z": z, "sid": case["sid"]})[0]
        assert got.shape == case["audio"].shape
        err = util.rms(got - case["audio"])
        print(name, "DecoderSession graph" if use_graph else "DecoderSession", "vs the is_onnx reference golden: abs rms", err)
        assert err < 1e-4
        assert net.is_onnx is False
    # the module itself still computes torch.istft: 1 / 0.375 louder in the interior (hop = n_fft / 4)
    own = net.export_decoder_forward(torch.from_numpy(z).cuda(), torch.from_numpy(case["sid"]).cuda()).cpu().numpy()
    nf = int(cfg.istft_n_fft)
    if own.shape[-1] > 2 * nf:
        ratio = util.rms(case["audio"][..., nf:-nf]) / util.rms(own[..., nf:-nf])
        assert abs(ratio - 0.375) < 1e-3, ratio
    # is_onnx=False at construction: the module's arithmetic (a graph traced without export_onnx.py:59)
    same = DecoderSession(net, is_onnx=False).run(None, {"z": z, "sid": case["sid"]})[0]
    assert np.array_equal(same, own)


def test_inference_session_on_a_vocos_model_equals_export_forward_of_the_is_onnx_module():
    from wetts_amd import SynthesizerTrn
    from wetts_amd.session import InferenceSession
    case = util.load_case("tiny_vocos_onnx_b2")
    cfg, sd, W, _ = util.case_model(case)
    plain = dict(util.model_dict(case))
    plain.pop("is_onnx")
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]), **plain)
    net.load_state_dict(sd).to("cuda")
    B = case["x"].shape[0]
    scales = np.tile(np.array([[0.667, 1.0, 0.8]], np.float32), (B, 1))
    feeds = {"input": case["x"], "input_lengths": case["x_lengths"], "scales": scales, "sid": case["sid"]}
    torch.manual_seed(5)
    a = InferenceSession(net).run(None, feeds)[0]
    assert net.is_onnx is False
    net.set_is_onnx(True)
    torch.manual_seed(5)
    b = net.export_forward(torch.from_numpy(case["x"]).cuda(), torch.from_numpy(case["x_lengths"]).cuda(),
                           torch.from_numpy(scales), torch.from_numpy(case["sid"]).cuda()).cpu().numpy()
    net.set_is_onnx(False)
    torch.manual_seed(5)
    c = net.export_forward(torch.from_numpy(case["x"]).cuda(), torch.from_numpy(case["x_lengths"]).cuda(),
                           torch.from_numpy(scales), torch.from_numpy(case["sid"]).cuda()).cpu().numpy()
    assert np.array_equal(a, b) and a.shape == c.shape and not np.allclose(a, c, atol=1e-6)
    # bucketed sub-batches go through infer() inside the session too
    torch.manual_seed(5)
    d = InferenceSession(net, max_pad_frac=0.0).run(None, feeds)[0]
    assert net.is_onnx is False and d.shape[0] == B and np.isfinite(d).all()


def test_streamed_vocos_decode_matches_the_references_client_on_the_exported_model():
    """tests/golden/vocos_onnx_stream_kat.npz: the reference's own chunk loop (inference_onnx.py:37-76,146-158) over
    export_decoder_forward of an is_onnx=True vits2_vocos_v1 module.  stream_decode over a DecoderSession must give
    that stream sample for sample (1e-4 abs RMS), edges of every window included."""
    from wetts_amd import SynthesizerTrn
    from wetts_amd.session import DecoderSession, stream_decode
    case = util.load_case("vocos_onnx_stream_kat")
    cfg, sd, W, _ = util.case_model(case)
    plain = dict(util.model_dict(case))
    plain.pop("is_onnx")
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]), **plain)
    net.load_state_dict(sd).to("cuda")
    dec = DecoderSession(net)
    whole = dec.run(None, {"z": case["z"], "sid": case["sid"]})[0]
    assert util.rms(whole - case["whole"]) < 1e-4
    for block, pad in ((40, 10), (16, 4)):
        got = np.concatenate(list(stream_decode(dec, case["z"], case["sid"], chunk_size=block, pad_size=pad)))
        want = case[f"stream_{block}_{pad}"][0]
        assert got.shape == want.shape
        err = util.rms(got - want)
        print("streamed Vocos (exported arithmetic)", block, pad, "abs rms vs the reference client", err)
        assert err < 1e-4


def test_istft_mode_through_the_c_abi_and_the_native_infer_entry():
    """wetts_set_istft_mode / wetts_get_istft_mode on a live handle: the config's is_onnx is the mode at create, an
    invalid mode is refused without changing anything, a HiFi-GAN model takes the call without effect, and
    wetts_infer -- the native hosts' entry -- follows the handle's mode (its audio equals the Python composition's in
    either mode, and the two modes differ)."""
    from wetts_amd import _lib
    lib = _lib.load()
    net, case, cfg, sd, W = _model("tiny_vocos_onnx_b2")
    assert lib.wetts_get_istft_mode(net._handle) == 1 and net.is_onnx
    assert lib.wetts_set_istft_mode(net._handle, 7) != 0 and "istft mode" in _lib.last_error()
    assert lib.wetts_get_istft_mode(net._handle) == 1
    z = torch.from_numpy((case["z"] * case["y_mask"])).cuda()
    g = torch.nn.functional.embedding(util.t(case["sid"]), W["emb_g.weight"]).cuda()
    a_onnx = net.hifigan(z, g).cpu().numpy()
    assert util.rms(a_onnx - case["audio"]) < 1e-4
    net.set_is_onnx(False)
    assert lib.wetts_get_istft_mode(net._handle) == 0
    a_torch = net.hifigan(z, g).cpu().numpy()
    ref = util.load_case("tiny_vocos_b2")  # same weights, inputs and noise: the torch.istft head's golden
    assert util.rms(a_torch - ref["audio"]) < 1e-4 and util.rms(a_torch - a_onnx) > 1e-3
    hnet, *_ = _model("tiny_sdp_b3")  # HiFi-GAN: the flag exists and does nothing
    zz = torch.randn(1, hnet.inter_channels, 9, device="cuda")
    gg = torch.zeros(1, hnet.gin_channels, device="cuda")
    before = hnet.hifigan(zz, gg).cpu().numpy()
    hnet.set_is_onnx(True)
    assert np.array_equal(hnet.hifigan(zz, gg).cpu().numpy(), before)
