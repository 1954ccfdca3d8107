"""GPU tier: the HIP path (through the C ABI) against the golden vectors of the real reference
and against the oracle on the same seeded inputs.  Tolerance: BASELINE.json's north_star gate is
waveform RMS error <= 1e-3 at fp32; the f32-MFMA path is an exact-f32 fma chain, so we hold it
to 1e-4 absolute and 2e-3 relative (round-off of ~70 stacked convs)."""
import json
import os

import numpy as np
import pytest
import torch

from tests import util

pytestmark = pytest.mark.gpu

ABS_RMS_GATE = 1e-3  # north_star
ABS_RMS_OURS = 1e-4


def _model(case):
    from wetts_amd import SynthesizerTrn, config
    cfg, sd, W, blob = util.case_model(case)
    net = SynthesizerTrn(int(case["n_vocab"]), 513, 32, n_speakers=int(case["n_speakers"]),
                         **util.model_dict(case))
    net.load_state_dict(sd)
    net.to("cuda")
    return net, cfg, W


def _report(name, rows):
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"parity_{name}.json"), "w") as f:
        json.dump(rows, f, indent=1)


_GOLDEN_CASES = util.INFER_CASES  # every committed reference golden


@pytest.mark.parametrize("name", _GOLDEN_CASES)
def test_infer_matches_reference_golden(name):
    case = util.load_case(name)
    net, cfg, W = _model(case)
    ns, ls, nsw = [float(v) for v in case["scales"]]
    dev = "cuda"
    o, attn, y_mask, (z, z_p, m_p, logs_p) = net.infer(
        util.t(case["x"]).to(dev), util.t(case["x_lengths"]).to(dev),
        sid=util.t(case["sid"]).to(dev), noise_scale=ns, length_scale=ls, noise_scale_w=nsw,
        eps_w=util.t(case["eps_w"]).to(dev), eps_z=util.t(case["eps_z"]).to(dev))
    st = net._last
    rows = {}
    # stage-wise report first (so a failure shows where parity is lost), asserts after
    rows["x_enc"] = util.rel_rms(st["x_enc"].cpu().numpy(), case["x_enc"])
    I = cfg.inter_channels
    rows["m_p"] = util.rel_rms(st["stats"][:, :I].cpu().numpy(), case["m_p"])
    rows["logs_p"] = util.rel_rms(st["stats"][:, I:].cpu().numpy(), case["logs_p"])
    rows["logw_maxabs"] = float(np.abs(st["logw"].cpu().numpy() - case["logw"][:, 0]).max())
    rows["y_mask_equal"] = bool(np.array_equal(y_mask.cpu().numpy(), case["y_mask"]))
    same_T = tuple(attn.shape) == tuple(case["attn"].shape)
    rows["attn_equal"] = bool(same_T and np.array_equal(attn.cpu().numpy().astype(np.uint8),
                                                        case["attn"]))
    rows["w_err_x10_vs_ceil_margin"] = [
        10 * float(np.abs(np.exp(st["logw"].cpu().numpy()) - np.exp(case["logw"][:, 0])).max()) * ls,
        float(case["ceil_margin"])]
    if same_T:
        if "z_p" in case:  # the compact full-size fixtures keep z and audio only
            rows["m_p_exp"] = util.rel_rms(m_p.cpu().numpy(), case["m_p_exp"])
            rows["z_p"] = util.rel_rms(z_p.cpu().numpy(), case["z_p"])
        rows["z"] = util.rel_rms(z.cpu().numpy(), case["z"])
        rows["audio_abs_rms"] = util.rms(o.cpu().numpy() - case["audio"])
        rows["audio_rel_rms"] = util.rel_rms(o.cpu().numpy(), case["audio"])
        rows["audio_ref_rms"] = util.rms(case["audio"])
    _report(name, rows)
    print(name, rows)
    assert rows["x_enc"] < 1e-4 and rows["m_p"] < 1e-4 and rows["logs_p"] < 1e-4
    assert rows["logw_maxabs"] < 1e-4
    # y_mask / attn equality is only a fair demand while the error in w = exp(logw)*length_scale is
    # an order of magnitude inside the fixture's ceil() margin
    assert rows["w_err_x10_vs_ceil_margin"][0] < rows["w_err_x10_vs_ceil_margin"][1]
    assert rows["y_mask_equal"] and rows["attn_equal"]
    assert rows.get("z_p", 0.0) < 1e-4 and rows["z"] < 2e-4
    assert rows["audio_abs_rms"] < ABS_RMS_OURS < ABS_RMS_GATE
    assert rows["audio_rel_rms"] < 2e-3


@pytest.mark.parametrize("name", ["tiny_sdp_b3", "v3_b2"])
def test_hifigan_standalone_and_chunked(name):
    """wetts_hifigan on [B,192,L] slices == Generator.forward (the streaming decoder surface,
    export_decoder_forward, models.py:360-363), checked against the oracle."""
    from oracle import vits_oracle as vo
    case = util.load_case(name)
    net, cfg, W = _model(case)
    cd = util.cfg_dict(cfg)
    z = util.t(case["z"]) * util.t(case["y_mask"])
    sid = util.t(case["sid"])
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).unsqueeze(-1)
    with torch.no_grad():
        ref = vo.hifigan(W, cd, z, g).numpy()
    got = net.export_decoder_forward(z.transpose(1, 2).cuda(), sid.cuda()).cpu().numpy()
    assert util.rms(got - ref) < ABS_RMS_OURS
    # a time slice through strides (no copy): first 7 frames
    L = min(7, z.shape[2])
    with torch.no_grad():
        ref_s = vo.hifigan(W, cd, z[:, :, :L], g).numpy()
    o, *_ = net.infer(util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda(),
                      sid=sid.cuda(), noise_scale=float(case["scales"][0]),
                      length_scale=float(case["scales"][1]),
                      noise_scale_w=float(case["scales"][2]), max_len=L,
                      eps_w=util.t(case["eps_w"]).cuda(), eps_z=util.t(case["eps_z"]).cuda())
    assert o.shape[-1] == L * net.hop_length
    assert util.rms(o.cpu().numpy() - ref_s) < 5e-4  # z itself carries flow round-off here


def test_mas_bit_exact_known_answers():
    from wetts_amd import monotonic_align
    d = np.load(util.GOLDEN + "/mas_kat.npz")
    for i in range(int(d["n"])):
        neg = torch.from_numpy(d[f"neg{i}"]).cuda()
        b, ty, tx = neg.shape
        t_y, t_x = torch.from_numpy(d[f"ty{i}"]), torch.from_numpy(d[f"tx{i}"])
        mask = ((torch.arange(ty).view(1, ty, 1) < t_y.view(b, 1, 1)) &
                (torch.arange(tx).view(1, 1, tx) < t_x.view(b, 1, 1))).float().cuda()
        p = monotonic_align.maximum_path(neg, mask).cpu().numpy()
        assert np.array_equal(p.astype(np.int8), d[f"path{i}"]), f"case {i}"


def test_mas_large_matches_oracle_and_properties():
    """Training-shaped sizes (b=16, t_t~800, t_s~128): bit-exact vs the numpy oracle on two
    items, structural properties (one 1 per valid row, monotone, ends at t_x-1) on all."""
    from oracle import vits_oracle as vo
    from wetts_amd import monotonic_align
    g = torch.Generator().manual_seed(3)
    b, ty, tx = 16, 800, 128
    neg = torch.randn(b, ty, tx, generator=g)
    t_y = torch.randint(400, ty + 1, (b,), generator=g)
    t_x = torch.randint(40, tx + 1, (b,), generator=g)
    mask = ((torch.arange(ty).view(1, ty, 1) < t_y.view(b, 1, 1)) &
            (torch.arange(tx).view(1, 1, tx) < t_x.view(b, 1, 1))).float()
    p = monotonic_align.maximum_path(neg.cuda(), mask.cuda()).cpu().numpy()
    ref = vo.maximum_path_numpy(neg[:2].numpy(), t_y[:2].numpy(), t_x[:2].numpy())
    assert np.array_equal(p[:2].astype(np.int32), ref)
    for i in range(b):
        yy, xx = int(t_y[i]), int(t_x[i])
        assert (p[i, :yy].sum(1) == 1).all() and p[i, yy:].sum() == 0
        cols = p[i, :yy].argmax(1)
        d = np.diff(cols)
        assert ((d == 0) | (d == 1)).all() and cols[0] == 0 and cols[-1] == xx - 1


@pytest.mark.parametrize("b,ty,tx", [(5, 300, 50), (3, 700, 128), (4, 500, 200), (2, 400, 300), (2, 300, 600),
                                     (64, 1000, 200)])
def test_mas_wave_kernel_shapes_bit_exact_vs_c_oracle(b, ty, tx):
    """Every column-per-lane variant of mas_wave_kernel (Tx <= 64 / 128 / 256 / 512 / 1024) and the bench's large
    shape against oracle/mas_oracle.c (pinned to the reference's own maximum_path by mas_kat.npz in the CPU tier),
    bit for bit, ragged t_y / t_x, plus a batch item with all-equal scores (ties: the strict `<` of the
    backtrack) and, in the small cases, items with t_x > t_y -- not a valid alignment problem, but the reference
    returns something for it (wrapped row reads) and so does the general kernel those items fall back to."""
    from tests.test_oracle_golden import _c_mas
    from wetts_amd import monotonic_align
    run = _c_mas()
    g = torch.Generator().manual_seed(b * 7 + tx)
    neg = torch.randn(b, ty, tx, generator=g)
    neg[b - 1] = 0.25
    t_y = torch.randint(ty // 2, ty + 1, (b,), generator=g)
    t_x = torch.minimum(torch.randint(1, tx + 1, (b,), generator=g), t_y)
    if b <= 5 and tx < ty:
        t_y[0] = max(1, tx // 2)  # t_x > t_y
        t_x[0] = tx
    t_y[b - 1], t_x[b - 1] = ty, min(tx, ty)
    mask = ((torch.arange(ty).view(1, ty, 1) < t_y.view(b, 1, 1)) &
            (torch.arange(tx).view(1, 1, tx) < t_x.view(b, 1, 1))).float()
    p = monotonic_align.maximum_path(neg.cuda(), mask.cuda()).cpu().numpy().astype(np.int32)
    ref = run(neg.numpy(), t_y.numpy().astype(np.int32), t_x.numpy().astype(np.int32))
    for i in range(b):
        assert np.array_equal(p[i], ref[i]), f"item {i}: t_y={int(t_y[i])} t_x={int(t_x[i])}"


def test_audio_to_int16():
    from oracle import vits_oracle as vo
    case = util.load_case("tiny_sdp_b3")
    net, _, _ = _model(case)
    a = util.t(case["audio"])
    pcm = net.audio_to_int16(a.cuda()).cpu().numpy()
    for b in range(a.shape[0]):
        ref = vo.audio_to_int16(a[b, 0].numpy())
        assert np.abs(pcm[b].astype(np.int32) - ref.astype(np.int32)).max() <= 1


def test_product_path_fails_loudly_without_device_weights():
    from wetts_amd import SynthesizerTrn, _lib, config
    net = SynthesizerTrn(10, 513, 32, n_speakers=0, **config.MODEL_CONFIGS["tiny"])
    with pytest.raises(_lib.WettsError):
        net.infer(torch.zeros(1, 3, dtype=torch.long), torch.tensor([3]))


@pytest.mark.parametrize("name", ["v1_b2", "v3_b2"])
def test_hifigan_f16_mode_matches_its_numerics_spec(name):
    """IEEE-half decoder (configs[4] precision): 11-bit mantissa => ~8x tighter than bf16."""
    from oracle import vits_oracle as vo
    case = util.load_case(name)
    net, cfg, W = _model(case)
    cd = util.cfg_dict(cfg)
    z = util.t(case["z"]) * util.t(case["y_mask"])
    sid = util.t(case["sid"])
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).unsqueeze(-1)
    with torch.no_grad():
        spec = vo.hifigan_16bit_sim(W, cd, z, g, torch.float16).numpy()
        f32 = vo.hifigan(W, cd, z, g).numpy()
    net.set_decoder_dtype(torch.float16)
    got = net.hifigan(z.cuda(), g[:, :, 0].cuda()).cpu().numpy()
    r_spec, r_f32 = util.rel_rms(got, spec), util.rel_rms(got, f32)
    print(name, "f16 vs spec", r_spec, "f16 vs f32", r_f32)
    assert np.isfinite(got).all() and r_spec < 2e-3 and r_f32 < 1e-2


@pytest.mark.parametrize("name", ["v1_b2", "v3_b2"])
def test_hifigan_bf16_mode_matches_its_numerics_spec(name):
    """bf16 decoder (configs[2]/[4] precision): against oracle.hifigan_bf16sim (same rounding
    points) and, loosely, against the f32 oracle.  Tolerances: rel RMS 1e-2 vs the bf16 spec
    (f32-accumulation order can flip an occasional bf16 rounding), 6e-2 vs f32."""
    from oracle import vits_oracle as vo
    case = util.load_case(name)
    net, cfg, W = _model(case)
    cd = util.cfg_dict(cfg)
    z = util.t(case["z"]) * util.t(case["y_mask"])
    sid = util.t(case["sid"])
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).unsqueeze(-1)
    with torch.no_grad():
        spec = vo.hifigan_bf16sim(W, cd, z, g).numpy()
        f32 = vo.hifigan(W, cd, z, g).numpy()
    net.set_decoder_dtype(torch.bfloat16)
    got = net.hifigan(z.cuda(), g[:, :, 0].cuda()).cpu().numpy()
    net.set_decoder_dtype(torch.float32)
    back = net.hifigan(z.cuda(), g[:, :, 0].cuda()).cpu().numpy()
    r_spec, r_f32 = util.rel_rms(got, spec), util.rel_rms(got, f32)
    print(name, "bf16 vs spec", r_spec, "bf16 vs f32", r_f32, "spec vs f32", util.rel_rms(spec, f32))
    assert got.shape == f32.shape and np.isfinite(got).all()
    assert r_spec < 1e-2
    assert r_f32 < 6e-2
    assert util.rms(back - f32) < ABS_RMS_OURS  # switching back restores the exact-f32 path


@pytest.mark.parametrize("cname", ["v1_b2", "v3_b2", "v3_b2:stage"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("L", [1, 37, 200])
def test_fused_resblock_pair_bit_identical(dtype, L, cname):
    """The fused ResBlock kernels (resblock_chain32.hip / resblock32.hip at f32; resblock16.hip pairs / ResBlock2
    chains and resblock2_stage16.hip whole stages at 16 bit) perform the
    arithmetic of two conv launches in the same order with the same rounding points: outputs must
    be EQUAL, including at tile seams
    (L*hop spans several time tiles), sequence ends (zero padding of c2's input) and L=1."""
    # v1: ResBlock1 pairs; v3: ResBlock2 chains (RB2 instantiations), one launch per chain -- or, ":stage", one launch
    # per STAGE at 16 bit (resblock2_stage16.hip: all chains of a stage, the running sum in registers)
    cname, _, variant = cname.partition(":")
    if variant == "stage" and dtype == torch.float32:
        pytest.skip("the stage kernel is a 16-bit decoder kernel")
    stage_pct = 100 if variant == "stage" else 0
    case = util.load_case(cname)
    # also fuse launches too small to fill the chip, ResBlock2 shapes with a wide second halo, and whole
    # ResBlock1 chains whatever their halo costs (every fused kernel must be exercised here)
    os.environ["WETTS_TUNE"] = ("fuse_min_blocks=0,fuse2_waste_pct=100,chain_whole_pct=100,chain_whole_maxc=128,"
                                f"stage2_pct={stage_pct},"
                                "small_max_tiles=0")  # (conv_small_kernel sums K in another order)
    try:
        net, cfg, W = _model(case)
    finally:
        del os.environ["WETTS_TUNE"]
    torch.manual_seed(5)
    z = torch.randn(2, cfg.inter_channels, L)
    g = torch.nn.functional.embedding(util.t(case["sid"]), W["emb_g.weight"])
    net.set_decoder_dtype(dtype, fused=True)
    a = net.hifigan(z.cuda(), g.cuda()).cpu().numpy()
    net.set_decoder_dtype(dtype, fused=False)
    b = net.hifigan(z.cuda(), g.cuda()).cpu().numpy()
    net.set_decoder_dtype(torch.float32)
    assert a.shape == b.shape and np.isfinite(a).all()
    assert np.array_equal(a, b), f"max |diff| {np.abs(a - b).max()}"


@pytest.mark.parametrize("cname,B,L", [("v1_b2", 4, 300), ("v1_b2", 2, 37), ("v2_b2", 3, 200), ("stress48k_b2", 2, 150)])
def test_three_stream_fork_of_the_chains_is_bit_identical_to_one_stream(cname, B, L):
    """The default f32 decoder schedule runs the k = 3 / 7 / 11 ResBlock chains of a stage on three HIP streams (forked
    from / joined to the caller's stream; the running MRF sum is ordered chain j - 1 -> chain j by events);
    WETTS_DECODER_SERIAL keeps every launch on the caller's stream with the grouped launches of round 4.  Same kernels on
    the same values in the same sum order: the audio must be EQUAL -- on big launches (whole-chain kernels, grouped
    convs), on launches of a few tiles, and when called twice back to back (stream re-use)."""
    case = util.load_case(cname)
    net, cfg, W = _model(case)
    torch.manual_seed(6)
    z = torch.randn(B, cfg.inter_channels, L).cuda()
    sid = torch.zeros(B, dtype=torch.long)
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).cuda() if "emb_g.weight" in W else None
    net.set_decoder_dtype(torch.float32, serial=True)
    a = net.hifigan(z, g).cpu().numpy()
    net.set_decoder_dtype(torch.float32, serial=False)
    b1 = net.hifigan(z, g)
    b2 = net.hifigan(z, g)  # no synchronisation in between: the second call re-uses the aux streams and events
    torch.cuda.synchronize()
    assert np.isfinite(a).all() and np.array_equal(a, b1.cpu().numpy()) and np.array_equal(a, b2.cpu().numpy())


@pytest.mark.parametrize("name", ["tiny_sdp_b3", "v1_b2", "v1_b4x128"])
def test_wn_update_in_the_conv_epilogue_is_bit_identical(name):
    """The f32 flow's residual / skip update (modules.py:79-86) runs in the epilogue of the res_skip conv
    (ConvParams.wn_*; conv_small_kernel for short calls, pw_gemm_kernel for B = 4 x 128) instead of wn_update_kernel:
    for the short calls the same additions on the same values, so z and the audio must be EQUAL."""
    case = util.load_case(name)
    outs = []
    for fuse in ("2", "0"):  # 2 = at every size (1, the default, fuses only the small launches)
        os.environ["WETTS_TUNE"] = "wn_fuse=" + fuse
        try:
            net, cfg, W = _model(case)
        finally:
            del os.environ["WETTS_TUNE"]
        ns, ls, nsw = [float(v) for v in case["scales"]]
        o, _, _, (z, *_r) = net.infer(util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda(),
                                      sid=util.t(case["sid"]).cuda(), noise_scale=ns, length_scale=ls,
                                      noise_scale_w=nsw, eps_w=util.t(case["eps_w"]).cuda(),
                                      eps_z=util.t(case["eps_z"]).cuda())
        outs.append((z.cpu().numpy(), o.cpu().numpy()))
    assert np.isfinite(outs[0][0]).all()
    if name == "v1_b4x128":
        # launches above the small-launch limit take the LDS-DMA GEMM (gemm_pw.hip), where the old h / skip values come in
        # through the accumulator init: (h + conv) + bias instead of h + (conv + bias) -- one rounding apart per element
        assert util.rel_rms(outs[0][0], outs[1][0]) < 1e-6 and util.rel_rms(outs[0][1], outs[1][1]) < 1e-5
        return
    assert np.array_equal(outs[0][0], outs[1][0]), f"z: max |diff| {np.abs(outs[0][0] - outs[1][0]).max()}"
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("name", ["tiny_sdp_b3", "v1_b2", "v1_b4x128", "aishell3_b4x128", "tiny_mono_post_b2x64"])
def test_wn_gate_in_the_conv_epilogue_is_bit_identical(name):
    """The f32 flow's gate tanh(a[:H]) * sigmoid(a[H:]) (commons.py:98-105) runs in the epilogue of the in_layer conv
    (OUT_GATE: weight rows packed interleaved so that a lane holds both halves of an output row; conv_small_kernel for
    short calls, the 64x64 tiles above) instead of gate_kernel on the 2H-row tensor: the same sums, the same bias order,
    the same tanhf / expf, so z and the audio must be EQUAL -- with and without the speaker term."""
    case = util.load_case(name)
    outs = []
    for gate in ("1", "0"):
        os.environ["WETTS_TUNE"] = "wn_gate=" + gate
        try:
            net, cfg, W = _model(case)
        finally:
            del os.environ["WETTS_TUNE"]
        ns, ls, nsw = [float(v) for v in case["scales"]]
        o, _, _, (z, *_r) = net.infer(util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda(),
                                      sid=util.t(case["sid"]).cuda(), noise_scale=ns, length_scale=ls,
                                      noise_scale_w=nsw, eps_w=util.t(case["eps_w"]).cuda(),
                                      eps_z=util.t(case["eps_z"]).cuda())
        outs.append((z.cpu().numpy(), o.cpu().numpy()))
    assert np.isfinite(outs[0][0]).all()
    assert np.array_equal(outs[0][0], outs[1][0]), f"z: max |diff| {np.abs(outs[0][0] - outs[1][0]).max()}"
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("mode", ["1", "2"])
@pytest.mark.parametrize("name", ["tiny_sdp_b3", "v1_b4x128", "aishell3_b4x128"])
def test_dds_fused_kernel_matches_the_layerwise_path(name, mode):
    """dds_fused.hip (a whole DDSConv of the stochastic duration predictor in one launch: 48 launches -> 4 per
    utterance; mode 1 = the 32-column tiles small launches take by default, mode 2 = the 64-column tiles) against
    the conv-by-conv path: per column the same arithmetic except the
    summation order of the 1x1 convs (the small-launch kernel splits K over its waves), so logw agrees to 1e-5 and
    the durations -- ceil(exp(logw) * length_scale) -- do not move.  Ragged lengths and tile seams (Tx = 128 is 22 tiles
    of 6 valid columns / 4 of 38, 13-column halos) are in the cases."""
    case = util.load_case(name)
    outs = []
    for fused in (mode, "0"):
        os.environ["WETTS_TUNE"] = "dds_fused=" + fused
        try:
            net, cfg, W = _model(case)
        finally:
            del os.environ["WETTS_TUNE"]
        ns, ls, nsw = [float(v) for v in case["scales"]]
        net.infer(util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda(), sid=util.t(case["sid"]).cuda(),
                  noise_scale=ns, length_scale=ls, noise_scale_w=nsw, eps_w=util.t(case["eps_w"]).cuda(),
                  eps_z=util.t(case["eps_z"]).cuda())
        outs.append((net._last["logw"].cpu().numpy(), net._last["w_ceil"].cpu().numpy()))
    d = float(np.abs(outs[0][0] - outs[1][0]).max())
    print(name, "fused vs layerwise logw max |diff|", d)
    assert np.isfinite(outs[0][0]).all() and d < 1e-5
    assert np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("L", [3, 50, 200])
def test_small_launch_conv_schedule_matches_the_big_tile_kernel(L):
    """conv_small_kernel (32x32 tiles, K split over the block's waves; B = 1 streaming windows and short
    texts) against conv_mfma_kernel on the same decoder input: same arithmetic, another summation
    order.  L = 200 at B = 2 mixes both kernels inside one decode (the C = 256 stage is below the
    256-tile threshold, the later stages above it)."""
    case = util.load_case("v1_b2")
    os.environ["WETTS_TUNE"] = "small_max_tiles=0"
    try:
        net, cfg, W = _model(case)  # (a per-model setting, read at create)
    finally:
        del os.environ["WETTS_TUNE"]
    torch.manual_seed(11)
    z = torch.randn(2, cfg.inter_channels, L).cuda()
    g = torch.nn.functional.embedding(util.t(case["sid"]), W["emb_g.weight"]).cuda()
    big = net.hifigan(z, g).cpu().numpy()
    net2, _, _ = _model(case)
    small = net2.hifigan(z, g).cpu().numpy()
    assert big.shape == small.shape and np.isfinite(small).all()
    r = util.rel_rms(small, big)
    print("small-launch schedule vs big tiles, L =", L, "rel rms", r)
    # ~60 convs deep, each summing K = 100..2800 terms in another order: 1e-6-level differences
    assert r < 1e-5 and not np.array_equal(small, big)  # (equal would mean the switch did nothing)


@pytest.mark.parametrize("name,dtype", [("v3_b3x128", torch.bfloat16), ("v3_b3x128", torch.float16),
                                        ("v1_b4x128", torch.bfloat16)])
def test_flow_16bit_mode_matches_its_numerics_spec(name, dtype):
    """16-bit WaveNet layers of the flow (wetts_set_flow_precision; BASELINE configs[2] precision for
    the part of the step that dominates at B = 64): against oracle.wn_16bit_sim (same rounding
    points, f32 accumulation) and, loosely, against the reference's f32 z.  The alignment (duration
    path) must not move at all; switching back restores the exact-f32 flow."""
    from oracle import vits_oracle as vo
    case = util.load_case(name)
    net, cfg, W = _model(case)
    cd = util.cfg_dict(cfg)
    ns, ls, nsw = [float(v) for v in case["scales"]]
    args = dict(sid=util.t(case["sid"]).cuda(), noise_scale=ns, length_scale=ls, noise_scale_w=nsw,
                eps_w=util.t(case["eps_w"]).cuda(), eps_z=util.t(case["eps_z"]).cuda())
    x, xl = util.t(case["x"]).cuda(), util.t(case["x_lengths"]).cuda()
    o32, attn32, ym32, (z32, zp32, _, _) = net.infer(x, xl, **args)
    net.set_flow_dtype(dtype)
    o16, attn16, ym16, (z16, zp16, _, _) = net.infer(x, xl, **args)
    net.set_flow_dtype(torch.float32)
    o_back, *_ = net.infer(x, xl, **args)
    assert torch.equal(attn16, attn32) and torch.equal(ym16, ym32) and torch.equal(zp16, zp32)
    assert torch.equal(o_back, o32)
    sid = util.t(case["sid"])
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).unsqueeze(-1) if cfg.n_speakers > 0 else None
    with torch.no_grad():
        spec = vo.flow_reverse(W, cd, zp32.cpu(), ym32.cpu(), g, wn_dtype=dtype)
    valid = ym32.cpu().bool().expand_as(z16.cpu()).numpy()
    r_spec = util.rel_rms(z16.cpu().numpy()[valid], spec.numpy()[valid])
    r_f32 = util.rel_rms(z16.cpu().numpy()[valid], case["z"][valid])
    r_audio = util.rel_rms(o16.cpu().numpy(), case["audio"])
    print(name, dtype, "z vs 16-bit spec", r_spec, "z vs reference f32", r_f32, "audio vs reference", r_audio)
    tol_spec, tol_f32 = (1e-2, 3e-2) if dtype == torch.bfloat16 else (2e-3, 5e-3)
    assert np.isfinite(z16.cpu().numpy()).all()
    assert r_spec < tol_spec and r_f32 < tol_f32


@pytest.mark.parametrize("B,Cin,Cout,k,dil,T", [(2, 32, 32, 3, 1, 300), (3, 192, 64, 7, 1, 77),
                                                (1, 64, 64, 11, 5, 1000), (2, 256, 1, 1, 1, 1),
                                                (2, 40, 96, 5, 3, 129), (1, 32, 1, 7, 1, 513),
                                                # round 6 (LDS-tile conv, flat range pass): an odd element count (the
                                                # general range kernel), the widest tile (k = 11 at dilation 5, C = 128),
                                                # two m-tiles of rows (C = 256), a tile wider than the sequence
                                                (1, 33, 32, 3, 2, 131), (1, 128, 128, 11, 5, 700),
                                                (1, 256, 256, 7, 3, 333), (2, 32, 32, 11, 5, 40),
                                                # a tile above the default 64 KB of dynamic LDS (the opt-in path)
                                                (1, 512, 128, 3, 1, 200)])
def test_dynamic_quant_conv1d_is_bit_exact_to_its_restatement(B, Cin, Cout, k, dil, T):
    """The uint8 dynamic-quantisation conv (qconv_u8.hip, v_mfma_i32_32x32x32_i8) against the exact-integer
    restatement of DynamicQuantizeLinear -> ConvInteger -> scale + bias (oracle.dynamic_quant_conv1d):
    integer work => array_equal.  Covers channel counts that need padding to 32, a single output channel
    (conv_post), T = 1 (the speaker conditioning conv), tiles straddling the sequence end, dilation."""
    from oracle import vits_oracle as vo
    from wetts_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(B * 1000 + Cin + k)
    x = torch.randn(B, Cin, T, generator=g) * 0.7 + 0.1
    w = torch.randn(Cout, Cin, k, generator=g) / (Cin * k) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    pad = (k - 1) * dil // 2
    with torch.no_grad():
        ref = vo.dynamic_quant_conv1d(x, w, b, dilation=dil, padding=pad)
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    out = torch.empty(B, Cout, T, device="cuda")
    _lib.check(lib.wetts_dynamic_quant_conv1d(_lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), B, Cin, Cout, k, dil,
                                              pad, T, _lib.ptr(out), None), "dynamic_quant_conv1d")
    got = out.cpu()
    assert torch.isfinite(got).all()
    assert torch.equal(got, ref), float((got - ref).abs().max())


def test_dynamic_quant_conv1d_reproduces_the_onnx_spec_known_answers():
    """The pin of the uint8 variant: the published ONNX operator-specification examples of
    DynamicQuantizeLinear-11 and ConvInteger-10 (tests/test_oracle_golden.py holds the literals and checks the
    oracle against them on the CPU) through the HIP conv node `wetts_dynamic_quant_conv1d`.
    * DynamicQuantizeLinear: x = the spec's X as [1,1,N]; a 1x1 weight tensor [[1],[255]] quantises to
      w_q = [1,255], s_w = 1, z_w = 0, so output channel 0 = (Y - Y_ZeroPoint) * Y_Scale, bit for bit.
    * ConvInteger: the spec's uint8 x (3x3, zero point 1) and 2x2 kernel of ones in the 1-D form of
      onnx_convinteger_as_conv1d(); float x = (x_q - 1) / 4 plus one extra batch item holding q = 0 and q = 255,
      which makes DynamicQuantizeLinear return exactly that x_q with scale 1/4 and zero point 1; a second output
      channel holding 255 pins s_w = 1.  Channel 0 of items 0..3 = the spec's y_with_padding / 4 (and its inner
      2x2 = the unpadded y)."""
    from tests.test_oracle_golden import ONNX_DQL_KATS, ONNX_CONVINT_XZP, onnx_convinteger_as_conv1d
    from wetts_amd import _lib
    lib = _lib.load()

    def run(x, w, k, pad):
        B, Cin, T = x.shape
        Cout = w.shape[0]
        xd, wd = x.cuda().contiguous(), w.cuda().contiguous()
        out = torch.full((B, Cout, T), float("nan"), device="cuda")
        _lib.check(lib.wetts_dynamic_quant_conv1d(_lib.ptr(xd), _lib.ptr(wd), None, B, Cin, Cout, k, 1, pad, T,
                                                  _lib.ptr(out), None), "dynamic_quant_conv1d")
        return out.cpu().numpy()

    for name, X, shape, Y, scale, zp in ONNX_DQL_KATS:
        x = torch.tensor(X, dtype=torch.float32).view(1, 1, -1)
        got = run(x, torch.tensor([[[1.0]], [[255.0]]]), 1, 0)
        want = (np.array(Y, np.float32) - np.float32(zp)) * np.float32(scale)
        assert np.array_equal(got[0, 0], want), (name, got[0, 0], want)
        # channel 1 (w_q = 255): the int32 sum is (Y - zp) * 255, THEN the cast and one multiply by s_x * s_w
        want1 = ((np.array(Y, np.int32) - zp) * 255).astype(np.float32) * np.float32(scale)
        assert np.array_equal(got[0, 1], want1), (name, got[0, 1], want1)

    xq, wq, want = onnx_convinteger_as_conv1d()
    s = np.float32(0.25)
    x = (xq.astype(np.float32) - ONNX_CONVINT_XZP) * s
    pin = np.full((1, 2, 4), 0.0, np.float32)
    pin[0, 0, 0], pin[0, 1, 3] = (0 - ONNX_CONVINT_XZP) * s, (255 - ONNX_CONVINT_XZP) * s
    w = np.concatenate([wq.astype(np.float32), np.zeros((1, 2, 3), np.float32)])
    w[1, 0, 1] = 255.0
    got = run(torch.from_numpy(np.concatenate([x, pin])), torch.from_numpy(w), 3, 1)
    assert np.array_equal(got[:4, 0], want[:, 0].astype(np.float32) * s), got[:4, 0]


@pytest.mark.parametrize("name", ["tiny_sdp_b3", "v1_b2", "v3_b2"])
def test_hifigan_uint8_dynamic_variant(name):
    """Decoder precision 3 = the `export_onnx.py --quant` graph (export_onnx.py:149-157): every Conv1d
    dynamically quantised to uint8, ConvTranspose1d in f32.  Against the oracle's restatement of that
    graph (parity with onnxruntime itself is unpinned: ORT is not in the reference tree).  The integer
    convs are exact; the f32 ConvTranspose1d in between differs from ATen's by round-off, which can move
    a value across a quantisation step now and then => a small tolerance, and a loose one vs f32."""
    from oracle import vits_oracle as vo
    case = util.load_case(name)
    net, cfg, W = _model(case)
    cd = util.cfg_dict(cfg)
    z = util.t(case["z"]) * util.t(case["y_mask"])
    sid = util.t(case["sid"])
    g = torch.nn.functional.embedding(sid, W["emb_g.weight"]).unsqueeze(-1)
    with torch.no_grad():
        spec = vo.hifigan_uint8_dynamic(W, cd, z, g).numpy()
        f32 = vo.hifigan(W, cd, z, g).numpy()
    net.set_decoder_dtype(torch.uint8)
    got = net.hifigan(z.cuda(), g[:, :, 0].cuda()).cpu().numpy()
    net.set_decoder_dtype(torch.float32)
    back = net.hifigan(z.cuda(), g[:, :, 0].cuda()).cpu().numpy()
    r_spec, r_f32 = util.rel_rms(got, spec), util.rel_rms(got, f32)
    print(name, "uint8 vs restatement", r_spec, "uint8 vs f32", r_f32, "restatement vs f32", util.rel_rms(spec, f32))
    assert got.shape == f32.shape and np.isfinite(got).all()
    # vs the restatement: the integer convs are bit-exact (test above); what differs is the f32 arithmetic
    # between them (ConvTranspose, adds), whose last-bit differences flip a few activations across a
    # quantisation boundary of the NEXT conv.  Measured 2e-3 .. 5e-3 depending on the summation order of
    # the f32 kernels in use; the scheme's own distance from f32 is 1.3e-2 .. 2.6e-2.
    assert r_spec < 1.2e-2 and r_f32 < 0.15
    assert util.rms(back - f32) < ABS_RMS_OURS
