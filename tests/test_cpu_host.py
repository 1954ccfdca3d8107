"""CPU tier: the C-ABI library loads and exports every symbol include/wetts_hip.h declares,
the blob layout agrees with the reference's state_dict, host logic behaves."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests import util
from wetts_amd import _lib, checkpoint, config, synth

HEADER = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include",
                      "wetts_hip.h")


def test_library_exports_every_declared_symbol():
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(wetts_[a-z0-9_]+)\s*\(", src))
    declared -= {"wetts_config", "wetts_model"}
    lib = C.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.load().wetts_abi_version() == _lib.ABI_VERSION
    m = re.search(r"#define WETTS_ABI_VERSION (\d+)", src)
    assert m and int(m.group(1)) == _lib.ABI_VERSION  # header and ctypes binding move together


@pytest.mark.parametrize("mname", ["v1", "v2", "v3", "stress48k", "tiny", "tiny_dp", "vocos", "tiny_vocos", "vits2_vocos_v1",
                                   "tiny_vits2_vocos", "vits2_v1", "tiny_preconv2_spk"])
def test_blob_layout_is_consistent(mname):
    cfg = config.make_config(config.MODEL_CONFIGS[mname], 100, 4)
    lay = checkpoint.blob_layout(cfg)
    names = [n for n, *_ in lay]
    assert len(set(names)) == len(names)
    end = 0
    for n, off, numel, shape in lay:
        assert off >= end and off % 64 == 0 and numel == int(np.prod(shape))
        end = off + numel
    assert checkpoint.blob_numel(cfg) >= end


def test_layout_names_and_shapes_match_reference_state_dict():
    """Golden list of the reference's own state_dict keys/shapes (dumped from the real
    SynthesizerTrn in the build container) covers every blob tensor after weight-norm folding."""
    from oracle import ref_import
    if not ref_import.available():
        pytest.skip("reference not present on this box")
    import contextlib, io
    S, *_ = ref_import.import_reference()
    for mname, nspk in [("v1", 1), ("v3", 2), ("vocos", 2), ("vits2_vocos_v1", 1),
                        ("tiny_preconv2_spk", 3)]:
        with contextlib.redirect_stdout(io.StringIO()):
            net = S(50, 513, 32, n_speakers=nspk, **config.MODEL_CONFIGS[mname])
        ref = {k: tuple(v.shape) for k, v in checkpoint.fold_weight_norm(net.state_dict()).items()}
        cfg = config.make_config(config.MODEL_CONFIGS[mname], 50, nspk)
        for n, _, _, shape in checkpoint.blob_layout(cfg):
            assert n in ref, n
            assert ref[n] == shape, (n, ref[n], shape)
        # and nothing on the infer path is left out
        ours = {n for n, *_ in checkpoint.blob_layout(cfg)}
        skipped = [k for k in ref if k not in ours and not (
            k.startswith("enc_q.") or k.startswith("dp.post_") or k.startswith("dp.flows.1.")
            or ".post_transformer." in k)]  # post_transformer: built but unused (flows.py:152-154)
        assert not skipped, skipped


def test_fold_weight_norm_matches_torch():
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(6, 8, 3))
    convt = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2))
    with torch.no_grad():
        conv.weight_g.mul_(1.3)
        convt.weight_g.mul_(0.7)
    x = torch.randn(2, 6, 11)
    ref_w, ref_wt = conv(x), convt(x)
    f = checkpoint.fold_weight_norm({**{"a." + k: v for k, v in conv.state_dict().items()},
                                     **{"b." + k: v for k, v in convt.state_dict().items()}})
    got = torch.nn.functional.conv1d(x, f["a.weight"], f["a.bias"])
    gott = torch.nn.functional.conv_transpose1d(x, f["b.weight"], f["b.bias"], stride=2)
    assert torch.allclose(got, ref_w, atol=1e-6) and torch.allclose(gott, ref_wt, atol=1e-6)


def test_pack_blob_roundtrip_and_errors():
    cfg = config.make_config(config.MODEL_CONFIGS["tiny"], 30, 2)
    sd = synth.make_state_dict(cfg, 1)
    blob = checkpoint.pack_blob(cfg, sd)
    W = checkpoint.fold_weight_norm(sd)
    for n, off, numel, shape in checkpoint.blob_layout(cfg):
        assert torch.equal(blob[off:off + numel].view(shape), W[n].float())
    bad = dict(sd)
    bad.pop("enc_p.proj.bias")
    with pytest.raises(KeyError):
        checkpoint.pack_blob(cfg, bad, strict=True)
    bad = dict(sd)
    bad["enc_p.proj.bias"] = torch.zeros(3)
    with pytest.raises(ValueError):
        checkpoint.pack_blob(cfg, bad)
    # non-strict: the reference's deterministic init values, not zeros, for what is missing
    bad = dict(sd)
    bad.pop("enc_p.encoder.norm_layers_1.0.gamma")
    lay = {n: (off, numel) for n, off, numel, _ in checkpoint.blob_layout(cfg)}
    off, numel = lay["enc_p.encoder.norm_layers_1.0.gamma"]
    assert torch.equal(checkpoint.pack_blob(cfg, bad, strict=False)[off:off + numel],
                       torch.ones(numel))


def test_load_checkpoint_is_strict(tmp_path):
    """A checkpoint that lacks tensors the config asks for (wrong n_layers / flow type / vocoder
    pairing) must be refused, not zero-filled (round-1 ADVICE)."""
    from wetts_amd import SynthesizerTrn, models
    net = SynthesizerTrn(30, 513, 32, n_speakers=2, **config.MODEL_CONFIGS["tiny"])
    sd = synth.make_state_dict(net.cfg, 1)
    sd.pop("dec.conv_post.weight")
    torch.save({"model": sd, "iteration": 7, "learning_rate": 1e-4}, tmp_path / "G_7.pth")
    with pytest.raises(KeyError):
        models.load_checkpoint(str(tmp_path / "G_7.pth"), net, None)


def test_chunk_helpers_match_the_references_own_functions():
    """tests/golden/chunk_kat.npz holds the outputs of get_chunks / depadding lifted out of the
    reference's inference_onnx.py:37-76 (make_golden.py:chunk_kat) over 184 windows: lengths around
    the block boundaries, pad > block, pad = 0, the single-window mode."""
    from wetts_amd.session import depad_bounds, get_chunks
    rows = np.load(os.path.join(util.GOLDEN, "chunk_kat.npz"))["rows"].tolist()
    assert len(rows) > 100
    for (L, block, pad, hop, n, i, a, b, lo, hi) in rows:
        wins = get_chunks(L, block, pad)
        assert len(wins) == n and wins[i] == (a, b), (L, block, pad, i)
        if block == -1:
            continue
        g = depad_bounds(n, i, block, pad, hop, (b - a) * hop)
        g = (g[0], g[1]) if g[1] > g[0] else (0, 0)
        assert g == (lo, hi), (L, block, pad, i, g, (lo, hi))


def test_triton_min_chunk_helpers_match_the_references_own_functions():
    """tests/golden/chunk_kat_triton.npz: get_chunks / depadding of the Triton streaming twin
    (stream_tts/1/model.py:58-111, MIN_CHUNK 65, reflect-padded short last window) lifted from the reference by
    AST (make_golden.py:chunk_kat_triton).  Windows, pad_end, the reflected frame indices and the kept sample
    ranges are equal wherever the reference returns a result; the two documented deviations are exactly the
    reference's crash (`-None`) and its single-window over-read."""
    from wetts_amd.session import depad_bounds_min, get_chunks_min
    rows = np.load(os.path.join(util.GOLDEN, "chunk_kat_triton.npz"))["rows"].tolist()
    seen_pad = seen_raise = seen_single = 0
    for (L, block, pad, hop, n, i, a, wlen, pad_end, cks, lo, hi, raised) in rows:
        wins, pe = get_chunks_min(L, block, pad, 65)
        assert len(wins) == n and (pe if pe is not None else -1) == pad_end, (L, block, pad)
        ws, we = wins[i]
        last_pad = pe if (pe and i == n - 1) else 0
        idx = np.arange(ws, we, dtype=np.int64)
        if last_pad:  # the reflected frames, through the same numpy call stream_decode makes
            idx = np.pad(idx.reshape(1, -1, 1), ((0, 0), (0, last_pad), (0, 0)), mode="reflect").reshape(-1)
            seen_pad += 1
        assert ws == a and idx.shape[0] == wlen, (L, block, pad, i)
        assert int(((np.arange(idx.shape[0]) + 1) * idx).sum()) == cks, (L, block, pad, i)
        g = depad_bounds_min(n, i, block, pad, hop, wlen * hop, last_pad or None)
        g = (g[0], g[1]) if g[1] > g[0] else (0, 0)
        if raised:  # reference: TypeError; here the un-padded tail is kept
            seen_raise += 1
            assert i == n - 1 and n > 1 and pad_end == -1 and g == (min(i * block, pad) * hop, wlen * hop)
        elif n == 1 and pad_end > 0:  # reference keeps audio of reflected frames; here clipped to real frames
            seen_single += 1
            assert (lo, hi) == (0, min(block, wlen) * hop) and g == (0, L * hop)
            # ... and strict_reference=True reproduces the reference client's range sample for sample
            assert depad_bounds_min(n, i, block, pad, hop, wlen * hop, last_pad or None, strict_reference=True) == (lo, hi)
        else:
            assert g == (lo, hi), (L, block, pad, i, g, (lo, hi))
    assert seen_pad > 10 and seen_raise > 0 and seen_single > 0
    # the pieces always tile [0, L*hop) exactly
    for L in (30, 65, 139, 141, 333):
        wins, pe = get_chunks_min(L, 70, 10, 65)
        total = 0
        for i, (ws, we) in enumerate(wins):
            lp = pe if (pe and i == len(wins) - 1) else None
            a, b = depad_bounds_min(len(wins), i, 70, 10, 256, (we - ws + (lp or 0)) * 256, lp)
            total += b - a
        assert total == L * 256


def test_state_dict_round_trip_and_reference_keys():
    """SynthesizerTrn.state_dict(): folded tensors under the reference's keys; loading it into a second model
    reproduces the blob bit for bit; with the live reference present, the dict loads into the reference module
    after its own remove_weight_norm() and every tensor lands (CPU tier: read from the host blob; the device
    read-back through wetts_get_blob is the GPU tier's)."""
    from wetts_amd import SynthesizerTrn
    for mname, n_spk in (("tiny", 3), ("tiny_mono_post", 2), ("tiny_vocos", 2)):
        cfg = config.make_config(config.MODEL_CONFIGS[mname], 40, n_spk)
        sd = synth.make_state_dict(cfg, 5)
        net = SynthesizerTrn(40, 513, 32, n_speakers=n_spk, **config.MODEL_CONFIGS[mname]).load_state_dict(sd)
        out = net.state_dict()
        W = checkpoint.fold_weight_norm(sd)
        assert list(out) == [n for n, *_ in checkpoint.blob_layout(cfg)]
        for k, v in out.items():
            assert tuple(v.shape) == tuple(W[k].shape) and torch.equal(v, W[k].to(torch.float32)), k
        net2 = SynthesizerTrn(40, 513, 32, n_speakers=n_spk, **config.MODEL_CONFIGS[mname]).load_state_dict(out)
        assert torch.equal(net2._blob, net._blob)
    from oracle import ref_import
    if not ref_import.available():
        return
    import contextlib
    import io
    Ref, *_ = ref_import.import_reference()
    with contextlib.redirect_stdout(io.StringIO()):
        ref = Ref(40, 513, 32, n_speakers=3, **config.MODEL_CONFIGS["tiny"]).eval()
    ref.dec.remove_weight_norm()
    ref.flow.remove_weight_norm()
    cfg = config.make_config(config.MODEL_CONFIGS["tiny"], 40, 3)
    net = SynthesizerTrn(40, 513, 32, n_speakers=3, **config.MODEL_CONFIGS["tiny"]).load_state_dict(
        synth.make_state_dict(cfg, 5))
    missing, unexpected = ref.load_state_dict(net.state_dict(), strict=False)
    assert not unexpected
    assert all(k.startswith(("enc_q.", "dp.post_", "dp.flows.1.")) for k in missing), missing


def test_config_validation_and_unsupported_options():
    with pytest.raises(NotImplementedError):
        config.make_config(dict(config.MODEL_CONFIGS["v1"], use_transformer_flows=True,
                                transformer_flow_type="fft"), 10, 1)
    sp = config.make_config(dict(config.MODEL_CONFIGS["v1"], use_spk_conditioned_encoder=True,
                                 use_transformer_flows=True, transformer_flow_type="pre_conv2"), 10, 2)
    assert sp.use_spk_conditioned_encoder == 1 and sp.transformer_flows == 2
    # without speakers the reference's enc_gin_channels is 0 (models.py:87-90): option is a no-op
    assert config.make_config(dict(config.MODEL_CONFIGS["v1"], use_spk_conditioned_encoder=True),
                              10, 0).use_spk_conditioned_encoder == 0
    # a config that omits transformer_flow_type gets the reference's default,
    # "mono_layer_post_residual" (models.py:74-75), never pre_conv
    mp = config.make_config(dict(config.MODEL_CONFIGS["v1"], use_transformer_flows=True), 10, 1)
    mi = config.make_config(dict(config.MODEL_CONFIGS["v1"], use_transformer_flows=True,
                                 transformer_flow_type="mono_layer_inter_residual"), 10, 1)
    assert (mp.transformer_flows, mi.transformer_flows) == (4, 3)
    # [RCL, Flip, Mono] x 4 (flows.py:391-425): coupling layers at flow.flows.{0,3,6,9}, mono layers at {2,5,8,11}
    names = {n for n, *_ in checkpoint.blob_layout(mp)}
    assert "flow.flows.9.enc.in_layers.3.weight" in names and "flow.flows.11.post.weight" in names
    assert "flow.flows.11.pre_transformer.attn_layers.1.conv_o.weight" in names
    assert not any(n.startswith("flow.flows.2.enc") or n.startswith("flow.flows.1.") for n in names)
    with pytest.raises(NotImplementedError):
        config.make_config(dict(config.MODEL_CONFIGS["v1"], vocoder_type="bigvgan"), 10, 1)
    vc = config.make_config(dict(config.MODEL_CONFIGS["v1"], vocoder_type="vocos"), 10, 1)
    assert (vc.vocoder_type, vc.vocos_channels, vc.istft_n_fft, vc.istft_hop_length) == (1, 512, 1024, 256)
    with pytest.raises(ValueError):  # out channels must be n_fft + 2 (decoders.py:296)
        config.make_config(dict(config.MODEL_CONFIGS["vocos"], vocos_out_channels=1000), 10, 1)
    vc.istft_win_length = 512
    assert _lib.load().wetts_blob_num_tensors(C.byref(vc)) < 0 and "win_length" in _lib.last_error()
    cfg = config.make_config(config.MODEL_CONFIGS["v1"], 10, 1)
    cfg.resblock = 3
    assert _lib.load().wetts_blob_num_tensors(C.byref(cfg)) < 0
    assert "resblock" in _lib.last_error()


def test_hifigan_cost_matches_survey_figures():
    """v1 2.402 MFLOP / 21,239 B per output sample (fp32, per-conv traffic)."""
    lib = _lib.load()
    for mname, mflop, byts in [("v1", 2.402, 21239), ("v3", 0.177, 5007), ("v2", 0.151, 5315)]:
        cfg = config.make_config(config.MODEL_CONFIGS[mname], 10, 1)
        fl, by, mfl, mby = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        assert lib.wetts_hifigan_cost(C.byref(cfg), C.byref(fl), C.byref(by), C.byref(mfl),
                                      C.byref(mby)) == 0
        hop = int(np.prod(config.MODEL_CONFIGS[mname]["upsample_rates"]))
        assert abs(fl.value / hop / 1e6 - mflop) < 0.002, fl.value / hop / 1e6
        assert abs(by.value / hop - byts) < 2, by.value / hop
        assert mfl.value < fl.value and mby.value < by.value


def test_drop_in_surface_and_loud_failure_without_gpu():
    from wetts_amd import SynthesizerTrn
    net = SynthesizerTrn(10, 513, 32, n_speakers=0, **config.MODEL_CONFIGS["tiny"]).eval()
    assert net.hop_length == 8
    for meth in ("infer", "infer_encoder", "export_forward", "export_encoder_forward",
                 "export_decoder_forward", "load_state_dict", "to", "eval"):
        assert callable(getattr(net, meth))
    with pytest.raises(_lib.WettsError):  # no silent CPU fallback
        net.infer(torch.zeros(1, 3, dtype=torch.long), torch.tensor([3]))
    with pytest.raises(NotImplementedError):
        net.forward()


def test_hparams_mirror(tmp_path):
    p = tmp_path / "c.json"
    p.write_text('{"train": {"segment_size": 8192}, "data": {"hop_length": 256, '
                 '"sampling_rate": 22050}, "model": {"resblock": "1"}}')
    h = config.get_hparams_from_file(str(p))
    assert h.train.segment_size // h.data.hop_length == 32 and "resblock" in h.model.keys()
    assert dict(**h.model) == {"resblock": "1"}


def test_cli_mirror_host_side(tmp_path):
    """wetts_amd.inference keeps the reference CLI's flags / table format (inference.py:28-64) and
    refuses to run without a HIP device instead of falling back to a CPU path."""
    from wetts_amd import inference
    (tmp_path / "phones.txt").write_text("sil 0\na 1\n\nb 2\n")
    (tmp_path / "spk.txt").write_text("baker 0\n")
    assert inference.read_table(str(tmp_path / "phones.txt")) == {"sil": 0, "a": 1, "b": 2}
    with pytest.raises(AssertionError):
        (tmp_path / "bad.txt").write_text("a 1 2\n")
        inference.read_table(str(tmp_path / "bad.txt"))
    args = inference.get_args(["--checkpoint", "G.pth", "--cfg", "c.json", "--outdir", "o",
                               "--phone_table", "p", "--test_file", "t"])
    assert (args.gpu, args.batch, args.seed) == (0, 1, None)
    with pytest.raises(SystemExit):  # a required reference flag is missing
        inference.get_args(["--cfg", "c.json"])
    if not torch.cuda.is_available():
        cfgp = tmp_path / "c.json"
        import json
        cfgp.write_text(json.dumps({"train": {"segment_size": 64}, "data": {
            "hop_length": 8, "filter_length": 1024, "sampling_rate": 16000},
            "model": config.MODEL_CONFIGS["tiny"]}))
        hps = config.get_hparams_from_file(str(cfgp))
        with pytest.raises(SystemExit, match="no CPU path"):
            inference.build_model(args, {"a": 0, "b": 1}, {"s": 0}, hps)


def test_unshard_inverts_shard_utterances():
    from wetts_amd import sharding
    lens = [5, 9, 3, 9, 1, 7, 2]
    for world in (1, 2, 3):
        shards = sharding.shard_utterances(lens, world)
        per_rank = [[f"u{i}" for i in idxs] for idxs in shards]
        assert sharding.unshard(shards, per_rank) == [f"u{i}" for i in range(len(lens))]
        assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1


def test_ctypes_config_mirror_matches_the_c_struct(tmp_path):
    """Field order, offsets and size of wetts_amd._lib.Config against wetts_config_t as the C
    compiler lays it out (include/wetts_hip.h) -- a drifted mirror would silently shift fields."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = [n for n, _ in _lib.Config._fields_]
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "wetts_hip.h"', 'int main(void) {',
            '  printf("%zu\\n", sizeof(wetts_config_t));']
    prog += [f'  printf("%zu\\n", offsetof(wetts_config_t, {n}));' for n in names]
    prog += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(prog))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == C.sizeof(_lib.Config)
    assert out[1:] == [getattr(_lib.Config, n).offset for n in names]
    # and the header declares no field the mirror lacks
    body = open(HEADER).read().split("typedef struct wetts_config {")[1].split("} wetts_config_t;")[0]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)  # comments mention other identifiers
    declared = re.findall(r"int32_t\s+(\w+)", body)
    assert declared == names


def test_split_bf16_error_estimate_backs_the_design_claim(capsys):
    """six bf16 cross products accumulated in f32 are at least as accurate as
    today's f32 MFMA chain; three stay far inside the 1e-3 gate (tools/split_bf16_error.py)."""
    import importlib.util
    path = os.path.join(os.path.dirname(HEADER), "..", "tools", "split_bf16_error.py")
    spec = importlib.util.spec_from_file_location("split_bf16_error", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    rows = {}
    for line in capsys.readouterr().out.splitlines():
        name, rest = line.split("rel rms err")
        rows[name.strip()] = float(rest.split()[0])
    assert rows["split bf16, 6 products, f32 acc"] <= rows["f32 MFMA chain (today)"]
    assert rows["truncation alone (6 products)"] < 1e-7
    assert rows["split bf16, 3 products, f32 acc"] < 1e-4


def test_export_prelude_of_the_reference_runs_on_the_drop_in_module():
    """export_onnx.py:79-83 before it traces: `net_g.flow.remove_weight_norm()` (guarded by hasattr),
    `net_g.dec.remove_weight_norm()`, `net_g.forward = net_g.export_forward`, `net_g.eval()`, then the module is CALLED.
    Weight norm is folded at load time here, so the removals are no-ops; the call must go through the instance's
    `forward` like nn.Module's does."""
    from wetts_amd import SynthesizerTrn
    net = SynthesizerTrn(50, 513, 32, n_speakers=2, **config.MODEL_CONFIGS["tiny"])
    if hasattr(net.flow, "remove_weight_norm"):
        net.flow.remove_weight_norm()
    net.dec.remove_weight_norm()
    with pytest.raises(NotImplementedError):
        net(1, 2)  # the training forward is out of scope
    seen = []
    net.forward = lambda *a: seen.append(a) or "audio"
    assert net.eval() is net and net(1, 2, 3) == "audio" and seen == [(1, 2, 3)]
    net.forward = net.export_forward  # bound method of the instance, as the reference assigns it
    assert net.forward.__self__ is net


def test_mrf_mean_quotient_is_the_ieee_division_for_every_float(tmp_path):
    """common.h: div_small_const -- the 3-operation form of xs / num_kernels (decoders.py:77) in every decoder epilogue --
    equals the IEEE division bit for bit for ALL 2^32 float inputs at the divisors it is used for (2 and 3), except the
    sign of the zero it returns for v = -0.0 (exhaustive host run of tests/native/div_small_const_check.c, ~2 s each)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native", "div_small_const_check.c")
    exe = str(tmp_path / "divchk")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-mfma", "-ffp-contract=off", src, "-o", exe, "-lm"])
    for d in ("3", "2"):
        bad, zsign = (int(v) for v in subprocess.check_output([exe, d], timeout=600).split())
        assert bad == 0 and zsign == 1, (d, bad, zsign)


def test_is_onnx_is_a_constructor_argument_not_an_ignored_kwarg():
    """models.py:50,111: `is_onnx` selects VocosGenerator's iSTFT (decoders.py:279-283); export_onnx.py:59 sets it for
    every exported graph.  It must reach the C config, be switchable, and show in state_dict() as the reference's two
    OnnxSTFT buffers; a HiFi-GAN model takes it without effect (the reference's Generator has no such argument)."""
    from oracle import vits_oracle as vo
    from wetts_amd import SynthesizerTrn
    assert config.make_config(config.MODEL_CONFIGS["vocos"], 10, 1).is_onnx == 0
    assert config.make_config(dict(config.MODEL_CONFIGS["vocos"], is_onnx=True), 10, 1).is_onnx == 1
    assert config.make_config(dict(config.MODEL_CONFIGS["v1"], is_onnx=True), 10, 1).is_onnx == 1
    net = SynthesizerTrn(40, 513, 32, n_speakers=2, **dict(config.MODEL_CONFIGS["tiny_vocos"], is_onnx=True))
    assert net.is_onnx is True and net.cfg.is_onnx == 1
    sd = synth.make_state_dict(net.cfg, 3)
    net.load_state_dict(sd)
    out = net.state_dict()
    inv = vo.onnx_stft_inverse_basis(64, 16, 64)
    assert tuple(out["dec.stft.inverse_basis"].shape) == (66, 1, 64) == tuple(out["dec.stft.forward_basis"].shape)
    assert float((out["dec.stft.inverse_basis"] - inv).abs().max()) < 1e-9
    net.set_is_onnx(False)
    assert net.cfg.is_onnx == 0 and "dec.stft.inverse_basis" not in net.state_dict()
    # a checkpoint saved from an is_onnx module carries the buffers: loading it must not trip on them
    net.load_state_dict(dict(sd, **{"dec.stft.inverse_basis": inv, "dec.stft.forward_basis": inv}))
    from oracle import ref_import
    if ref_import.available():
        Ref, *_ = ref_import.import_reference()
        import contextlib
        import io
        with contextlib.redirect_stdout(io.StringIO()):
            ref = Ref(40, 513, 32, n_speakers=2, **dict(config.MODEL_CONFIGS["tiny_vocos"], is_onnx=True)).eval()
        rsd = ref.state_dict()
        net.set_is_onnx(True)
        out = net.state_dict()
        for k in ("dec.stft.forward_basis", "dec.stft.inverse_basis"):
            assert tuple(rsd[k].shape) == tuple(out[k].shape)
            assert float((rsd[k] - out[k]).abs().max()) < 1e-8, k
