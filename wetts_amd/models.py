"""SynthesizerTrn -- drop-in for the reference's inference surface
(wetts/vits/model/models.py:14-363) running on hand-written gfx950 HIP kernels through the
C ABI in include/wetts_hip.h.  PyTorch is used for device memory, streams and nothing else:
there is no eager / CPU fallback anywhere in this module.

Same constructor arguments, same `infer` / `infer_encoder` / `export_*` signatures and return
tuples as the reference; training-only members (`forward`, `voice_conversion`, `enc_q`) are out of
scope and raise.

Extra, optional keyword arguments (not in the reference): `eps_w` / `eps_z` inject the two
standard-normal draws the reference makes with torch.randn (duration_predictors.py:257,
models.py:267) so results can be compared bit-for-noise with a CPU run.  Without them the draws
come from the library's Philox kernel (`wetts_randn`), driven by the device generator's (seed, offset) so
that torch.manual_seed() makes a run reproducible the way it does for the reference.

One host synchronisation per infer(): y_lengths and the device status word (errors the reference
raises from inside its modules: IndexError of nn.Embedding, the spline's discriminant assert) come
back in ONE D2H copy.
"""
import ctypes as C
import os

import torch

from . import _lib, checkpoint, config as _config


class _Workspace:
    """Grow-only device scratch (one per model; the kernels themselves never allocate).

    Grows with 25 % headroom: the frame count of a batch is data dependent (durations are sampled), so a
    buffer sized to the request would be re-allocated every time a batch sets a new maximum -- a multi-GB
    hipMalloc in the middle of serving (seen as 75-88 ms bench steps instead of 73 on a fresh process)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = None  # release the old block to the caching allocator first
            self.buf = torch.empty(int(nbytes) + int(nbytes) // 4, dtype=torch.uint8, device=device)
        return self.buf


class _FoldedPart:
    """`net_g.dec` / `net_g.flow` as far as the reference's export prelude touches them (export_onnx.py:79-81): weight
    norm is folded into the weights when a checkpoint is loaded (checkpoint.fold_weight_norm), so removing it is a
    no-op here."""

    def remove_weight_norm(self):
        return None


class SynthesizerTrn:
    """Inference-only synthesizer with the reference's constructor signature (models.py:19-51)."""

    def __init__(self, n_vocab, spec_channels, segment_size, inter_channels, hidden_channels,
                 filter_channels, n_heads, n_layers, kernel_size, p_dropout, resblock,
                 resblock_kernel_sizes, resblock_dilation_sizes, upsample_rates,
                 upsample_initial_channel, upsample_kernel_sizes, n_speakers=0, gin_channels=0,
                 use_sdp=True, vocoder_type="hifigan", **kwargs):
        self.n_vocab = n_vocab
        self.spec_channels = spec_channels
        self.segment_size = segment_size
        self.inter_channels = inter_channels
        self.hidden_channels = hidden_channels
        self.n_speakers = n_speakers
        self.gin_channels = gin_channels
        self.use_sdp = use_sdp
        self.upsample_rates = list(upsample_rates)
        model = dict(inter_channels=inter_channels, hidden_channels=hidden_channels,
                     filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers,
                     kernel_size=kernel_size, p_dropout=p_dropout, resblock=resblock,
                     resblock_kernel_sizes=resblock_kernel_sizes,
                     resblock_dilation_sizes=resblock_dilation_sizes,
                     upsample_rates=upsample_rates,
                     upsample_initial_channel=upsample_initial_channel,
                     upsample_kernel_sizes=upsample_kernel_sizes, gin_channels=gin_channels,
                     use_sdp=use_sdp, vocoder_type=vocoder_type, **kwargs)
        self.cfg = _config.make_config(model, n_vocab, n_speakers)
        self.vocoder_type = vocoder_type
        self.is_onnx = bool(self.cfg.is_onnx)  # models.py:50,111: the Vocos head's iSTFT (see set_is_onnx)
        self.hop_length = 1
        for u in upsample_rates:
            self.hop_length *= int(u)
        if vocoder_type == "vocos":  # samples per frame = the iSTFT hop (decoders.py:279,304)
            self.hop_length = int(self.cfg.istft_hop_length)
        self.device = torch.device("cpu")
        self._blob = None      # CPU float32 blob (weights live here until .to(device))
        self._handle = None    # wetts_model_t*
        self._ws = _Workspace()
        self._ws_dec = _Workspace()  # the decoder's own scratch in overlap mode (see set_overlap)
        self._enc_stream = None
        self.overlap = False
        self._debug_overlap = bool(os.environ.get("WETTS_DEBUG_OVERLAP"))  # check the overlap contract on every call
        self.dec = _FoldedPart()
        self.flow = _FoldedPart()
        self.quiet = True      # the reference prints stage timers on every call (:273-279)
        self.last_status = 0

    # ---- nn.Module-shaped plumbing the reference's callers use -------------------------------
    def eval(self):
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("training is out of scope of the MI355X inference path")
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device != self.device:
            self._destroy()
            self.device = device
            if self._blob is not None and device.type == "cuda":
                self._create()
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def set_decoder_dtype(self, dtype, fused=True, serial=False):
        """HiFi-GAN arithmetic: torch.float32 (default, parity-gated), torch.bfloat16 or
        torch.float16 (16-bit activations/weights, f32 accumulation; encoder / duration / flow stay f32).
        `fused=False` runs every ResBlock1 (c1, c2) pair as two conv launches instead of the fused
        LDS-resident kernel -- bit-identical, for diagnostics.  `serial=True` keeps the f32 decoder's launches on the
        caller's stream one after the other instead of running a stage's three ResBlock chains on three streams --
        bit-identical; the form in which a kernel trace's per-kernel durations do not overlap."""
        prec = {torch.float32: 0, "f32": 0, "fp32": 0, torch.bfloat16: 1, "bf16": 1,
                torch.float16: 2, "f16": 2, "fp16": 2,
                # the `export_onnx.py --quant` variant (export_onnx.py:149-157): uint8 dynamic quantisation
                torch.uint8: 3, torch.quint8: 3, "uint8": 3, "u8": 3}[dtype]
        if not fused:
            prec |= 0x10  # WETTS_DECODER_UNFUSED
        if serial:
            prec |= 0x20  # WETTS_DECODER_SERIAL: one launch after the other on the caller's stream (measurement)
        self._decoder_precision = prec
        if self._handle is not None:
            _lib.check(_lib.load().wetts_set_decoder_precision(self._handle, prec),
                       "set_decoder_precision")
        return self

    def set_is_onnx(self, on=True):
        """The `is_onnx` ctor flag of the reference (models.py:50,111) on a live model: which iSTFT a Vocos head ends in
        (decoders.py:300-304).  False: torchaudio's InverseSpectrogram == torch.istft -- what `infer()` of the PyTorch CLI
        (inference.py) computes.  True: OnnxSTFT.inverse (utils/stft.py:325-340; conv_transpose1d with
        pinv(scale * basis)^T * hann, no window-envelope division) -- what every graph export_onnx.py writes computes,
        because that script forces hps.model.is_onnx = True (export_onnx.py:59).  At hop = n_fft / 4 the interior of
        the two differs by the factor 0.375, the first / last n_fft/2 samples also in shape.  HiFi-GAN models take the
        flag without effect, like the reference.  The ORT-shaped sessions (session.py) switch to True for their calls."""
        self.is_onnx = bool(on)
        self.cfg.is_onnx = int(self.is_onnx)
        if self._handle is not None:
            _lib.check(_lib.load().wetts_set_istft_mode(self._handle, int(self.is_onnx)), "set_istft_mode")
        return self

    def set_flow_dtype(self, dtype):
        """Arithmetic of the flow's WaveNet layers: torch.float32 (default, parity-gated),
        torch.bfloat16 or torch.float16 (16-bit activations / weights, f32 accumulation, f32 skip
        sum).  BASELINE.json configs[2] precision for the flow."""
        prec = {torch.float32: 0, "f32": 0, "fp32": 0, torch.bfloat16: 1, "bf16": 1,
                torch.float16: 2, "f16": 2, "fp16": 2}[dtype]
        self._flow_precision = prec
        if self._handle is not None:
            _lib.check(_lib.load().wetts_set_flow_precision(self._handle, prec),
                       "set_flow_precision")
        return self

    def set_overlap(self, on=True):
        """Back-to-back calls as a two-stage pipeline (not in the reference).  With overlap on, the encoder stages of
        an infer() call -- text encoder, durations, flow: a few hundred small launches that cannot fill the chip --
        run on a side stream and therefore BESIDE whatever the caller's stream still holds, in a serving loop the
        previous call's decoder; the decoder waits for them by an event and runs on the caller's stream with its own
        workspace.  Results are identical; every returned tensor is safe to use on the caller's stream.  Contract: the
        input tensors of a call must already be materialised when it is made -- they are read on the side stream
        without waiting for the caller's stream (waiting would put the call behind the previous decoder again).  Device
        tensors that exist before the serving loop starts, or that come from `upload()`, satisfy it."""
        if bool(on) != self.overlap and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # the two modes share the encoder workspace differently: drain first
        self.overlap = bool(on)
        return self

    def _new_enc_stream(self):
        """The side stream of overlap mode (stream priorities were measured in round 5: not a lever)."""
        return torch.cuda.Stream(device=self.device)

    def upload(self, t, dtype=None, consumer="encoder", non_blocking=False):
        """Host tensor / array -> device tensor on the stream that will read it.  `consumer="encoder"` (ids, lengths,
        speaker ids: what infer() / infer_encoder() read): the caller's stream, or in overlap mode the encoder's side
        stream, whose work is NOT ordered behind the caller's stream (see set_overlap).  `consumer="decoder"` (a z chunk
        for export_decoder_forward / hifigan): always the caller's stream, where the decoder runs in either mode.
        `non_blocking=True` (opt-in, pinned sources only): the call returns before the copy has finished -- the copy is
        stream-ordered in front of its reader, but the CALLER must not refill the pinned buffer until that stream has
        passed it (a serving loop that reuses one staging buffer would otherwise overwrite ids still in flight); the
        default waits for the copy, like `tensor.to(device)` of a pageable source.  The wetts_amd hosts (sessions,
        batching, CLI) put their inputs on the device through this."""
        t = torch.as_tensor(t)
        nb = bool(non_blocking and t.device.type == "cpu" and t.is_pinned())
        if consumer != "encoder" or not (self.overlap and self.device.type == "cuda"):
            return t.to(device=self.device, dtype=dtype, non_blocking=nb)
        if self._enc_stream is None:
            self._enc_stream = self._new_enc_stream()
        with torch.cuda.stream(self._enc_stream):
            out = t.to(device=self.device, dtype=dtype, non_blocking=nb)
        if not nb and t.device.type == "cpu" and t.is_pinned():
            self._enc_stream.synchronize()  # (a pinned source is copied asynchronously on a non-default stream)
        return out

    def blob_layout(self):
        return checkpoint.blob_layout(self.cfg)

    def load_state_dict(self, state_dict, strict=True):
        """Accepts the reference's `G_*.pth["model"]` dict (weight-norm pairs folded here)."""
        self._blob = checkpoint.pack_blob(self.cfg, state_dict, strict=strict)
        self._destroy()
        if self.device.type == "cuda":
            self._create()
        return self

    def state_dict(self):
        """The module's tensors under the reference's `state_dict()` keys (utils/task.py:59-76 saves
        `model.state_dict()`), read back from the model's own memory.  Weight-norm pairs are stored folded, so
        `dec.ups.0.weight` stands where the reference has `weight_g` / `weight_v` -- the spelling
        `remove_weight_norm()` leaves (decoders.py:84-88) and `load_state_dict` accepts.  Tensors live on the
        module's device (views of one blob), like nn.Module.state_dict()."""
        from collections import OrderedDict
        n = checkpoint.blob_numel(self.cfg)
        if self._handle is not None:
            blob = torch.empty(n, dtype=torch.float32, device=self.device)
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().wetts_get_blob(self._handle, _lib.ptr(blob), n, _lib.current_stream_ptr()),
                           "get_blob")
        elif self._blob is not None:
            blob = self._blob
        else:
            raise _lib.WettsError("no weights loaded: call load_state_dict(...) first")
        sd = OrderedDict((name, blob[off:off + numel].view(shape))
                         for name, off, numel, shape in checkpoint.blob_layout(self.cfg))
        if self.is_onnx and self.vocoder_type == "vocos":
            # the two buffers OnnxSTFT registers (utils/stft.py:289-290); constants of the config, not of the checkpoint
            fwd, inv = _onnx_stft_bases(int(self.cfg.istft_n_fft), int(self.cfg.istft_hop_length))
            sd["dec.stft.forward_basis"] = fwd.to(blob.device)
            sd["dec.stft.inverse_basis"] = inv.to(blob.device)
        return sd

    def load_blob(self, blob):
        """Adopts an already packed float32 blob (CPU or device tensor), e.g. after a broadcast."""
        n = checkpoint.blob_numel(self.cfg)
        if blob.dtype != torch.float32 or blob.numel() != n:
            raise ValueError(f"blob must be float32[{n}]")
        self._destroy()
        if blob.device.type == "cuda":
            self.device = blob.device
            self._blob = None
            self._create(dev_blob=blob.contiguous())
        else:
            self._blob = blob.contiguous()
            if self.device.type == "cuda":
                self._create()
        return self

    def _create(self, dev_blob=None):
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.WettsError("no HIP device visible; the product path has no CPU fallback")
        with torch.cuda.device(self.device):
            if dev_blob is None:
                dev_blob = self._blob.to(self.device)
            h = C.c_void_p()
            rc = lib.wetts_create(C.byref(self.cfg), _lib.ptr(dev_blob), dev_blob.numel(),
                                  _lib.current_stream_ptr(), C.byref(h))
            _lib.check(rc, "wetts_create")
            self._handle = h
            if getattr(self, "_decoder_precision", 0):
                _lib.check(lib.wetts_set_decoder_precision(h, self._decoder_precision),
                           "set_decoder_precision")
            if getattr(self, "_flow_precision", 0):
                _lib.check(lib.wetts_set_flow_precision(h, self._flow_precision),
                           "set_flow_precision")

    def _destroy(self):
        if self._handle is not None:
            _lib.load().wetts_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _require(self):
        if self._handle is None:
            raise _lib.WettsError(
                "model is not on a HIP device with weights loaded: call load_state_dict(...) and "
                ".to('cuda') first (there is no CPU path)")
        return _lib.load()

    def _workspace(self, B, Tx, Ty):
        lib = _lib.load()
        n = lib.wetts_workspace_bytes(self._handle, B, Tx, Ty)
        if n < 0:
            raise _lib.WettsError("workspace_bytes failed")
        return self._ws.get(n, self.device), n

    # ---- out-of-scope training surface ---------------------------------------------------------
    def forward(self, *a, **k):
        raise NotImplementedError("SynthesizerTrn.forward (training, models.py:161-226) is out of "
                                  "scope; this is the inference hot path only")

    def __call__(self, *a, **k):
        # through the instance, like nn.Module: the reference's export scripts assign `net_g.forward =
        # net_g.export_forward` (export_onnx.py:82,94,127) and then call the module
        return self.forward(*a, **k)

    def voice_conversion(self, *a, **k):
        raise NotImplementedError("voice_conversion needs the posterior encoder (out of scope)")

    # ---- stages ----------------------------------------------------------------------------------
    def _ids(self, t):
        return t.to(device=self.device, dtype=torch.int64).contiguous()

    def _f32(self, t):
        return t.to(device=self.device, dtype=torch.float32).contiguous()

    def _speaker(self, sid, B):
        lib = self._require()
        gin = max(1, self.gin_channels)
        g = torch.empty(B, gin, dtype=torch.float32, device=self.device)
        if self.n_speakers > 0:
            if sid is None:
                raise ValueError("sid is required when n_speakers > 0")
            sid = self._ids(sid)
        else:
            sid = None
        _lib.check(lib.wetts_speaker_embedding(self._handle, _lib.ptr(sid), B, _lib.ptr(g),
                                               _lib.current_stream_ptr()), "speaker_embedding")
        return g if self.n_speakers > 0 else None

    def _encode(self, x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z):
        """Everything of infer() up to and including flow^-1 (== infer_encoder, models.py:282-331).
        Returns a dict of stage tensors."""
        lib = self._require()
        x = self._ids(x)
        x_lengths = self._ids(x_lengths)
        return self._encode_stages(lib, x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z)

    def _randn(self, *shape):
        """Standard-normal tensor from the library's Philox kernel (replaces torch.randn).

        The (seed, offset) pair is the device generator's own -- torch.cuda.default_generators[device] -- read
        and advanced here exactly as an ATen kernel would: torch.manual_seed(s) therefore rewinds the stream
        (also when s is the seed already in use: the reference's `manual_seed(0); a = infer(); manual_seed(0);
        b = infer()` gives a == b, and so does this), successive calls continue it, and no ATen kernel runs."""
        return self._randn_into(torch.empty(*shape, dtype=torch.float32, device=self.device))

    def _randn_into(self, out):
        """Fills a contiguous float32 device tensor from the same stream (see _randn)."""
        lib = self._require()
        gen = torch.cuda.default_generators[self.device.index]
        seed = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF
        offset = int(gen.get_offset())
        assert out.is_contiguous() and out.dtype == torch.float32
        n = out.numel()
        _lib.check(lib.wetts_randn(_lib.ptr(out), n, seed, offset, _lib.current_stream_ptr()), "randn")
        used = (n + 3) // 4  # Philox counters consumed (4 normals each)
        gen.set_offset(offset + (used + 3) // 4 * 4)  # ATen keeps the offset a multiple of 4
        return out

    # The encoder call in two halves around its one host synchronisation.  Each half is (a) a set of device buffers
    # and (b) stream-ordered launches that touch nothing else -- so a half can also be captured once into a HIP graph
    # and replayed (session.GraphedEncoder), which is why the buffers are explicit.
    def _pre_buffers(self, B, Tx, own=False):
        """Buffers of the first half: speaker vector -> text encoder -> duration predictor -> durations.  `own`: also the
        input tensors and a private workspace (a captured graph bakes every pointer in)."""
        dev = self.device
        H, I = self.hidden_channels, self.inter_channels
        f32 = dict(dtype=torch.float32, device=dev)
        pb = dict(
            B=B, Tx=Tx, g=torch.empty(B, max(1, self.gin_channels), **f32),
            x_enc=torch.empty(B, H, Tx, **f32), stats=torch.empty(B, 2 * I, Tx, **f32),
            x_mask=torch.empty(B, Tx, **f32), logw=torch.empty(B, Tx, **f32), w_ceil=torch.empty(B, Tx, **f32),
            cum=torch.empty(B, Tx, **f32), eps_w=None, sid=None,
            # [B] y_lengths + one word of WETTS_STATUS_* bits: read back together, one sync
            meta=torch.empty(B + 1, dtype=torch.int64, device=dev))
        if own:
            nws = int(_lib.load().wetts_workspace_bytes(self._handle, B, Tx, 0))
            pb.update(x=torch.zeros(B, Tx, dtype=torch.int64, device=dev),
                      x_lengths=torch.zeros(B, dtype=torch.int64, device=dev),
                      sid=torch.zeros(B, dtype=torch.int64, device=dev), eps_w=torch.zeros(B, 2, Tx, **f32),
                      ws=torch.empty(max(nws, 16), dtype=torch.uint8, device=dev), nws=nws)
        return pb

    def _launch_pre(self, lib, pb, length_scale, noise_scale_w):
        B, Tx = pb["B"], pb["Tx"]
        s = _lib.current_stream_ptr()
        status_ptr = C.c_void_p(pb["meta"].data_ptr() + 8 * B)
        _lib.check(lib.wetts_set_status_word(self._handle, status_ptr, s), "set_status_word")
        try:
            g = pb["g"] if self.n_speakers > 0 else None
            _lib.check(lib.wetts_speaker_embedding(self._handle, _lib.ptr(pb["sid"]) if self.n_speakers > 0 else None, B,
                                                   _lib.ptr(pb["g"]), s), "speaker_embedding")
            _lib.check(lib.wetts_text_encoder(self._handle, _lib.ptr(pb["x"]), _lib.ptr(pb["x_lengths"]), _lib.ptr(g), B,
                                              Tx, _lib.ptr(pb["x_enc"]), _lib.ptr(pb["stats"]), _lib.ptr(pb["x_mask"]),
                                              _lib.ptr(pb["ws"]), pb["nws"], s), "text_encoder")
            if self.use_sdp:
                _lib.check(lib.wetts_duration_sdp(self._handle, _lib.ptr(pb["x_enc"]), _lib.ptr(pb["x_mask"]), _lib.ptr(g),
                                                  _lib.ptr(pb["eps_w"]), float(noise_scale_w), B, Tx, _lib.ptr(pb["logw"]),
                                                  status_ptr, _lib.ptr(pb["ws"]), pb["nws"], s), "duration_sdp")
            else:
                _lib.check(lib.wetts_duration_dp(self._handle, _lib.ptr(pb["x_enc"]), _lib.ptr(pb["x_mask"]), _lib.ptr(g), B,
                                                 Tx, _lib.ptr(pb["logw"]), _lib.ptr(pb["ws"]), pb["nws"], s), "duration_dp")
            _lib.check(lib.wetts_durations_to_lengths(_lib.ptr(pb["logw"]), _lib.ptr(pb["x_mask"]), float(length_scale), B,
                                                      Tx, _lib.ptr(pb["w_ceil"]), _lib.ptr(pb["cum"]), _lib.ptr(pb["meta"]),
                                                      status_ptr, s), "durations_to_lengths")
        finally:
            lib.wetts_set_status_word(self._handle, None, s)

    def _read_lengths(self, pb):
        """The one host sync of infer(): y_lengths and the status word in one D2H copy (the output length is data
        dependent, commons.py:114-115); raises what the reference raises from inside its modules."""
        B = pb["B"]
        meta_host = pb["meta"].cpu()
        y_host = meta_host[:B]
        self.last_status = int(meta_host[B].item()) & 0xFFFFFFFF
        if self.last_status & _lib.STATUS_PHONE_ID_RANGE:
            raise IndexError("index out of range in self (phoneme id outside emb, encoders.py:48)")
        if self.last_status & _lib.STATUS_SPEAKER_ID_RANGE:
            raise IndexError("index out of range in self (sid outside emb_g, models.py:239)")
        if self.last_status & (_lib.STATUS_SPLINE_DOMAIN | _lib.STATUS_DURATION_NONFINITE):
            # the reference dies on `assert (discriminant >= 0).all()` (transforms.py:171)
            raise AssertionError("spline inverse: negative discriminant (transforms.py:171)")
        return y_host, (int(y_host.max().item()) if B > 0 else 0)

    def _post_buffers(self, B, Tx, Ty, own=False):
        """Buffers of the second half: length regulation -> prior sampling -> flow^-1, for Ty frames."""
        dev = self.device
        I = self.inter_channels
        f32 = dict(dtype=torch.float32, device=dev)
        qb = dict(B=B, Tx=Tx, Ty=Ty, eps_z=None,
                  f2p=torch.empty(B, Ty, dtype=torch.int32, device=dev), y_mask=torch.empty(B, Ty, **f32),
                  attn=torch.empty(B, Ty, Tx, **f32), m_p=torch.empty(B, I, Ty, **f32),
                  logs_p=torch.empty(B, I, Ty, **f32), z_p=torch.empty(B, I, Ty, **f32), z=torch.empty(B, I, Ty, **f32))
        if own:
            nws = int(_lib.load().wetts_workspace_bytes(self._handle, B, Tx, Ty))
            qb.update(eps_z=torch.zeros(B, I, Ty, **f32), ws=torch.empty(max(nws, 16), dtype=torch.uint8, device=dev),
                      nws=nws)
        return qb

    def _launch_post(self, lib, pb, qb, noise_scale):
        B, Tx, Ty = qb["B"], qb["Tx"], qb["Ty"]
        I = self.inter_channels
        s = _lib.current_stream_ptr()
        g = pb["g"] if self.n_speakers > 0 else None
        _lib.check(lib.wetts_length_regulate(self._handle, _lib.ptr(pb["stats"]), _lib.ptr(pb["cum"]),
                                             _lib.ptr(pb["x_mask"]), _lib.ptr(pb["meta"]), _lib.ptr(qb["eps_z"]), I * Ty, Ty,
                                             float(noise_scale), B, Tx, Ty, _lib.ptr(qb["f2p"]), _lib.ptr(qb["y_mask"]),
                                             _lib.ptr(qb["attn"]), _lib.ptr(qb["m_p"]), _lib.ptr(qb["logs_p"]),
                                             _lib.ptr(qb["z_p"]), s), "length_regulate")
        _lib.check(lib.wetts_flow_reverse(self._handle, _lib.ptr(qb["z_p"]), _lib.ptr(qb["y_mask"]), _lib.ptr(g), B, Ty,
                                          _lib.ptr(qb["z"]), _lib.ptr(qb["ws"]), qb["nws"], s), "flow_reverse")

    def _encode_stages(self, lib, x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z):
        B, Tx = x.shape
        I = self.inter_channels
        if self.n_speakers > 0 and sid is None:
            raise ValueError("sid is required when n_speakers > 0")
        pb = self._pre_buffers(B, Tx)
        pb["x"], pb["x_lengths"] = x, x_lengths
        if self.n_speakers > 0:
            pb["sid"] = self._ids(sid)
        pb["ws"], pb["nws"] = self._workspace(B, Tx, 0)
        if self.use_sdp:
            pb["eps_w"] = self._randn(B, 2, Tx) if eps_w is None else self._f32(eps_w)
            if tuple(pb["eps_w"].shape) != (B, 2, Tx):
                raise ValueError(f"eps_w must be [{B},2,{Tx}]")
        self._launch_pre(lib, pb, length_scale, noise_scale_w)
        y_host, Ty = self._read_lengths(pb)
        qb = self._post_buffers(B, Tx, Ty)
        qb["eps_z"] = self._randn(B, I, Ty) if eps_z is None else self._f32(eps_z)
        if tuple(qb["eps_z"].shape) != (B, I, Ty):
            raise ValueError(f"eps_z must be [{B},{I},{Ty}], got {tuple(qb['eps_z'].shape)}")
        qb["ws"], qb["nws"] = self._workspace(B, Tx, Ty)
        self._launch_post(lib, pb, qb, noise_scale)
        return self._stage_dict(pb, qb, y_host, Ty)

    def _stage_dict(self, pb, qb, y_host, Ty):
        B = pb["B"]
        return dict(g=pb["g"] if self.n_speakers > 0 else None, x_enc=pb["x_enc"], stats=pb["stats"], x_mask=pb["x_mask"],
                    logw=pb["logw"], w_ceil=pb["w_ceil"], y_lengths=pb["meta"][:B], y_lengths_host=y_host,
                    frame2phone=qb["f2p"], y_mask=qb["y_mask"], attn=qb["attn"], m_p=qb["m_p"], logs_p=qb["logs_p"],
                    z_p=qb["z_p"], z=qb["z"], B=B, Tx=pb["Tx"], Ty=Ty)

    def ragged_supported(self):
        """True when the decoder can run a batch ragged (`infer(..., ragged=True)`): float32 ResBlock1 HiFi-GAN."""
        lib = self._require()
        return bool(lib.wetts_hifigan_ragged_supported(self._handle))

    def _decode(self, z, g, y_mask, L, y_lengths=None):
        """dec((z * y_mask)[:, :, :L], g) without materialising the masked / sliced copy.  With `y_lengths`
        (int64 device tensor [B]) the batch is decoded ragged: every utterance over its own frames, as if alone."""
        lib = self._require()
        B = z.shape[0]
        if B == 0 or L == 0:  # (z * y_mask)[:, :, :0] -> empty audio, nothing to launch
            return torch.empty(B, 1, 0, dtype=torch.float32, device=self.device)
        if self.overlap:  # the next call's encoder stages may be using self._ws on the side stream by now
            nws = lib.wetts_workspace_bytes(self._handle, B, 0, L)
            if nws < 0:
                raise _lib.WettsError("workspace_bytes failed")
            ws = self._ws_dec.get(nws, self.device)
        else:
            ws, nws = self._workspace(B, 0, L)
        audio = torch.empty(B, 1, L * self.hop_length, dtype=torch.float32, device=self.device)
        if y_lengths is not None:
            _lib.check(lib.wetts_hifigan_ragged(self._handle, _lib.ptr(z), z.stride(0), z.stride(1),
                                                _lib.ptr(y_lengths), _lib.ptr(g), B, L, _lib.ptr(audio),
                                                _lib.ptr(ws), nws, _lib.current_stream_ptr()), "hifigan_ragged")
            return audio
        _lib.check(lib.wetts_hifigan(self._handle, _lib.ptr(z), z.stride(0), z.stride(1),
                                     _lib.ptr(y_mask),
                                     y_mask.stride(0) if y_mask is not None else 0,
                                     _lib.ptr(g), B, L, _lib.ptr(audio), _lib.ptr(ws), nws,
                                     _lib.current_stream_ptr()), "hifigan")
        return audio

    # ---- the reference's public inference API ----------------------------------------------------
    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.0,
              max_len=None, eps_w=None, eps_z=None, ragged=False):
        """models.py:228-280.  Returns (o [B,1,Ty*hop], attn [B,1,Ty,Tx], y_mask [B,1,Ty],
        (z, z_p, m_p, logs_p) [B,inter,Ty]); z is unmasked, as in the reference.

        `ragged=True` (not in the reference): the generator has no masks, so the reference decodes every row of a
        padded batch to the longest utterance; ragged decodes row b over its own y_lengths[b] frames -- the audio
        the reference returns when that utterance is synthesised alone (its CLI's call shape) -- and writes zeros
        behind it.  The masked stages (encoder, durations, flow) are batch independent either way."""
        st = self._encode_ordered(x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z)
        Ty = st["Ty"]
        L = Ty if max_len is None else max(0, min(Ty, int(max_len)))
        o = self._decode(st["z"], st["g"], st["y_mask"], L, st["y_lengths"] if ragged else None)
        self._last = st
        return (o, st["attn"].unsqueeze(1), st["y_mask"].unsqueeze(1),
                (st["z"], st["z_p"], st["m_p"], st["logs_p"]))

    def _encode_ordered(self, x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z):
        """_encode for the public entry points (infer, infer_encoder, export_encoder_forward, the sessions): on the
        caller's stream, or -- overlap mode -- on the side stream with the result ordered in front of whatever the
        caller's stream does next.  EVERY entry point goes through here, so in overlap mode the encoder workspace
        (self._ws) is only ever touched on the side stream: mixing infer() with infer_encoder() / the encoder session
        cannot race on it.  WETTS_DEBUG_OVERLAP=1 checks the one contract of the mode on every call (see set_overlap):
        it compares what the side stream reads of the id tensors NOW with their contents once the caller's stream has
        drained, and raises if a device op that produces them was still pending (it serialises the pipeline: debug only)."""
        if not (self.overlap and self.device.type == "cuda"):
            return self._encode(x, x_lengths, sid, float(noise_scale), float(length_scale),
                                float(noise_scale_w), eps_w, eps_z)
        main = torch.cuda.current_stream(self.device)
        if self._enc_stream is None:
            self._enc_stream = self._new_enc_stream()
        if self._debug_overlap:
            ins = [t for t in (x, x_lengths, sid, eps_w, eps_z) if isinstance(t, torch.Tensor) and t.is_cuda]
            with torch.cuda.stream(self._enc_stream):
                seen = [t.clone() for t in ins]  # what the side stream reads at this point of the caller's stream
            main.synchronize()
            self._enc_stream.synchronize()
            for a, b in zip(seen, ins):
                if not torch.equal(a, b):
                    raise RuntimeError(
                        "overlap mode: an input tensor was still being produced on the caller's stream when the call "
                        "was made; the encoder stages read inputs on a side stream WITHOUT waiting for the caller's "
                        "stream (SynthesizerTrn.set_overlap).  Materialise inputs first (upload(), or synchronise), "
                        "or call set_overlap(False).")
        with torch.cuda.stream(self._enc_stream):
            st = self._encode(x, x_lengths, sid, float(noise_scale), float(length_scale),
                              float(noise_scale_w), eps_w, eps_z)
            done = torch.cuda.Event()
            done.record(self._enc_stream)
        main.wait_event(done)
        for v in st.values():  # allocated on the side stream, consumed (and later freed) on the caller's
            if isinstance(v, torch.Tensor) and v.is_cuda:
                v.record_stream(main)
        return st

    def infer_encoder(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1,
                      noise_scale_w=1.0, eps_w=None, eps_z=None):
        """models.py:282-331.  Returns (attn, y_mask, (z*y_mask, z_p, m_p, logs_p), g)."""
        st = self._encode_ordered(x, x_lengths, sid, noise_scale, length_scale, noise_scale_w, eps_w, eps_z)
        z = torch.empty_like(st["z"])
        _lib.check(_lib.load().wetts_mask_rows(_lib.ptr(st["z"]), _lib.ptr(st["y_mask"]), st["B"],
                                               self.inter_channels, st["Ty"], _lib.ptr(z),
                                               _lib.current_stream_ptr()), "mask_rows")
        g = st["g"].unsqueeze(-1) if st["g"] is not None else None
        return (st["attn"].unsqueeze(1), st["y_mask"].unsqueeze(1),
                (z, st["z_p"], st["m_p"], st["logs_p"]), g)

    def export_forward(self, x, x_lengths, scales, sid):
        """models.py:333-344: row 0 of `scales` [B,3] = (noise_scale, length_scale, noise_scale_w)
        applies to the whole batch."""
        sc = torch.as_tensor(scales).detach().cpu().to(torch.float32)
        audio, *_ = self.infer(x, x_lengths, sid, noise_scale=float(sc[0][0]),
                               length_scale=float(sc[0][1]), noise_scale_w=float(sc[0][2]))
        return audio

    def export_encoder_forward(self, x, x_lengths, scales, sid):
        """models.py:346-358: returns z [B, L, inter] (time-major so callers slice chunks)."""
        sc = torch.as_tensor(scales).detach().cpu().to(torch.float32)
        _, _, (z, _, _, _), _ = self.infer_encoder(x, x_lengths, sid, noise_scale=float(sc[0][0]),
                                                   length_scale=float(sc[0][1]),
                                                   noise_scale_w=float(sc[0][2]))
        return z.transpose(1, 2)

    def export_decoder_forward(self, z, sid):
        """models.py:360-363: z [B, L, inter] -> audio [B,1,L*hop]."""
        z = self._f32(z).transpose(1, 2).contiguous()
        g = self._speaker(sid, z.shape[0])
        return self._decode(z, g, None, z.shape[2])

    def hifigan(self, z, g=None):
        """Generator.forward(z, g) (decoders.py:63-82) on [B,inter,L]; g is [B,gin] or [B,gin,1]."""
        z = self._f32(z)
        if g is not None:
            g = self._f32(g).reshape(z.shape[0], -1)
        return self._decode(z, g, None, z.shape[2])

    def audio_to_int16(self, audio, lengths_samples=None):
        """inference.py:100-110 scaling on the device; audio [B,1,L] or [B,L] -> int16 [B,L]."""
        lib = self._require()
        a = self._f32(audio)
        if a.dim() == 3:
            a = a[:, 0].contiguous()
        B, L = a.shape
        pcm = torch.empty(B, L, dtype=torch.int16, device=self.device)
        ln = self._ids(lengths_samples) if lengths_samples is not None else None
        _lib.check(lib.wetts_audio_to_int16(_lib.ptr(a), _lib.ptr(ln), B, L, _lib.ptr(pcm),
                                            _lib.current_stream_ptr()), "audio_to_int16")
        return pcm


def _onnx_stft_bases(n_fft, hop):
    """OnnxSTFT's `forward_basis` / `inverse_basis` buffers [n_fft + 2, 1, n_fft] (utils/stft.py:266-290, hann window,
    win_length == n_fft) for state_dict().  The device kernels build their own copy of the inverse one in closed form
    (kernels.hip: istft_basis_kernel); this host restatement serves the state_dict surface only."""
    import numpy as np
    n = np.arange(n_fft)
    k = np.arange(n_fft // 2 + 1)
    ang = 2.0 * np.pi * ((k[:, None] * n[None, :]) % n_fft) / n_fft
    basis = np.vstack([np.cos(ang), -np.sin(ang)])  # real and imaginary rows of fft(eye)[:cutoff]
    win = torch.from_numpy(0.5 - 0.5 * np.cos(2.0 * np.pi * n / n_fft)).float()  # scipy get_window("hann", fftbins=True)
    fwd = torch.from_numpy(basis[:, None, :]).float() * win
    inv = torch.from_numpy(np.linalg.pinv((n_fft / hop) * basis).T[:, None, :]).float() * win
    return fwd, inv


def load_checkpoint(checkpoint_path, model, optimizer=None):
    """task.load_checkpoint (utils/task.py:31-56) for the drop-in model: reads `G_*.pth`, folds
    weight norm, uploads.  Returns (model, optimizer, learning_rate, iteration) like the
    reference."""
    sd, iteration, lr = checkpoint.load_state_dict_file(checkpoint_path)
    # strict: a tensor the config asks for and the checkpoint lacks means the two do not belong
    # together (wrong n_layers / flow type / vocoder); the reference would keep a random init value
    # for it (task.py:44-49) and synthesise noise, here the blob slot would stay zero -- refuse.
    model.load_state_dict(sd, strict=True)
    return model, optimizer, lr, iteration
