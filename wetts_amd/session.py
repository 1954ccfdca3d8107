"""ONNX-Runtime-shaped sessions over the HIP path.

Every Python caller of the reference's exported models goes through
`session.run(output_names, feeds)` with the tensor names fixed by export_onnx.py:95-189:
    full model : input int64[B,T], input_lengths int64[B], scales f32[B,3], sid int64[B]
                 -> output f32[B,1,T_audio]                       (wetts/cli/model.py:48-58)
    encoder    : same feeds -> z f32[B,L,192]                     (inference_onnx.py:139-145)
    decoder    : z f32[B,L,192], sid int64[B] -> output f32[B,1,L*hop]   (inference_onnx.py:146-151)
These shims accept / return numpy arrays exactly like ort.InferenceSession so wetts/cli/model.py,
inference_onnx.py or the Triton python backend can swap the session object and nothing else.

Arithmetic: a session stands for a GRAPH export_onnx.py wrote, and that script builds its model with
`hps['model']['is_onnx'] = True` (export_onnx.py:59).  For a Vocos model that flag swaps the decoder's last step
(decoders.py:300-304): OnnxSTFT.inverse (utils/stft.py:325-340 -- conv_transpose1d with pinv(scale * basis)^T * hann,
no window-envelope division: 0.375 x the torch.istft audio in the interior at hop = n_fft / 4, different first / last
n_fft/2 samples) instead of torchaudio's InverseSpectrogram.  Sessions that run a decoder therefore compute the
OnnxSTFT head for the duration of their calls, whatever the wrapped module's own `is_onnx` says (`is_onnx=False` at
construction keeps the module's arithmetic -- the graph a patched export script without line 59 would trace).
HiFi-GAN models have one arithmetic; the flag does nothing there.
"""
import contextlib

import numpy as np
import torch

from .models import SynthesizerTrn


class _Arg:
    def __init__(self, name, shape, type_):
        self.name, self.shape, self.type = name, shape, type_


class _SessionBase:
    def __init__(self, model: SynthesizerTrn, is_onnx=True):
        if model._handle is None:
            raise RuntimeError("model must have weights loaded and live on a HIP device")
        self.model = model
        self.is_onnx = bool(is_onnx)

    @contextlib.contextmanager
    def _graph_arithmetic(self):
        """The exported graph's iSTFT head for the calls made inside (module docstring).  The switch is a host flag of
        the handle read at launch time, so it covers exactly the launches issued in the block (the reference's sessions
        are used from one thread at a time, like the module)."""
        m = self.model
        if not self.is_onnx or m.vocoder_type != "vocos" or m.is_onnx:
            yield
            return
        m.set_is_onnx(True)
        try:
            yield
        finally:
            m.set_is_onnx(False)

    def _dev(self, a, dtype, consumer="encoder"):
        return self.model.upload(np.asarray(a), dtype, consumer=consumer)

    def get_providers(self):
        return ["WettsHIPExecutionProvider"]

    def _check(self, output_names):
        names = [o.name for o in self.get_outputs()]
        if output_names is not None:
            for n in output_names:
                if n not in names:
                    raise ValueError(f"unknown output {n!r}; available: {names}")


class InferenceSession(_SessionBase):
    """The non-streaming model (export_forward, models.py:333-344).

    `max_pad_frac=None` (default) is the reference's call: the whole feed is ONE padded batch.  With a float, a
    B > 1 feed of ragged `input_lengths` is decoded as the length-sorted padded sub-batches `batching.plan`
    chooses (padding share <= max_pad_frac) and the rows are put back in feed order: `output` keeps its shape
    [B,1,T_audio]; row b holds its utterance's valid audio (what infer() returns for it inside its sub-batch)
    followed by zeros instead of the decoded padding tail.  `max_batch` caps a sub-batch (the Triton
    `generator` model's max_batch_size is 32, generator/config.pbtxt)."""

    def __init__(self, model: SynthesizerTrn, max_pad_frac=None, max_batch=0, ragged=False, is_onnx=True):
        super().__init__(model, is_onnx)
        self.max_pad_frac = max_pad_frac
        self.max_batch = max_batch
        self.ragged = ragged  # batching.synthesize(ragged=): rows decoded as the reference decodes them alone
        self.last_plan_stats = None

    def get_inputs(self):
        return [_Arg("input", ["B", "T"], "tensor(int64)"),
                _Arg("input_lengths", ["B"], "tensor(int64)"),
                _Arg("scales", ["B", 3], "tensor(float)"), _Arg("sid", ["B"], "tensor(int64)")]

    def get_outputs(self):
        return [_Arg("output", ["B", 1, "L"], "tensor(float)")]

    def run(self, output_names, feeds, run_options=None):
        self._check(output_names)
        with self._graph_arithmetic():
            return self._run(feeds)

    def _run(self, feeds):
        scales = np.asarray(feeds["scales"], dtype=np.float32)
        ids = np.asarray(feeds["input"])
        lens = np.asarray(feeds["input_lengths"]).reshape(-1)  # Triton feeds [B,1] (tts/1/model.py:128)
        sid = feeds.get("sid")  # the Triton `generator` model omits it (config.pbtxt:21-46)
        B = ids.shape[0]
        if self.max_pad_frac is not None and B > 1 and len(set(int(v) for v in lens)) > 1:
            return [self._run_bucketed(ids, lens, scales, sid)]
        x = self._dev(ids, torch.int64)
        xl = self._dev(lens, torch.int64)
        sid = self._dev(sid, torch.int64) if sid is not None else \
            torch.zeros(x.shape[0], dtype=torch.int64, device=self.model.device)
        audio = self.model.export_forward(x, xl, torch.from_numpy(scales), sid)
        return [audio.cpu().numpy()]

    def _run_bucketed(self, ids, lens, scales, sid):
        from . import batching
        m = self.model
        B = ids.shape[0]
        seqs = [ids[b, :int(lens[b])] for b in range(B)]
        sids = np.zeros(B, np.int64) if sid is None else np.asarray(sid).reshape(-1)
        # row 0 of `scales` applies to the whole batch, as in export_forward (models.py:333-344)
        outs, st = batching.synthesize(m, seqs, sids, noise_scale=float(scales[0][0]),
                                       length_scale=float(scales[0][1]), noise_scale_w=float(scales[0][2]),
                                       max_pad_frac=self.max_pad_frac, max_batch=self.max_batch,
                                       return_stats=True, ragged=self.ragged)
        self.last_plan_stats = st
        T = max(int(o.numel()) for o in outs)
        audio = torch.zeros(B, 1, T, dtype=torch.float32, device=m.device)
        for b, o in enumerate(outs):
            audio[b, 0, :o.numel()] = o
        return audio.cpu().numpy()


class EncoderSession(_SessionBase):
    """Streaming front half (export_encoder_forward, models.py:346-358): -> z [B, L, inter].
    `use_graph=True` replays captured HIP graphs for repeating shapes (see GraphedEncoder)."""

    def __init__(self, model, use_graph=False, frame_bucket=32, phoneme_bucket=16, max_shapes=8, max_buckets=8):
        super().__init__(model)
        self._graphed = GraphedEncoder(model, frame_bucket, phoneme_bucket, max_shapes, max_buckets) if use_graph else None

    def get_inputs(self):
        return InferenceSession.get_inputs(self)

    def get_outputs(self):
        return [_Arg("z", ["B", "L", self.model.inter_channels], "tensor(float)")]

    def run(self, output_names, feeds, run_options=None):
        self._check(output_names)
        # (the graphed path copies its inputs into graph-owned buffers on the CALLER's stream, so they are uploaded there
        # too; the plain path reads them on the encoder's side stream in overlap mode)
        where = "decoder" if self._graphed is not None else "encoder"
        x = self._dev(feeds["input"], torch.int64, consumer=where)
        xl = self._dev(feeds["input_lengths"], torch.int64, consumer=where)
        scales = np.asarray(feeds["scales"], dtype=np.float32)
        sid = self._dev(feeds["sid"], torch.int64, consumer=where)
        if self._graphed is not None:
            # export_encoder_forward (models.py:346-358): row 0 of `scales`, z * y_mask, time-major
            st = self._graphed.encode(x, xl, sid, float(scales[0][0]), float(scales[0][1]), float(scales[0][2]))
            z = (st["z"] * st["y_mask"].unsqueeze(1)).transpose(1, 2)
            return [z.contiguous().cpu().numpy()]
        z = self.model.export_encoder_forward(x, xl, torch.from_numpy(scales), sid)
        return [z.contiguous().cpu().numpy()]


class GraphedEncoder:
    """hipGraph replay of the encoder call (infer_encoder, models.py:282-331) for repeating shapes.

    The B = 1 encoder call is ~140 launches of ~10 us each with the host issuing them one by one; a kernel trace
    shows 15-20 % of its span idle between launches (profiles/r04_b1_anatomy.txt).  Its two halves -- up to the
    durations, and from length regulation through the flow -- sit on either side of the call's one host
    synchronisation (the frame count is data dependent), so each is captured into its own graph:
      * the first half per (B, Tx): speaker vector, text encoder, duration predictor, durations -> lengths;
      * the second half per (B, Tx, frame bucket): the frame count is rounded up to a multiple of `frame_bucket` and
        the extra frames are frames with mask 0 -- what a shorter utterance's tail is in a padded batch -- so that
        utterances of similar length replay the same graph.  Results are returned as views of the first Ty frames.
    Same kernels as the plain path; the flow's tile shapes follow the bucketed length, so z agrees with the plain call
    to round-off (1e-6), not bit for bit.  The two standard-normal draws are made OUTSIDE the graphs (the Philox
    (seed, offset) pair is a kernel argument) into the graphs' input buffers, from the same stream positions as the
    plain path: a seed gives the same audio either way.  Buffers, workspaces included, are owned per graph entry
    because a graph bakes the pointers in.

    The caches are BOUNDED.  A serving process sees a new phoneme count with almost every request, and an entry is a
    captured graph plus private input / output buffers plus a whole private workspace: unbounded, device memory grows
    with every distinct shape and most requests pay a capture instead of a replay.  So (a) the phoneme count is bucketed
    like the frame count: ids are padded with zeros to a multiple of `phoneme_bucket` and `x_lengths` keeps the true
    length -- exactly a shorter utterance in a padded batch; the Tx-shaped results are returned as views of the first
    Tx columns -- and (b) both caches are LRU: at most `max_shapes` first-half entries, each holding at most
    `max_buckets` second-half graphs (a second-half graph reads its first half's buffers, so it lives and dies inside
    that entry).  An evicted entry drops its graph, buffers and workspace."""

    def __init__(self, model: SynthesizerTrn, frame_bucket=32, phoneme_bucket=16, max_shapes=8, max_buckets=8):
        import collections
        if model._handle is None:
            raise RuntimeError("model must have weights loaded and live on a HIP device")
        self.model = model
        self.frame_bucket = int(frame_bucket)
        self.phoneme_bucket = max(1, int(phoneme_bucket))
        self.max_shapes, self.max_buckets = max(1, int(max_shapes)), max(1, int(max_buckets))
        self._pre = collections.OrderedDict()  # key -> {"pb", "graph", "post": OrderedDict(key2 -> {"qb", "graph"})}
        self.captures = 0  # graphs captured so far (a replay does not count): what a cache-efficiency test reads

    @staticmethod
    def _lru_get(cache, key):
        e = cache.get(key)
        if e is not None:
            cache.move_to_end(key)
        return e

    @staticmethod
    def _lru_put(cache, key, e, cap):
        cache[key] = e
        while len(cache) > cap:
            cache.popitem(last=False)  # least recently used: its graph, buffers and workspace go with it
        return e

    def _capture(self, launch):
        launch()  # un-captured first: one-time initialisation must not land in the capture
        torch.cuda.synchronize(self.model.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            launch()
        self.captures += 1
        return graph

    def encode(self, x, x_lengths, sid, noise_scale, length_scale, noise_scale_w):
        """-> the stage dict of SynthesizerTrn._encode (tensors are views of graph-owned buffers: consume or copy them
        before the next call with the same shapes)."""
        from . import _lib
        m = self.model
        lib = _lib.load()
        import collections
        B, Tx_in = x.shape
        I = m.inter_channels
        pbk = self.phoneme_bucket
        Tx = max(pbk, -(-int(Tx_in) // pbk) * pbk)  # padded phoneme count: positions >= x_lengths[b] are masked
        key = (int(B), int(Tx), float(length_scale), float(noise_scale_w))
        e = self._lru_get(self._pre, key)
        if e is None:
            pb = m._pre_buffers(B, Tx, own=True)
            e = {"pb": pb, "post": collections.OrderedDict()}
            e["graph"] = self._capture(lambda: m._launch_pre(lib, pb, length_scale, noise_scale_w))
            self._lru_put(self._pre, key, e, self.max_shapes)
        pb = e["pb"]
        if Tx == Tx_in:
            pb["x"].copy_(x)
        else:
            pb["x"].zero_()
            pb["x"][:, :Tx_in].copy_(x)
        pb["x_lengths"].copy_(x_lengths)
        if m.n_speakers > 0:
            pb["sid"].copy_(sid)
        if m.use_sdp:
            # randn(B, 2, Tx) of duration_predictors.py:257 is drawn contiguously at the TRUE phoneme count, as the
            # plain path draws it (same stream positions), and placed in the first columns of the bucketed buffer
            if Tx == Tx_in:
                m._randn_into(pb["eps_w"])
            else:
                pb["eps_w"][:, :, :Tx_in].copy_(m._randn(B, 2, Tx_in))
        e["graph"].replay()
        y_host, Ty = m._read_lengths(pb)
        fb = self.frame_bucket
        Tyb = max(fb, -(-Ty // fb) * fb)
        key2 = (Tyb, float(noise_scale))  # (the second graph reads the first one's buffers: it lives inside that entry)
        e2 = self._lru_get(e["post"], key2)
        if e2 is None:
            qb = m._post_buffers(B, Tx, Tyb, own=True)
            e2 = {"qb": qb}
            e2["graph"] = self._capture(lambda: m._launch_post(lib, pb, qb, noise_scale))
            self._lru_put(e["post"], key2, e2, self.max_buckets)
        qb = e2["qb"]
        # randn_like(m_p) of models.py:267 is [B, I, Ty]: drawn contiguously, as the plain path draws it, and placed
        # in the first Ty frames of the bucketed buffer (the rest is multiplied by mask 0)
        if Tyb == Ty:
            m._randn_into(qb["eps_z"])
        else:
            qb["eps_z"][:, :, :Ty].copy_(m._randn(B, I, Ty))
        e2["graph"].replay()
        view = {k: (qb[k][:, :Ty] if k in ("f2p", "y_mask", "attn") else qb[k][:, :, :Ty])
                for k in ("f2p", "y_mask", "attn", "m_p", "logs_p", "z_p", "z")}
        view["attn"] = view["attn"][:, :, :Tx_in]
        st = m._stage_dict(pb, dict(view, B=B, Tx=Tx, Ty=Ty), y_host, Ty)
        if Tx != Tx_in:  # the Tx-shaped stage tensors: views of the true phoneme count
            for k in ("x_enc", "stats"):
                st[k] = st[k][:, :, :Tx_in]
            for k in ("x_mask", "logw", "w_ceil"):
                st[k] = st[k][:, :Tx_in]
            st["Tx"] = int(Tx_in)
        return st


class GraphedDecoder:
    """hipGraph replay of the generator for fixed window shapes.

    The streaming decode loop (vits_model.cc:128-153, inference_onnx.py:37-76) runs the decoder on
    50-60 frame windows: ~85 kernel launches of a few microseconds each, i.e. launch-bound.  The
    window shapes repeat (first / middle / last chunk), so each (B, L) is captured once into a HIP
    graph -- same kernels, same order, bit-identical output -- and replayed with one launch.
    Buffers the graph reads and writes are owned here (its z window, speaker vector, workspace
    and audio), because a graph bakes the pointers in."""

    def __init__(self, model: SynthesizerTrn):
        if model._handle is None:
            raise RuntimeError("model must have weights loaded and live on a HIP device")
        self.model = model
        self._entries = {}

    def _build(self, B, L, has_g):
        from . import _lib
        m = self.model
        lib = _lib.load()
        dev = m.device
        e = {"z": torch.zeros(B, m.inter_channels, L, dtype=torch.float32, device=dev),
             "g": torch.zeros(B, max(1, m.gin_channels), dtype=torch.float32, device=dev) if has_g else None,
             "audio": torch.empty(B, 1, L * m.hop_length, dtype=torch.float32, device=dev)}
        nws = int(lib.wetts_workspace_bytes(m._handle, B, 0, L))
        e["ws"] = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)

        def launch():
            _lib.check(lib.wetts_hifigan(m._handle, _lib.ptr(e["z"]), e["z"].stride(0),
                                         e["z"].stride(1), None, 0, _lib.ptr(e["g"]), B, L,
                                         _lib.ptr(e["audio"]), _lib.ptr(e["ws"]), nws,
                                         _lib.current_stream_ptr()), "hifigan")

        launch()  # un-captured first: one-time initialisation must not land in the capture
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            launch()
        e["graph"] = graph
        return e

    def __call__(self, z, g=None):
        """z [B, inter, L] (any strides), g [B, gin] or None -> audio [B, 1, L*hop] (a view of the
        graph's output buffer: consume or copy it before the next call with the same shape)."""
        B, _, L = z.shape
        # (a graph bakes in which iSTFT basis a Vocos head multiplies by: one entry per arithmetic)
        key = (int(B), int(L), g is not None, bool(self.model.is_onnx))
        e = self._entries.get(key)
        if e is None:
            e = self._entries[key] = self._build(int(B), int(L), g is not None)
        e["z"].copy_(z)
        if g is not None:
            e["g"].copy_(g.reshape(B, -1))
        e["graph"].replay()
        return e["audio"]


class DecoderSession(_SessionBase):
    """Streaming back half (export_decoder_forward, models.py:360-363): z chunk -> audio.
    `use_graph=True` replays a captured HIP graph per window shape (see GraphedDecoder)."""

    def __init__(self, model, use_graph=False, is_onnx=True):
        super().__init__(model, is_onnx)
        self._graphed = GraphedDecoder(model) if use_graph else None

    def get_inputs(self):
        return [_Arg("z", ["B", "L", self.model.inter_channels], "tensor(float)"),
                _Arg("sid", ["B"], "tensor(int64)")]

    def get_outputs(self):
        return [_Arg("output", ["B", 1, "L"], "tensor(float)")]

    def run(self, output_names, feeds, run_options=None):
        self._check(output_names)
        with self._graph_arithmetic():
            return self._run(feeds)

    def _run(self, feeds):
        # (read by the decoder, which runs on the caller's stream in either mode -- not the encoder's side stream)
        z = self._dev(feeds["z"], torch.float32, consumer="decoder")
        sid = self._dev(feeds["sid"], torch.int64, consumer="decoder")
        if self._graphed is not None and z.shape[1] > 0:
            g = self.model._speaker(sid, z.shape[0])
            return [self._graphed(z.transpose(1, 2), g).cpu().numpy()]
        return [self.model.export_decoder_forward(z, sid).cpu().numpy()]


# ---- chunked streaming helpers -----------------------------------------------------------------
# Same protocol as the reference's streaming clients (inference_onnx.py:37-76,
# runtime/core/model/vits_model.cc:96-126): ceil(L/block) windows, each widened by `pad` frames
# on both sides (clipped to [0, L]); after decoding, the samples that came from the padding are
# discarded so the pieces concatenate to exactly L*hop samples.
def get_chunks(mel_len, block_size, pad_size):
    """[(window_start, window_end)] in frames; block_size == -1 => one window."""
    if block_size == -1:
        return [(0, mel_len)]
    n = -(-mel_len // block_size)
    return [(max(0, i * block_size - pad_size), min((i + 1) * block_size + pad_size, mel_len))
            for i in range(n)]


def depad_bounds(chunk_num, chunk_id, block, pad, upsample, n_samples):
    """Sample range of a decoded window to keep (depadding, inference_onnx.py:60-76).  Mirrored to the sample, including
    what the reference does with block == -1 (one window): `audio[:, :block * upsample]` is then `audio[:, :-upsample]`,
    i.e. its streaming client drops the last frame's samples in that mode (tests/test_cpu_properties.py)."""
    front = min(chunk_id * block, pad)
    if chunk_id == 0:
        return 0, min(n_samples, block * upsample)
    if chunk_id == chunk_num - 1:
        return front * upsample, n_samples
    return front * upsample, (front + block) * upsample


# The Triton streaming twin (runtime/cpu_triton_stream/model_repo/stream_tts/1/model.py:11-14,58-111) adds a minimum
# window: MIN_CHUNK = 65 frames, VOC_BLOCK_SIZE = 70, VOC_PAD_SIZE = 10.  A last window shorter than MIN_CHUNK is
# reflect-padded at its end (np.pad mode="reflect" along the frame axis, :82-84) and the audio decoded from those
# `pad_end` frames is cut off again (:103-106).
TRITON_MIN_CHUNK, TRITON_BLOCK_SIZE, TRITON_PAD_SIZE = 65, 70, 10


def get_chunks_min(mel_len, block_size, pad_size, min_chunk=TRITON_MIN_CHUNK):
    """stream_tts/1/model.py:58-86: ([(window_start, window_end)], pad_end) -- the windows of get_chunks plus the
    number of reflected frames the LAST window is extended by (None when it already has min_chunk frames).
    block_size == -1 => the reference returns the bare list `[mel]`; here ([(0, L)], None)."""
    if block_size == -1:
        return [(0, mel_len)], None
    wins = [(max(0, i - pad_size), min(i + block_size + pad_size, mel_len)) for i in range(0, mel_len, block_size)]
    pad_end = None
    if wins and wins[-1][1] - wins[-1][0] < min_chunk:
        pad_end = min_chunk - (wins[-1][1] - wins[-1][0])
    return wins, pad_end


def depad_bounds_min(chunk_num, chunk_id, block, pad, upsample, n_samples, pad_end, strict_reference=False):
    """Sample range to keep of a window decoded under the min-chunk protocol (depadding,
    stream_tts/1/model.py:89-111): as depad_bounds, except that the last window drops `pad_end * upsample`
    samples at its end.  Two behaviours of the reference are NOT reproduced, because they are crashes, not
    results: a last window that needed no padding (pad_end None) with chunk_num > 1 raises TypeError there
    (`-pad_end * upsample`, :105) -- here the window's tail is kept, as the two other streaming clients do; a
    single window (chunk_id 0 is also the last) keeps `block * upsample` samples there, which includes audio
    decoded from reflected frames when L < block -- here it is clipped to the samples of real frames, unless
    `strict_reference` asks for the reference client's stream sample for sample (min(block, MIN_CHUNK) * upsample
    samples for a single short window: the tail is the vocoder's rendering of mirrored frames, not speech)."""
    front = min(chunk_id * block, pad)
    real = n_samples - (pad_end or 0) * upsample  # samples decoded from real (not reflected) frames
    if chunk_id == 0:
        return 0, min(n_samples if (strict_reference and chunk_num == 1) else real, block * upsample)
    if chunk_id == chunk_num - 1:
        return front * upsample, real
    return front * upsample, (front + block) * upsample


def stream_decode(decoder, z, sid, chunk_size=40, pad_size=10, min_chunk=None, strict_reference=False):
    """Decodes z [1,L,C] window by window with overlap-discard; yields float32 audio pieces
    (numpy) whose concatenation has L*hop samples.  `decoder` is a DecoderSession.

    `min_chunk` (e.g. TRITON_MIN_CHUNK with chunk_size=TRITON_BLOCK_SIZE, pad_size=TRITON_PAD_SIZE) selects the
    Triton twin's protocol: a short last window is reflect-padded to min_chunk frames before it is decoded.
    `strict_reference` (min-chunk protocol only): an utterance shorter than min_chunk yields the reference client's
    min(chunk_size, min_chunk) * hop samples -- its audio of the mirrored frames included -- instead of L * hop."""
    hop = decoder.model.hop_length
    if min_chunk is None:
        wins, pad_end = get_chunks(z.shape[1], chunk_size, pad_size), None
    else:
        wins, pad_end = get_chunks_min(z.shape[1], chunk_size, pad_size, min_chunk)
    for i, (ws, we) in enumerate(wins):
        zw = z[:, ws:we]
        last_pad = pad_end if (pad_end and i == len(wins) - 1) else None
        if last_pad:  # the same numpy call as the reference, on the frame axis of [1, L, C]
            zw = np.pad(np.asarray(zw), ((0, 0), (0, last_pad), (0, 0)), mode="reflect")
        out = decoder.run(None, {"z": zw, "sid": sid})[0].reshape(1, -1)
        if min_chunk is None:
            a, b = depad_bounds(len(wins), i, chunk_size, pad_size, hop, out.shape[1])
        else:
            a, b = depad_bounds_min(len(wins), i, chunk_size, pad_size, hop, out.shape[1], last_pad, strict_reference)
        yield out[0, a:b]
