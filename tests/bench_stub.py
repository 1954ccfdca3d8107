"""TEST INFRASTRUCTURE: a device-free stand-in for bench.HipBackend, so that bench.py's whole N > 1 control
flow (spawn -> process group -> broadcast + verification -> plan -> warm-up -> timed loop -> reductions ->
JSON line) runs at world size 2 on the CPU with gloo.  Selected by WETTS_BENCH_TEST_BACKEND=tests.bench_stub:StubBackend;
the JSON line then carries `backend_label` so it can never pass for a measurement."""
import math
import os
import time

import torch


class StubNet:
    def __init__(self, mname, n_vocab, n_speakers):
        from wetts_amd import config
        self.cfg = config.make_config(config.MODEL_CONFIGS[mname], n_vocab, n_speakers)
        self.hop_length = 1
        for i in range(self.cfg.n_upsamples):
            self.hop_length *= self.cfg.upsample_rates[i]
        self.calls = 0
        self.mrf_ms = 0.0
        self.mrf_launches = 0
        self.mrf_bytes = 0.0
        self.timing = False
        self._last = None

    def _cost(self):
        import ctypes as C
        from wetts_amd import _lib
        fl, by, mfl, mby = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        assert _lib.load().wetts_hifigan_cost(C.byref(self.cfg), C.byref(fl), C.byref(by), C.byref(mfl), C.byref(mby)) == 0
        return mfl.value, mby.value

    def load_blob(self, blob):
        from wetts_amd import checkpoint
        me = os.environ.get("RANK", "0")
        if os.environ.get("WETTS_STUB_HANG_RANK") == me and os.environ.get("WETTS_STUB_HANG_AT") == "load":
            time.sleep(3600)  # a rank whose GPU hangs after the rendezvous
        if os.environ.get("WETTS_STUB_CRASH_RANK") == me:
            raise RuntimeError("stub: this rank dies while loading its weights")
        assert blob.dtype == torch.float32 and blob.numel() == checkpoint.blob_numel(self.cfg)
        self.blob_sum = float(blob.double().sum())
        return self

    def set_decoder_dtype(self, d):
        self.ddtype = d
        return self

    def set_overlap(self, on):  # (bench.py's `roofline.isolated` pass switches it off and on again)
        return self

    def set_flow_dtype(self, d):
        return self

    def infer(self, x, x_lengths, sid=None, noise_scale=1, length_scale=1, noise_scale_w=1.0, ragged=False):
        assert x.dim() == 2 and x_lengths.shape == (x.shape[0],) and sid.shape == (x.shape[0],)
        assert int(x_lengths.max()) == x.shape[1], "a bucket must be cut to its longest utterance"
        yl = torch.ceil(x_lengths.double() * 6.0 * float(length_scale)).long()  # deterministic 'durations'
        Ty = int(yl.max())
        y_mask = (torch.arange(Ty).view(1, 1, Ty) < yl.view(-1, 1, 1)).float()
        o = torch.zeros(x.shape[0], 1, Ty * self.hop_length)
        time.sleep(2e-6 * x.shape[0] * Ty)
        self.calls += 1
        if self.timing:
            # a FUSED class, as the 16-bit kernels are: the launches move a quarter of SURVEY 8(d)'s per-conv bytes and
            # finish in the time a copy of 1.2x the per-conv bytes would take at the HBM peak -- so a roofline priced
            # with the per-conv figure reads 1.2 (the round-4 defect), one priced with the launched bytes 0.3
            # (the f32 class is compute-bound: half the f32 MFMA peak)
            frames = float(x.shape[0] * Ty)
            mfl, mby = self._cost()
            perconv = 0.5 * mby * frames
            if getattr(self, "ddtype", "f32") == "f32":
                self.mrf_ms += mfl * frames / (0.5 * 157.3e12) * 1e3
            else:
                self.mrf_ms += perconv / (1.2 * 8.0e12) * 1e3
            self.mrf_bytes += 0.25 * perconv
            self.mrf_launches += 3
        self._last = {"y_lengths_host": yl}
        return o, None, y_mask, None


class StubBackend:
    label = "STUB BACKEND (CPU control-flow test; no kernel ran)"

    def __init__(self, rank, local_rank):
        self.device = torch.device("cpu")
        self.rank = rank
        if os.environ.get("WETTS_STUB_HANG_RANK") == str(rank) and os.environ.get("WETTS_STUB_HANG_AT") == "device":
            time.sleep(3600)  # a rank that never gets its device: it never reaches the rendezvous
        if os.environ.get("WETTS_STUB_CORRUPT_RANK") == str(rank):
            from wetts_amd import sharding
            real = sharding.broadcast_blob

            def corrupt(blob, src=0):
                real(blob, src)
                blob[blob.numel() // 2] += 1.0  # one element wrong on this rank
                return blob
            sharding.broadcast_blob = corrupt

    def sync(self):
        pass

    def make_model(self, mname, n_vocab, n_speakers):
        return StubNet(mname, n_vocab, n_speakers)

    def set_mrf_timing(self, net, on):
        net.timing = bool(on)

    def read_mrf_timing(self, net):
        ms, nl = net.mrf_ms, net.mrf_launches
        net.mrf_ms, net.mrf_launches, net.mrf_bytes = 0.0, 0, 0.0
        return ms, nl

    def read_mrf_bytes(self, net):
        return net.mrf_bytes

    def hifigan_cost(self, cfg):
        import ctypes as C
        from wetts_amd import _lib
        lib = _lib.load()  # the real cost model (a host function of the library)
        fl, by, mfl, mby = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        assert lib.wetts_hifigan_cost(C.byref(cfg), C.byref(fl), C.byref(by), C.byref(mfl), C.byref(mby)) == 0
        return mfl.value, mby.value

    def pcie_pass(self, net, pinned, n, pipelined, infer_kw):
        t0, frames = time.perf_counter(), 0.0
        for _ in range(n):
            for (x, l, s) in pinned:
                net.infer(x, l, sid=s, **infer_kw)
                frames += float(net._last["y_lengths_host"].sum())
        return frames * net.hop_length / (time.perf_counter() - t0)

    def pin(self, t):
        return t


def expected_frames(lens, length_scale):
    return sum(math.ceil(int(v) * 6.0 * length_scale) for v in lens)
