"""CPU tier, property tests (hypothesis) of the host logic either side of the hot path: the chunk / overlap-discard
protocols of the streaming clients (inference_onnx.py:37-76, stream_tts/1/model.py:58-111) driven through
`session.stream_decode` with a stand-in decoder, and the batching plan (the Triton generator's padded-batch call shape,
gpu_triton/model_repo/tts/1/model.py:85-165).  The known-answer fixtures lifted from the reference pin the protocols at
the reference's own parameters; these tests hold the invariants at every length / window / padding."""
import numpy as np
from hypothesis import example, given, settings, strategies as st

from wetts_amd import batching, session

HOP = 4


class _PointwiseDecoder:
    """A "vocoder" without a receptive field: frame f -> samples f * HOP + j carrying the frame's own id.  With it the
    windowed decode must reproduce the one-shot decode EXACTLY, so any sample the protocol drops, repeats or takes from a
    padded / reflected frame shows up."""

    class model:
        hop_length = HOP

    def run(self, _names, feed):
        z = np.asarray(feed["z"])  # [1, L, C]
        ids = z[0, :, 0]
        return [(ids[:, None] * HOP + np.arange(HOP)[None, :]).reshape(1, 1, -1).astype(np.float32)]


def _z(L):
    z = np.zeros((1, L, 3), dtype=np.float32)
    z[0, :, 0] = np.arange(L)
    return z


@settings(max_examples=300, deadline=None)
@given(L=st.integers(1, 400), block=st.integers(1, 120), pad=st.integers(0, 30))
def test_stream_decode_tiles_the_utterance_exactly(L, block, pad):
    full = _PointwiseDecoder().run(None, {"z": _z(L)})[0].reshape(-1)
    pieces = list(session.stream_decode(_PointwiseDecoder(), _z(L), np.array([0]), chunk_size=block, pad_size=pad))
    assert len(pieces) == -(-L // block)
    got = np.concatenate(pieces)
    assert got.shape == full.shape and np.array_equal(got, full)
    # every window but the last yields exactly one block of audio (what the streaming clients play per step)
    assert all(len(p) == block * HOP for p in pieces[:-1])


@settings(max_examples=300, deadline=None)
@given(L=st.integers(1, 400), block=st.integers(2, 120), pad=st.integers(0, 30), min_chunk=st.integers(1, 90))
def test_stream_decode_min_chunk_protocol_tiles_the_utterance_exactly(L, block, pad, min_chunk):
    """The Triton twin's variant: a short last window is reflect-padded before decoding and the audio of the reflected
    frames is cut off again -- no reflected frame may reach the output."""
    full = _PointwiseDecoder().run(None, {"z": _z(L)})[0].reshape(-1)
    wins, pad_end = session.get_chunks_min(L, block, pad, min_chunk)
    last = wins[-1][1] - wins[-1][0]
    assert (pad_end is None) == (last >= min_chunk) and (pad_end is None or last + pad_end == min_chunk)
    if pad_end is not None and last < 2:
        return  # numpy cannot reflect a single frame (np.pad raises there too -- the reference's behaviour)
    got = np.concatenate(list(session.stream_decode(_PointwiseDecoder(), _z(L), np.array([0]), chunk_size=block,
                                                    pad_size=pad, min_chunk=min_chunk)))
    assert got.shape == full.shape and np.array_equal(got, full)


@settings(max_examples=60, deadline=None)
@given(L=st.integers(1, 300), pad=st.integers(0, 30))
def test_single_window_mode(L, pad):
    assert session.get_chunks(L, -1, pad) == [(0, L)] and session.get_chunks_min(L, -1, pad) == ([(0, L)], None)
    # ... in which the reference's depadding keeps `audio[:, :block * upsample]` = `audio[:, :-upsample]`
    # (inference_onnx.py:69-70 with block = -1): everything but the last frame's samples.  Mirrored, not "fixed".
    full = _PointwiseDecoder().run(None, {"z": _z(L)})[0].reshape(-1)
    got = np.concatenate(list(session.stream_decode(_PointwiseDecoder(), _z(L), np.array([0]), chunk_size=-1, pad_size=pad)))
    assert np.array_equal(got, full[:-HOP])


@settings(max_examples=150, deadline=None)
@given(lens=st.lists(st.integers(1, 300), min_size=0, max_size=90), world=st.integers(1, 8),
       mpf=st.floats(0.0, 0.5), max_batch=st.sampled_from([0, 1, 2, 7, 32]), ragged=st.booleans(),
       call_cost=st.floats(0.0, 500.0))
# found by this test in round 5: a plain cap of ceil(n / world) dealt 43 utterances over 3 ranks as 15 / 15 / 13
@example(lens=[1] * 42 + [3], world=3, mpf=0.0, max_batch=0, ragged=False, call_cost=0.0)
def test_plan_invariants(lens, world, mpf, max_batch, ragged, call_cost):
    pl = batching.plan(lens, world, max_pad_frac=mpf, call_cost=call_cost, max_batch=max_batch, ragged=ragged)
    assert len(pl.buckets) == world and len(pl.shards) == world
    # a partition of the utterances: each exactly once, each rank's buckets = its shard, in shard order
    assert sorted(i for bs in pl.buckets for b in bs for i in b.indices) == list(range(len(lens)))
    assert [[i for b in bs for i in b.indices] for bs in pl.buckets] == pl.shards
    counts = [len(s) for s in pl.shards]
    assert max(counts) - min(counts) <= 1  # equal counts per rank (LPT deal with a count cap)
    for bs in pl.buckets:
        for b in bs:
            assert len(b) >= 1 and b.tx == max(lens[i] for i in b.indices)
            assert [lens[i] for i in b.indices] == sorted((lens[i] for i in b.indices), reverse=True)
            assert not max_batch or len(b) <= max_batch
        # the padding bound holds on every rank: the share of the COST that is padding, a padded slot of a ragged decode
        # costing RAGGED_PAD_WEIGHT of a valid one (only the masked stages pay for it)
        padded = sum(len(b) * b.tx for b in bs)
        valid = sum(lens[i] for b in bs for i in b.indices)
        pad = (padded - valid) * (batching.RAGGED_PAD_WEIGHT if ragged else 1.0)
        assert padded == 0 or pad / (valid + pad) <= mpf + 1e-9
    res = [[("u", i) for b in bs for i in b.indices] for bs in pl.buckets]
    assert batching.unshard(pl, res) == [("u", i) for i in range(len(lens))]
