"""CPU tier: wetts_amd.batching -- the product-level plan that turns utterance lengths into per-rank padded
sub-batches (the reference's batched call shape: runtime/gpu_triton/model_repo/tts/1/model.py:85-165)."""
import itertools

import numpy as np
import pytest

from wetts_amd import batching


def _cost(ls, cuts, call_cost):
    return sum((b - a) * ls[a] + call_cost for a, b in cuts)


def _brute(ls, call_cost, max_batch):
    n, best = len(ls), None
    for k in range(n):
        for cut_pts in itertools.combinations(range(1, n), k):
            edges = [0, *cut_pts, n]
            cuts = list(zip(edges[:-1], edges[1:]))
            if max_batch and any(b - a > max_batch for a, b in cuts):
                continue
            c = _cost(ls, cuts, call_cost)
            if best is None or c < best:
                best = c
    return best


@pytest.mark.parametrize("seed", range(6))
def test_bucketize_is_optimal_against_brute_force(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 10))
    ls = sorted((int(v) for v in rng.integers(1, 130, n)), reverse=True)
    for call_cost, mb in ((40.0, 0), (0.0, 0), (7.5, 3), (1000.0, 4)):
        cuts = batching.bucketize(ls, call_cost, mb)
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        assert not mb or max(b - a for a, b in cuts) <= mb
        assert abs(_cost(ls, cuts, call_cost) - _brute(ls, call_cost, mb)) < 1e-9


def test_plan_partitions_bounds_padding_and_round_trips():
    rng = np.random.default_rng(0)
    lens = rng.integers(32, 129, size=512).tolist()  # BASELINE.json configs[3]: 512 utterances, Tx ~ U{32..128}
    for world in (1, 2, 8):
        for mpf in (0.15, 0.08, 0.02):
            pl = batching.plan(lens, world, max_pad_frac=mpf)
            flat = sorted(i for bs in pl.buckets for b in bs for i in b.indices)
            assert flat == list(range(512))
            assert [sorted(i for b in bs for i in b.indices) for bs in pl.buckets] == [sorted(s) for s in pl.shards]
            for bs in pl.buckets:
                for b in bs:
                    assert b.tx == max(lens[i] for i in b.indices)
                    assert [lens[i] for i in b.indices] == sorted((lens[i] for i in b.indices), reverse=True)
                padded = sum(len(b) * b.tx for b in bs)
                valid = sum(lens[i] for b in bs for i in b.indices)
                assert 1 - valid / padded <= mpf + 1e-12
            assert pl.stats["pad_frac"] <= mpf + 1e-12 and pl.stats["imbalance"] < 0.05
            res = [[("u", i) for b in bs for i in b.indices] for bs in pl.buckets]
            # results come back bucket by bucket = shard order, because buckets are consecutive cuts of the shard
            assert [[i for b in bs for i in b.indices] for bs in pl.buckets] == pl.shards
            assert batching.unshard(pl, res) == [("u", i) for i in range(512)]
    # a tighter padding bound costs calls, never correctness
    n8 = sum(batching.plan(lens, 8, max_pad_frac=0.08).stats["buckets_per_rank"])
    n2 = sum(batching.plan(lens, 8, max_pad_frac=0.02).stats["buckets_per_rank"])
    assert n2 >= n8


def test_plan_edge_cases():
    assert batching.plan([], 1).buckets == [[]]
    pl = batching.plan([128] * 16, 1)  # equal lengths (configs[1]): one call
    assert [len(b) for b in pl.buckets[0]] == [16] and pl.stats["pad_frac"] == 0.0
    pl = batching.plan([128] * 16, 1, call_cost=0.0)
    assert [len(b) for b in pl.buckets[0]] == [16]
    pl = batching.plan([128] * 70, 1, max_batch=32)  # the Triton generator's max_batch_size
    assert max(len(b) for b in pl.buckets[0]) <= 32 and sum(len(b) for b in pl.buckets[0]) == 70
    pl = batching.plan([5, 100], 1, max_pad_frac=0.0)
    assert [b.indices for b in pl.buckets[0]] == [[1], [0]]
    eq = batching.equal_count_buckets(list(range(10)), list(range(10, 0, -1)), 4)
    assert [len(b) for b in eq] == [3, 3, 3, 1] and eq[0].tx == 10
