"""Batched synthesis: how a list of utterances becomes infer() calls.

The reference synthesises one utterance per call in its CLI (wetts/vits/inference.py:83-110) and one
padded batch per call behind Triton (runtime/gpu_triton/model_repo/tts/1/model.py:85-165: dynamic batch
<= 32, every row padded to the longest, the padded tail decoded and returned).  The decoder has no masks
(decoders.py:63-82), so a padded batch costs B x max(frames) whatever the utterances' own lengths are.

`plan()` turns per-utterance phoneme counts into the work list of every rank:
  1. ranks: longest-processing-time-first deal (`sharding.shard_utterances`; equal counts, loads within one
     utterance of each other), so no rank idles while another decodes -- no collective is involved;
  2. buckets: each rank's shard, sorted by length, is cut into padded sub-batches by a dynamic programme
     that minimises   sum_buckets ( n_b * max_len_b + call_cost )   -- padded phoneme slots plus a fixed
     price per infer() call (launch-bound stages, the one host sync) -- subject to `max_batch`;
     if the padding share of that optimum is above `max_pad_frac` the call price is lowered until it is not.
`synthesize()` runs a plan on one rank and returns the valid audio of every utterance in input order;
`unshard()` puts per-rank results back in global order.

The padding share is measured in phoneme slots; frames per phoneme are sampled per utterance, so the frame
padding an executed plan really paid is reported by `synthesize()` (`stats["frame_pad_frac"]`).
"""
from dataclasses import dataclass, field
from typing import List, Sequence

from . import sharding

# price of one infer() call in phoneme slots (a slot = one phoneme position of one utterance).  Measured on
# MI355X with AISHELL-3 v1, 64 ragged utterances (profiles/r03_bucket_sweep.txt): going from 4 to 9 calls removed
# 9 % of the padded work and gained nothing -- a call costs ~3.7 ms beyond its per-slot work (launch-bound stages,
# the host sync, the decoder's tail at small batch), a slot ~0.036 ms: ~100 slots
DEFAULT_CALL_COST = 100.0
# share of a step that is spent in the MASKED stages (text encoder, durations, flow): with a ragged decode
# (SynthesizerTrn.infer(ragged=True)) only they pay for padding, so a padded slot costs this fraction of a slot
RAGGED_PAD_WEIGHT = 0.1


@dataclass
class Bucket:
    indices: List[int]  # global utterance indices, longest first
    tx: int             # padded phoneme length of the sub-batch (= the longest member)

    def __len__(self):
        return len(self.indices)


@dataclass
class Plan:
    lengths: List[int]
    shards: List[List[int]]                   # per rank: global indices, longest first
    buckets: List[List[Bucket]]               # per rank
    call_cost: float
    stats: dict = field(default_factory=dict)

    def rank_buckets(self, rank):
        return self.buckets[rank]


def bucketize(sorted_lengths: Sequence[int], call_cost: float = DEFAULT_CALL_COST, max_batch: int = 0):
    """Optimal cut of a DESCENDING length list into consecutive buckets: minimises
    sum (n_b * first_len_b + call_cost).  Returns [(start, end), ...].  O(n * max_batch)."""
    n = len(sorted_lengths)
    if n == 0:
        return []
    assert all(sorted_lengths[i] >= sorted_lengths[i + 1] for i in range(n - 1)), "lengths must be sorted descending"
    import numpy as np
    mb = n if not max_batch or max_batch <= 0 else min(n, max_batch)
    best = np.full(n + 1, np.inf)   # best[i] = cost of cutting the suffix i..n
    nxt = [n] * (n + 1)
    best[n] = 0.0
    for i in range(n - 1, -1, -1):  # the inner minimisation over j is one vector op (n = thousands of utterances)
        li = max(1, int(sorted_lengths[i]))
        hi = min(n, i + mb)
        c = np.arange(1, hi - i + 1, dtype=np.float64) * li + call_cost + best[i + 1:hi + 1]
        k = len(c) - 1 - int(np.argmin(c[::-1]))  # ties go to the LARGER bucket (equal lengths at call_cost 0 stay one)
        best[i], nxt[i] = c[k], i + 1 + k
    cuts, i = [], 0
    while i < n:
        cuts.append((i, nxt[i]))
        i = nxt[i]
    return cuts


def pad_fraction(sorted_lengths, cuts):
    """Share of the padded slots that are padding."""
    padded = sum((b - a) * max(1, int(sorted_lengths[a])) for a, b in cuts)
    valid = sum(max(1, int(v)) for v in sorted_lengths)
    return 0.0 if padded == 0 else 1.0 - valid / padded


def weighted_pad_fraction(sorted_lengths, cuts, pad_weight):
    """Padding share of the COST: a padded slot costs `pad_weight` of a valid one (1 = padded decode)."""
    padded = sum((b - a) * max(1, int(sorted_lengths[a])) for a, b in cuts)
    valid = sum(max(1, int(v)) for v in sorted_lengths)
    pad = (padded - valid) * pad_weight
    return 0.0 if padded == 0 else pad / (valid + pad)


def plan(lengths: Sequence[int], world: int = 1, max_pad_frac: float = 0.08, call_cost: float = DEFAULT_CALL_COST,
         max_batch: int = 0, ragged: bool = False) -> Plan:
    """Work list of every rank for `lengths` (phoneme counts).  `max_pad_frac` bounds the padding share of
    every rank's buckets (singleton buckets always satisfy it, so the bound is always met).  `ragged`: the
    buckets will be decoded with infer(ragged=True), where only the masked stages pay for padding
    (RAGGED_PAD_WEIGHT of a slot) -- the plan then makes few, large calls."""
    lengths = [int(v) for v in lengths]
    shards = sharding.shard_utterances(lengths, world)
    w = RAGGED_PAD_WEIGHT if ragged else 1.0
    all_buckets, fracs, used_cost = [], [], call_cost
    for idxs in shards:
        ls = [lengths[i] for i in idxs]
        cost = call_cost
        cuts = bucketize(ls, cost / w, max_batch)
        while weighted_pad_fraction(ls, cuts, w) > max_pad_frac and cost > 1e-3:
            cost *= 0.5
            cuts = bucketize(ls, cost / w, max_batch)
        if weighted_pad_fraction(ls, cuts, w) > max_pad_frac:
            cuts = bucketize(ls, 0.0, max_batch)
        used_cost = min(used_cost, cost)
        fracs.append(pad_fraction(ls, cuts))
        all_buckets.append([Bucket([idxs[k] for k in range(a, b)], max(1, ls[a])) for a, b in cuts])
    loads = [sum(len(b) * b.tx for b in bs) for bs in all_buckets]
    stats = {"world": world, "utterances": len(lengths), "ragged": bool(ragged),
             "buckets_per_rank": [len(b) for b in all_buckets],
             "pad_frac_per_rank": fracs, "pad_frac": max(fracs) if fracs else 0.0,
             "padded_slots_per_rank": loads,
             "imbalance": (max(loads) / (sum(loads) / len(loads)) - 1.0) if loads and sum(loads) else 0.0}
    return Plan(lengths, shards, all_buckets, used_cost, stats)


def equal_count_buckets(idxs: Sequence[int], lengths: Sequence[int], n_buckets: int) -> List[Bucket]:
    """Round-2 behaviour (fixed number of equal-count buckets over a length-sorted shard); kept for A/B."""
    n_buckets = max(1, min(int(n_buckets), len(idxs)))
    per = -(-len(idxs) // n_buckets)
    return [Bucket(list(idxs[i:i + per]), max(1, max(lengths[j] for j in idxs[i:i + per])))
            for i in range(0, len(idxs), per)]


def unshard(plan_: Plan, per_rank_results):
    """per_rank_results[r] = list of results in the order of plan.shards[r] -> global input order."""
    return sharding.unshard(plan_.shards, per_rank_results)


def synthesize(net, seqs, sids=None, noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8,
               max_pad_frac=0.08, max_batch=0, call_cost=DEFAULT_CALL_COST, buckets=None, return_stats=False,
               ragged="auto"):
    """Synthesises a list of phoneme-id sequences on `net`'s device: plans buckets (or takes `buckets`, a list of
    `Bucket`), runs one `net.infer()` per bucket and returns the VALID audio of every utterance (1-D float32
    device tensors, `y_lengths * hop` samples each) in input order.

    ragged = True / "auto" (when the model supports it: float32 ResBlock1 HiFi-GAN): every utterance is decoded
    over its own frames -- the audio the reference returns when it synthesises that utterance alone, its CLI's
    call shape (inference.py:83-110) -- and the plan makes few large calls, because padding then only costs in the
    masked stages.  ragged = False: every utterance's audio is what the reference's infer() returns for it inside
    the padded batch of its bucket (runtime/gpu_triton/model_repo/tts/1/model.py:85-165)."""
    import torch
    n = len(seqs)
    lens = [len(s) for s in seqs]
    if ragged == "auto":
        ragged = bool(net.ragged_supported())
    if buckets is None:
        buckets = plan(lens, 1, max_pad_frac, call_cost, max_batch, ragged=ragged).buckets[0] if n else []
    dev = net.device
    out = [None] * n
    valid_frames = padded_frames = 0
    for b in buckets:
        B, tx = len(b), b.tx
        x = torch.zeros(B, tx, dtype=torch.long)
        for r, i in enumerate(b.indices):
            x[r, :lens[i]] = torch.as_tensor(seqs[i], dtype=torch.long)
        xl = torch.tensor([lens[i] for i in b.indices], dtype=torch.long)
        sid = None if sids is None else torch.tensor([int(sids[i]) for i in b.indices], dtype=torch.long)
        up = getattr(net, "upload", None) or (lambda t, dtype=None: t.to(dev))
        o, _, y_mask, _ = net.infer(up(x), up(xl), sid=None if sid is None else up(sid),
                                    noise_scale=noise_scale, length_scale=length_scale,
                                    noise_scale_w=noise_scale_w, ragged=ragged)
        yl = net._last["y_lengths_host"]
        hop = net.hop_length
        for r, i in enumerate(b.indices):
            out[i] = o[r, 0, :int(yl[r]) * hop]
        valid_frames += int(yl.sum())
        padded_frames += B * int(y_mask.shape[-1])
    if return_stats:
        return out, {"calls": len(buckets), "ragged": bool(ragged), "valid_frames": valid_frames,
                     "padded_frames": padded_frames,
                     "frame_pad_frac": 0.0 if not padded_frames else 1.0 - valid_frames / padded_frames}
    return out
