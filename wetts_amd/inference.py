"""CLI drop-in for the reference's `wetts/vits/inference.py` (same flags, same test-file format,
same wav output) running the MI355X HIP path.

    python -m wetts_amd.inference --checkpoint G_x.pth --cfg config.json --outdir out \\
        --phone_table phones.txt --speaker_table speaker.txt --test_file test.txt --gpu 0

test-file line: `wav_path|speaker|ph ph ph ...` (inference.py:83-85); tables are `symbol id` lines
(inference.py:55-64).  Writes outdir/basename(wav_path) as int16 at hps.data.sampling_rate with
the reference's peak normalisation (inference.py:100-110).  Extra: --batch N plans the whole test file into
length-sorted padded sub-batches of at most N utterances (wetts_amd/batching.py; the reference loops one utterance
at a time).  Sub-batches are synthesised longest first; the wavs are written and their paths printed in the order
of the test file, each as soon as every line before it is done (with --batch 1 that is the reference's own
interleaving of path and RTF lines).
--decode selects what a batched utterance's audio is: `ragged` = what the reference's one-at-a-time loop gives
for it (default where the model supports it), `padded` = what its infer() returns inside the padded sub-batch."""
import argparse
import sys
import time

import numpy as np
import torch
from scipy.io import wavfile

from . import batching, config as _config
from .models import SynthesizerTrn, load_checkpoint


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="inference")
    parser.add_argument("--checkpoint", required=True, help="checkpoint")
    parser.add_argument("--cfg", required=True, help="config file")
    parser.add_argument("--outdir", required=True, help="ouput directory")
    parser.add_argument("--phone_table", required=True, help="input phone dict")
    parser.add_argument("--speaker_table", default=True, help="speaker table")
    parser.add_argument("--test_file", required=True, help="test file")
    parser.add_argument("--gpu", type=int, default=0, help="gpu id for this local rank")
    parser.add_argument("--batch", type=int, default=1,
                        help="largest padded sub-batch per infer() call (1 = the reference's loop)")
    parser.add_argument("--max_pad_frac", type=float, default=0.08,
                        help="padding share the sub-batch plan may spend (with --batch > 1)")
    parser.add_argument("--decode", choices=["auto", "padded", "ragged"], default="auto",
                        help="with --batch > 1: decode every utterance over its own frames (ragged; auto = when "
                             "the model supports it) or the whole padded sub-batch like infer() (padded)")
    parser.add_argument("--seed", type=int, default=None, help="seed for the sampling noise")
    return parser.parse_args(argv)


def read_table(path):
    table = {}
    for line in open(path):
        arr = line.strip().split()
        if not arr:
            continue
        assert len(arr) == 2, f"bad table line: {line!r}"
        table[arr[0]] = int(arr[1])
    return table


def build_model(args, phone_dict, speaker_dict, hps):
    posterior_channels = hps.data.filter_length // 2 + 1
    if "use_mel_posterior_encoder" in hps.model.keys() and hps.model.use_mel_posterior_encoder:
        posterior_channels = hps.data.n_mel_channels
    net_g = SynthesizerTrn(len(phone_dict), posterior_channels,
                           hps.train.segment_size // hps.data.hop_length,
                           n_speakers=len(speaker_dict), **dict(hps.model.items()))
    if args.gpu < 0 or not torch.cuda.is_available():
        raise SystemExit("the MI355X path needs a HIP device (--gpu >= 0); there is no CPU path")
    net_g = net_g.to(torch.device("cuda", args.gpu)).eval()
    load_checkpoint(args.checkpoint, net_g, None)
    return net_g


def main(argv=None):
    args = get_args(argv)
    print(args)
    phone_dict = read_table(args.phone_table)
    speaker_dict = read_table(args.speaker_table)
    hps = _config.get_hparams_from_file(args.cfg)
    net_g = build_model(args, phone_dict, speaker_dict, hps)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    lines = [l.strip().split("|") for l in open(args.test_file) if l.strip()]
    sr = hps.data.sampling_rate
    seqs = [[phone_dict[s] for s in text.split()] for (_, _, text) in lines]  # KeyError like the reference
    sids = [speaker_dict[spk] for (_, spk, _) in lines]
    ragged = False
    if args.batch <= 1:
        # the reference's loop: one utterance per infer() call (inference.py:83-110)
        buckets = [batching.Bucket([i], max(1, len(s))) for i, s in enumerate(seqs)]
    else:
        # --batch N: the whole file is planned at once -- length-sorted padded sub-batches of at most N
        # utterances with a bounded padding share (wetts_amd/batching.py), written back in file order
        # ragged decode where the model supports it: every utterance's audio is then what the one-at-a-time loop
        # of the reference CLI produces for it, whatever it is batched with
        ragged = net_g.ragged_supported() if args.decode == "auto" else args.decode == "ragged"
        if ragged and not net_g.ragged_supported():
            raise SystemExit("--decode ragged: this model's decoder has no ragged mode (float32 ResBlock1 HiFi-GAN only)")
        buckets = batching.plan([len(s) for s in seqs], 1, max_pad_frac=args.max_pad_frac,
                                max_batch=args.batch, ragged=ragged).buckets[0]
    pending = {}  # file index -> pcm, flushed in test-file order
    nxt = 0
    for bk in buckets:
        st = time.time()
        audio = batching.synthesize(net_g, seqs, sids, noise_scale=0.667, noise_scale_w=0.8, length_scale=1,
                                    buckets=[bk], ragged=ragged)
        n_total = 0
        for i in bk.indices:
            a = audio[i].reshape(1, -1)
            pending[i] = net_g.audio_to_int16(a).cpu().numpy()[0]  # per-utterance peak normalisation (:100-110)
            n_total += a.shape[1]
        torch.cuda.synchronize()
        dt = time.time() - st
        while nxt in pending:  # everything up to the first utterance still outstanding
            audio_path = lines[nxt][0]
            print(audio_path)
            wavfile.write(args.outdir + "/" + audio_path.split("/")[-1], sr, pending.pop(nxt).astype(np.int16))
            nxt += 1
        print("RTF {}".format(dt / (float(n_total) / sr)))
        sys.stdout.flush()
    assert not pending and nxt == len(lines)


if __name__ == "__main__":
    main()
