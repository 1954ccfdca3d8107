"""CLI drop-in for the reference's `wetts/vits/inference.py` (same flags, same test-file format,
same wav output) running the MI355X HIP path.

    python -m wetts_amd.inference --checkpoint G_x.pth --cfg config.json --outdir out \\
        --phone_table phones.txt --speaker_table speaker.txt --test_file test.txt --gpu 0

test-file line: `wav_path|speaker|ph ph ph ...` (inference.py:83-85); tables are `symbol id` lines
(inference.py:55-64).  Writes outdir/basename(wav_path) as int16 at hps.data.sampling_rate with
the reference's peak normalisation (inference.py:100-110).  Extra: --batch N groups lines into
padded batches (the reference loops one utterance at a time)."""
import argparse
import sys
import time

import numpy as np
import torch
from scipy.io import wavfile

from . import config as _config
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
    parser.add_argument("--batch", type=int, default=1, help="utterances per infer() call")
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
    device = net_g.device
    lines = [l.strip().split("|") for l in open(args.test_file) if l.strip()]
    sr = hps.data.sampling_rate
    for i in range(0, len(lines), max(1, args.batch)):
        group = lines[i:i + max(1, args.batch)]
        seqs = [[phone_dict[s] for s in text.split()] for (_, _, text) in group]  # KeyError like ref
        sids = [speaker_dict[spk] for (_, spk, _) in group]
        T = max(len(s) for s in seqs)
        x = torch.zeros(len(seqs), T, dtype=torch.long)
        for b, s in enumerate(seqs):
            x[b, :len(s)] = torch.tensor(s, dtype=torch.long)
        x_len = torch.tensor([len(s) for s in seqs], dtype=torch.long)
        st = time.time()
        o, _, y_mask, _ = net_g.infer(x.to(device), x_len.to(device),
                                      sid=torch.tensor(sids, dtype=torch.long, device=device),
                                      noise_scale=0.667, noise_scale_w=0.8, length_scale=1)
        n_valid = (y_mask[:, 0].sum(1) * net_g.hop_length).long()
        pcm = net_g.audio_to_int16(o, n_valid).cpu().numpy()
        torch.cuda.synchronize()
        dt = time.time() - st
        n_valid = n_valid.cpu().numpy()
        for b, (audio_path, _, _) in enumerate(group):
            print(audio_path)
            wavfile.write(args.outdir + "/" + audio_path.split("/")[-1], sr,
                          pcm[b, :int(n_valid[b])].astype(np.int16))
        print("RTF {}".format(dt / (float(n_valid.sum()) / sr)))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
