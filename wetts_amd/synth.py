"""Seeded synthetic checkpoints (there is no network for real ones).

Produces a state_dict keyed and shaped exactly like the reference's `G_*.pth["model"]`
(old-style weight-norm pairs `weight_g`/`weight_v` on dec.ups / dec.resblocks / flow WN layers,
SURVEY.md §5 "Checkpoint / resume") so that
  * tests/golden/make_golden.py can load it into the *reference* SynthesizerTrn unchanged, and
  * wetts_amd.checkpoint.pack_blob exercises the real fold + repack path.
Layers the reference zero-initialises (flow `post`, ConvFlow `proj`) get non-zero values so the
couplings / splines are not identities.  The duration heads are biased to ~6 frames per phoneme
(SURVEY.md §8d "duration pinning"): SDP via its ElementwiseAffine, DP via proj.bias = log 6.
"""
import math
import re

import torch

from . import checkpoint

_WN_PATTERNS = [
    re.compile(r"^dec\.ups\.\d+\.weight$"),
    re.compile(r"^dec\.resblocks\.\d+\.convs[12]?\.\d+\.weight$"),
    re.compile(r"^flow\.flows\.\d+\.enc\.(in_layers|res_skip_layers)\.\d+\.weight$"),
    re.compile(r"^flow\.flows\.\d+\.enc\.cond_layer\.weight$"),
]


def is_weight_normed(name):
    return any(p.match(name) for p in _WN_PATTERNS)


def _std_for(name, shape, cfg):
    """Effective weight std chosen so activations stay O(1) through ~70 conv layers."""
    if name.endswith("emb.weight"):
        return cfg.hidden_channels ** -0.5
    if name == "emb_g.weight":
        return 1.0
    if "emb_rel_" in name:
        return shape[-1] ** -0.5
    if name.startswith("dec.ups."):
        cin, _, k = shape
        i = int(name.split(".")[2])
        u = cfg.upsample_rates[i]
        return 1.0 / math.sqrt(cin * k / u)
    if name.startswith("dec.resblocks."):
        return 0.5 / math.sqrt(shape[1] * shape[2])
    if name == "dec.conv_post.weight":
        return 0.7 / math.sqrt(shape[1] * shape[2])
    if name.endswith(".scale"):  # ConvNeXtLayer.scale: the reference initialises it to 1/num_layers
        return 0.0
    if name == "dec.out_conv.weight":  # log-magnitude / phase head: keep exp() well below its clamp
        return 0.5 / math.sqrt(shape[1])
    if name.endswith("cond.weight") or name.endswith("cond_layer.weight"):
        return 0.3 / math.sqrt(shape[1])
    if name.endswith("post.weight"):
        return 0.3 / math.sqrt(shape[1])
    if name == "enc_p.proj.weight":
        return 0.3 / math.sqrt(shape[1])
    if name == "dp.proj.weight" and shape[0] == 1:
        return 0.1 / math.sqrt(shape[1])
    if "convs_sep" in name:
        return 1.0 / math.sqrt(shape[2])
    fan_in = shape[1] * (shape[2] if len(shape) > 2 else 1)
    return 1.0 / math.sqrt(fan_in)


def make_state_dict(cfg, seed=0):
    """Reference-keyed float32 state_dict for `cfg` (wetts_config_t), deterministic in `seed`."""
    g = torch.Generator().manual_seed(int(seed))
    sd = {}
    for name, _, _, shape in checkpoint.blob_layout(cfg):
        def rn(*s):
            return torch.randn(*s, generator=g, dtype=torch.float32)
        if name.endswith(".gamma"):
            sd[name] = 1.0 + 0.1 * rn(*shape)
        elif name.endswith(".beta") or name.endswith(".bias"):
            sd[name] = 0.05 * rn(*shape)
        elif name.endswith(".scale"):
            sd[name] = (1.0 / max(1, cfg.vocos_num_layers)) * (1.0 + 0.2 * rn(*shape))
        elif name == "dp.flows.0.m":
            sd[name] = 0.1 * rn(*shape)
        elif name == "dp.flows.0.logs":
            sd[name] = 0.1 * rn(*shape)
        else:
            w = rn(*shape) * _std_for(name, shape, cfg)
            if is_weight_normed(name):
                dims = tuple(range(1, w.dim()))
                norm = torch.linalg.vector_norm(w, ord=2, dim=dims, keepdim=True)
                gain = norm * (1.0 + 0.05 * rn(shape[0], *([1] * (w.dim() - 1))))
                base = name[:-len("weight")]
                sd[base + "weight_g"] = gain
                sd[base + "weight_v"] = w
            else:
                sd[name] = w
    # duration pinning: ~6 frames / phoneme
    if cfg.use_sdp:
        # reverse EA: logw = (z0 - m0) * exp(-logs0) = 0.1*z0 + log 6
        sd["dp.flows.0.logs"][0, 0] = math.log(10.0)
        sd["dp.flows.0.m"][0, 0] = -10.0 * math.log(6.0)
    else:
        sd["dp.proj.bias"][0] = math.log(6.0)
    return sd


def blob_checksum(blob):
    """Order-sensitive fingerprint of a float32 blob (guards golden fixtures against RNG drift)."""
    b = blob.detach().to(torch.float64)
    idx = torch.arange(1, b.numel() + 1, dtype=torch.float64)
    return float((b * torch.cos(idx * 0.37)).sum())
