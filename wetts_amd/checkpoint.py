"""Checkpoint handling: reference `G_*.pth` state_dict -> folded float32 weight blob.

Replaces utils/task.py:31-56 (load_checkpoint) + Generator/WN.remove_weight_norm
(decoders.py:84-88, modules.py:89-95).  The blob order is owned by the C library
(wetts_blob_tensor_info); this module only fills it.
"""
import ctypes as C
import logging

import torch

from . import _lib

logger = logging.getLogger(__name__)


def blob_layout(cfg):
    """[(name, offset, numel, shape)] as the library defines it (offsets / sizes in floats)."""
    lib = _lib.load()
    n = lib.wetts_blob_num_tensors(C.byref(cfg))
    if n < 0:
        raise _lib.WettsError(f"invalid config: {_lib.last_error()}")
    out = []
    buf = C.create_string_buffer(256)
    off, num, shape = C.c_int64(), C.c_int64(), (C.c_int64 * 4)()
    for i in range(n):
        _lib.check(lib.wetts_blob_tensor_info(C.byref(cfg), i, buf, 256, C.byref(off),
                                              C.byref(num), C.byref(shape)), "blob_tensor_info")
        shp = tuple(int(s) for s in shape if s > 0)
        out.append((buf.value.decode(), int(off.value), int(num.value), shp))
    return out


def blob_numel(cfg):
    n = _lib.load().wetts_blob_numel(C.byref(cfg))
    if n < 0:
        raise _lib.WettsError(f"invalid config: {_lib.last_error()}")
    return int(n)


def fold_weight_norm(state_dict):
    """Collapses old-style weight-norm pairs: w = g * v / ||v|| with the norm over every dim but 0
    (torch.nn.utils.weight_norm default dim=0; for ConvTranspose1d that is the *input*-channel dim,
    weight [Cin,Cout,k], exactly as torch._weight_norm computes it).  Also accepts the
    parametrizations spelling (`parametrizations.weight.original0/1`)."""
    out = {}
    for k, v in state_dict.items():
        if k.endswith(".weight_g") or k.endswith(".weight_v"):
            continue
        if ".parametrizations.weight.original" in k:
            continue
        out[k] = v
    for k, g in state_dict.items():
        if k.endswith(".weight_g"):
            base = k[:-len("weight_g")]
            v = state_dict[base + "weight_v"]
        elif k.endswith(".parametrizations.weight.original0"):
            base = k[:-len("parametrizations.weight.original0")]
            v = state_dict[base + "parametrizations.weight.original1"]
        else:
            continue
        g = g.detach().to(torch.float32)
        v = v.detach().to(torch.float32)
        dims = tuple(range(1, v.dim()))
        norm = torch.linalg.vector_norm(v, ord=2, dim=dims, keepdim=True)
        out[base + "weight"] = v * (g / norm)
    return out


def pack_blob(cfg, state_dict, strict=True):
    """Natural-layout float32 CPU blob from a (reference-keyed) state_dict."""
    sd = fold_weight_norm(state_dict)
    blob = torch.zeros(blob_numel(cfg), dtype=torch.float32)
    missing = []
    for name, off, numel, shape in blob_layout(cfg):
        t = sd.get(name)
        if t is None:
            missing.append(name)
            continue
        t = t.detach().to(torch.float32).contiguous()
        if tuple(t.shape) != shape:
            raise ValueError(f"{name}: checkpoint shape {tuple(t.shape)} != expected {shape}")
        blob[off:off + numel] = t.reshape(-1)
    if missing:
        # the reference tolerates missing keys (task.py:44-49: keeps the module's init value).  There
        # is no random init to keep here, so strict mode (the default, and what load_checkpoint uses)
        # refuses; non-strict fills the DETERMINISTIC init values of the reference (LayerNorm
        # gamma = 1, normalization.py:12; ConvNeXt scale = 1/num_layers, decoders.py:236-238), leaves
        # the randomly initialised rest at zero, and reports every name.
        msg = f"{len(missing)} tensors missing from checkpoint: {missing}"
        if strict:
            raise KeyError(msg)
        for name, off, numel, shape in blob_layout(cfg):
            if name in missing and name.endswith(".gamma"):
                blob[off:off + numel] = 1.0
            elif name in missing and name.endswith(".scale"):
                blob[off:off + numel] = 1.0 / max(1, int(cfg.vocos_num_layers))
        logger.warning(msg)
    return blob


def load_state_dict_file(checkpoint_path):
    """Reads `G_*.pth` as saved by task.save_checkpoint (task.py:59-76): {"model": state_dict,
    "iteration", "optimizer", "learning_rate"}; a bare state_dict is accepted too."""
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    if isinstance(ckpt, dict) and "model" in ckpt and isinstance(ckpt["model"], dict):
        return ckpt["model"], ckpt.get("iteration"), ckpt.get("learning_rate")
    return ckpt, None, None
