"""wetts_amd -- MI355X (gfx950) native VITS inference behind the WeTTS API.

Python host code + a C-ABI shared library of hand-written HIP kernels (include/wetts_hip.h).
Only what the SynthesizerTrn.infer() hot path needs lives here.
"""
from .config import HParams, get_hparams_from_file, make_config, MODEL_CONFIGS  # noqa: F401
from .models import SynthesizerTrn, load_checkpoint  # noqa: F401

__all__ = ["SynthesizerTrn", "load_checkpoint", "HParams", "get_hparams_from_file", "make_config",
           "MODEL_CONFIGS"]
