"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (wetts_amd/).

Imports the *unmodified* reference VITS code from /root/reference with three unused
third-party imports stubbed (torchaudio, librosa, numba), exactly as SURVEY.md §8(c)
documents.  Only usable inside the build container (the GPU box has no /root/reference);
used by tests/golden/make_golden.py to generate the committed golden vectors and by the
CPU tests that pin oracle/vits_oracle.py against the live reference when it is present.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("WETTS_REFERENCE", "/root/reference")
REF_VITS = os.path.join(REF_ROOT, "wetts", "vits")


def available():
    return os.path.isdir(REF_VITS)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _raiser(what):
    def f(*a, **k):
        raise RuntimeError(f"stubbed third-party symbol {what} was called")
    return f


class _Sub:
    """numba type dummy: void(...), int32[:, :, ::1] etc. all return self."""
    def __getitem__(self, k):
        return self

    def __call__(self, *a, **k):
        return self


def install_stubs():
    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        ta.transforms = _stub("torchaudio.transforms",
                              InverseSpectrogram=_raiser("InverseSpectrogram"),
                              Spectrogram=_raiser("Spectrogram"),
                              MelSpectrogram=_raiser("MelSpectrogram"),
                              Resample=_raiser("Resample"))
    if "librosa" not in sys.modules:
        lb = _stub("librosa")
        lb.util = _stub("librosa.util", pad_center=_raiser("pad_center"),
                        tiny=_raiser("tiny"), normalize=_raiser("normalize"))
        lb.filters = _stub("librosa.filters", mel=_raiser("mel"))
    if "numba" not in sys.modules:
        def jit(*a, **k):
            def deco(fn):
                return fn
            return deco
        s = _Sub()
        _stub("numba", jit=jit, void=s, int32=s, float32=s)


def import_reference():
    """Returns (SynthesizerTrn, task, commons, monotonic_align) from the reference."""
    if not available():
        raise RuntimeError(f"reference not found at {REF_VITS}")
    install_stubs()
    if REF_VITS not in sys.path:
        sys.path.insert(0, REF_VITS)
    from model.models import SynthesizerTrn  # noqa
    from utils import task, commons, monotonic_align  # noqa
    return SynthesizerTrn, task, commons, monotonic_align
