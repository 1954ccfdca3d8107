"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (wetts_amd/).

Imports the *unmodified* reference VITS code from /root/reference with three absent
third-party imports stubbed (torchaudio, librosa, numba), exactly as SURVEY.md §8(c)
documents.  Two of the stubbed symbols ARE on the inference path and are given their published
behaviour instead of a raiser: torchaudio's InverseSpectrogram (== torch.istft, below) and librosa's
pad_center (OnnxSTFT.__init__, utils/stft.py:283 -- reached by every is_onnx=True Vocos model, i.e. by every
exported graph; rounds 1-5 stubbed it as a raiser, so no such model could be built and the OnnxSTFT head went
unpinned).  Only usable inside the build container (the GPU box has no /root/reference);
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


class _InverseSpectrogram:
    """Stand-in for torchaudio.transforms.InverseSpectrogram (torchaudio is not installed): the
    reference's VocosGenerator (decoders.py:279,304) only constructs it with
    (n_fft, hop_length, win_length, center) and calls it on a complex spectrogram, which in
    torchaudio is exactly torch.istft with a hann window, normalized=False, onesided=True,
    length=None.  This is the ONE non-reference piece in the Vocos golden vectors."""

    def __init__(self, n_fft=400, win_length=None, hop_length=None, center=True, **kw):
        self.n_fft = n_fft
        self.win_length = win_length if win_length is not None else n_fft
        self.hop_length = hop_length if hop_length is not None else self.win_length // 2
        self.center = center

    def __call__(self, spectrogram, length=None):
        import torch
        return torch.istft(spectrogram, self.n_fft, self.hop_length, self.win_length,
                           window=torch.hann_window(self.win_length), center=self.center,
                           normalized=False, onesided=True, length=length, return_complex=False)


def _pad_center(data, size, axis=-1, **kwargs):
    """librosa.util.pad_center (librosa/util/utils.py): zero-pads `data` along `axis` to `size` with the data
    centred (lpad = (size - n) // 2); raises when size < n.  OnnxSTFT.__init__ (utils/stft.py:283) calls it with
    the window and filter_length; for win_length == filter_length (every reference config) it is the identity."""
    import numpy as np
    size = kwargs.get("size", size)
    n = data.shape[axis]
    if size < n:
        raise ValueError(f"Target size ({size}) must be at least input size ({n})")
    lpad = int((size - n) // 2)
    lengths = [(0, 0)] * data.ndim
    lengths[axis] = (lpad, int(size - n - lpad))
    return np.pad(data, lengths, **{k: v for k, v in kwargs.items() if k != "size"})


def install_stubs():
    if "torchaudio" not in sys.modules:
        ta = _stub("torchaudio")
        ta.transforms = _stub("torchaudio.transforms",
                              InverseSpectrogram=_InverseSpectrogram,
                              Spectrogram=_raiser("Spectrogram"),
                              MelSpectrogram=_raiser("MelSpectrogram"),
                              Resample=_raiser("Resample"))
    if "librosa" not in sys.modules:
        lb = _stub("librosa")
        lb.util = _stub("librosa.util", pad_center=_pad_center,
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
