"""ctypes loader of libwetts_bench.so (wetts_amd/csrc/bench_abi.h): kernel micro-benchmarks.
Measurement tooling; the product library is loaded first so both share one HIP runtime."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wetts_amd import _lib  # noqa: E402

_I32, _D = C.c_int32, C.POINTER(C.c_double)


def load():
    _lib.load()
    lib = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libwetts_bench.so"))
    lib.wetts_bench_conv.argtypes = [_I32] * 9 + [_D, _D]
    lib.wetts_bench_conv.restype = _I32
    lib.wetts_set_conv_variant.argtypes = [_I32]
    lib.wetts_bench_mfma_peak.argtypes = [_I32, _I32, _I32, _D, _D]
    lib.wetts_bench_resblock.argtypes = [_I32] * 9 + [_D, _D]
    lib.wetts_bench_resblock.restype = _I32
    lib.wetts_bench_mfma_loop.argtypes = [_I32, _I32, _I32, _I32, _D, _D]
    lib.wetts_bench_mfma_loop2.argtypes = [_I32, _I32, _I32, _I32, _D, _D]
    lib.wetts_bench_mfma_valu.argtypes = [_I32, _I32, _I32, _D, _D, _D]
    lib.wetts_bench_mfma16_loop.argtypes = [_I32] * 5 + [_D, _D, _D]
    lib.wetts_bench_mfma16_loop.restype = _I32
    return lib


last_error = _lib.last_error
