"""ctypes binding of libwetts_hip.so (include/wetts_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, this
module raises.  torch is imported first so that the process-wide HIP runtime is the one PyTorch
already loaded (both are `libamdhip64.so.7`; the dynamic linker then resolves our NEEDED entry
to the loaded image and device pointers / streams are shared).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwetts_hip.so")

MAX_STAGES = 8
MAX_RB_KERNELS = 8
MAX_RB_DILATIONS = 8


class WettsError(RuntimeError):
    pass


ABI_VERSION = 9  # WETTS_ABI_VERSION of include/wetts_hip.h this binding was written against


class Config(C.Structure):
    """Mirror of wetts_config_t."""
    _fields_ = [
        ("n_vocab", C.c_int32),
        ("inter_channels", C.c_int32),
        ("hidden_channels", C.c_int32),
        ("filter_channels", C.c_int32),
        ("n_heads", C.c_int32),
        ("n_layers", C.c_int32),
        ("kernel_size", C.c_int32),
        ("window_size", C.c_int32),
        ("resblock", C.c_int32),
        ("n_resblock_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * MAX_RB_KERNELS),
        ("n_resblock_dilations", C.c_int32),
        ("resblock_dilation_sizes", (C.c_int32 * MAX_RB_DILATIONS) * MAX_RB_KERNELS),
        ("n_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * MAX_STAGES),
        ("upsample_kernel_sizes", C.c_int32 * MAX_STAGES),
        ("upsample_initial_channel", C.c_int32),
        ("n_speakers", C.c_int32),
        ("gin_channels", C.c_int32),
        ("use_sdp", C.c_int32),
        ("flow_n_flows", C.c_int32),
        ("flow_wn_layers", C.c_int32),
        ("flow_kernel_size", C.c_int32),
        ("sdp_n_flows", C.c_int32),
        ("dp_filter_channels", C.c_int32),
        ("vocoder_type", C.c_int32),
        ("vocos_channels", C.c_int32),
        ("vocos_h_channels", C.c_int32),
        ("vocos_num_layers", C.c_int32),
        ("istft_n_fft", C.c_int32),
        ("istft_hop_length", C.c_int32),
        ("istft_win_length", C.c_int32),
        ("transformer_flows", C.c_int32),
        ("use_spk_conditioned_encoder", C.c_int32),
        ("is_onnx", C.c_int32),
        ("reserved", C.c_int32 * 6),
    ]


_P = C.c_void_p
_I32 = C.c_int32
_I64 = C.c_int64
_F = C.c_float
_CFG = C.POINTER(Config)

# name -> (restype, argtypes); must list every symbol include/wetts_hip.h declares
SIGNATURES = {
    "wetts_abi_version": (_I32, []),
    "wetts_last_error": (C.c_char_p, []),
    "wetts_blob_num_tensors": (_I32, [_CFG]),
    "wetts_blob_tensor_info": (_I32, [_CFG, _I32, C.c_char_p, C.c_size_t, C.POINTER(_I64),
                                      C.POINTER(_I64), C.POINTER(_I64 * 4)]),
    "wetts_blob_numel": (_I64, [_CFG]),
    "wetts_create": (_I32, [_CFG, _P, _I64, _P, C.POINTER(_P)]),
    "wetts_destroy": (None, [_P]),
    "wetts_hop_length": (_I32, [_P]),
    "wetts_get_blob": (_I32, [_P, _P, _I64, _P]),
    "wetts_workspace_bytes": (_I64, [_P, _I32, _I32, _I32]),
    "wetts_speaker_embedding": (_I32, [_P, _P, _I32, _P, _P]),
    "wetts_text_encoder": (_I32, [_P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _I64, _P]),
    "wetts_duration_sdp": (_I32, [_P, _P, _P, _P, _P, _F, _I32, _I32, _P, _P, _P, _I64, _P]),
    "wetts_duration_dp": (_I32, [_P, _P, _P, _P, _I32, _I32, _P, _P, _I64, _P]),
    "wetts_durations_to_lengths": (_I32, [_P, _P, _F, _I32, _I32, _P, _P, _P, _P, _P]),
    "wetts_set_status_word": (_I32, [_P, _P, _P]),
    "wetts_set_seed": (_I32, [_P, C.c_uint64]),
    "wetts_randn": (_I32, [_P, _I64, C.c_uint64, C.c_uint64, _P]),
    "wetts_mask_rows": (_I32, [_P, _P, _I32, _I32, _I32, _P, _P]),
    "wetts_length_regulate": (_I32, [_P, _P, _P, _P, _P, _P, _I64, _I64, _F, _I32, _I32, _I32,
                                     _P, _P, _P, _P, _P, _P, _P]),
    "wetts_flow_reverse": (_I32, [_P, _P, _P, _P, _I32, _I32, _P, _P, _I64, _P]),
    "wetts_hifigan": (_I32, [_P, _P, _I64, _I64, _P, _I64, _P, _I32, _I32, _P, _P, _I64, _P]),
    "wetts_hifigan_ragged_supported": (_I32, [_P]),
    "wetts_hifigan_ragged": (_I32, [_P, _P, _I64, _I64, _P, _P, _I32, _I32, _P, _P, _I64, _P]),
    "wetts_set_decoder_precision": (_I32, [_P, _I32]),
    "wetts_set_flow_precision": (_I32, [_P, _I32]),
    "wetts_set_istft_mode": (_I32, [_P, _I32]),
    "wetts_get_istft_mode": (_I32, [_P]),
    "wetts_dynamic_quant_conv1d": (_I32, [_P, _P, _P, _I32, _I32, _I32, _I32, _I32, _I32, _I32, _P, _P]),
    "wetts_mas": (_I32, [_P, _P, _P, _I32, _I32, _I32, _P, _P, _I64, _P]),
    "wetts_audio_to_int16": (_I32, [_P, _P, _I32, _I64, _P, _P]),
    "wetts_infer_workspace_bytes": (_I64, [_P, _I32, _I32, _I32]),
    "wetts_infer": (_I32, [_P, _P, _P, _P, _P, _P, _F, _F, _F, _I32, _I32, _I32, _P, _P,
                           C.POINTER(_I32), _P, _I64, _P]),
    "wetts_hifigan_cost": (_I32, [_CFG, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "wetts_set_mrf_timing": (_I32, [_P, _I32]),
    "wetts_read_mrf_timing": (_I32, [_P, C.POINTER(C.c_double), C.POINTER(_I64), C.POINTER(_I32)]),
    "wetts_read_mrf_bytes": (_I32, [_P, C.POINTER(C.c_double)]),
    "wetts_profile_hifigan": (_I32, [_P, _P, _I64, _I64, _P, _I32, _I32, _P, _P, _I64, _P,
                                     C.POINTER(C.c_double), C.POINTER(C.c_double),
                                     C.POINTER(_I32)]),
}

# WETTS_STATUS_* bits of include/wetts_hip.h
STATUS_SPLINE_DOMAIN, STATUS_PHONE_ID_RANGE, STATUS_SPEAKER_ID_RANGE, STATUS_DURATION_NONFINITE = 1, 2, 4, 8

_lib = None


def load():
    """Loads (once) and returns the CDLL.  Raises WettsError if the library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WettsError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m wetts_amd.build`). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError => header / library mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.wetts_abi_version() != ABI_VERSION:
        raise WettsError("libwetts_hip.so ABI version mismatch")
    _lib = lib
    return lib


def last_error():
    return load().wetts_last_error().decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise WettsError(f"{what} failed (code {rc}): {last_error()}")


def ptr(t):
    """Device (or host) address of a torch tensor / None."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
