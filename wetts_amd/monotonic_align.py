"""monotonic_align.maximum_path drop-in (wetts/vits/utils/monotonic_align.py:6-19) on the HIP
kernel in csrc/mas.hip: no GPU->CPU->GPU round trip, one workgroup per utterance."""
import torch

from . import _lib


def maximum_path(neg_cent: torch.Tensor, mask: torch.Tensor):
    """neg_cent: [b, t_t, t_s], mask: [b, t_t, t_s]  ->  path [b, t_t, t_s] in neg_cent's dtype
    and device (same contract as the reference wrapper)."""
    lib = _lib.load()
    if neg_cent.device.type != "cuda":
        raise _lib.WettsError("maximum_path: tensors must live on the HIP device (no CPU path)")
    device, dtype = neg_cent.device, neg_cent.dtype
    nc = neg_cent.detach().to(torch.float32).contiguous()
    b, t_t, t_s = nc.shape
    # t_t_max = mask.sum(1)[:, 0], t_s_max = mask.sum(2)[:, 0]  (monotonic_align.py:16-17)
    t_ys = mask.sum(1)[:, 0].to(torch.int32).contiguous()
    t_xs = mask.sum(2)[:, 0].to(torch.int32).contiguous()
    path = torch.empty(b, t_t, t_s, dtype=torch.int32, device=device)
    ws = torch.empty(max(1, b * t_t * t_s), dtype=torch.float32, device=device)
    with torch.cuda.device(device):
        _lib.check(lib.wetts_mas(_lib.ptr(nc), _lib.ptr(t_ys), _lib.ptr(t_xs), b, t_t, t_s,
                                 _lib.ptr(path), _lib.ptr(ws), ws.numel() * 4,
                                 _lib.current_stream_ptr()), "wetts_mas")
    return path.to(dtype=dtype)
