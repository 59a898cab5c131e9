"""``fused_leaky_relu`` / ``FusedLeakyReLU`` with the reference's signatures on the gfx950 HIP kernel.

Mirrors /root/reference/op/fused_act.py:74-97 and the pybind entry op/fused_bias_act.cpp:11-20
(``fused_bias_act(input, bias, refer, act, grad, alpha, scale)``; empty bias/refer = unused,
op/fused_bias_act_kernel.cu:62-63).  Inference only; no CPU fallback (CPU tensors raise like CHECK_CUDA).
"""
import torch
from torch import nn

from .. import _lib


def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Native-op boundary: returns a fresh tensor shaped like ``input``."""
    lib = _lib.load()
    x = _lib.require_cuda_any(input, "input")  # half / float / double, as the reference dispatches (fused_bias_act_kernel.cu:79)
    b = _lib.require_cuda_any(bias, "bias").to(x.dtype) if bias is not None and bias.numel() else None
    r = _lib.require_cuda_any(refer, "refer").to(x.dtype) if refer is not None and refer.numel() else None
    fn = "maua_fused_bias_act_" + _lib.DTYPE_SUFFIX[x.dtype]
    if r is not None and r.numel() != x.numel():
        raise RuntimeError("refer must have as many elements as input")
    step_b = 1
    for d in x.shape[2:]:
        step_b *= d
    y = torch.empty_like(x)
    if x.numel():
        with torch.cuda.device(x.device):
            rc = getattr(lib, fn)(
                x.data_ptr(), _lib.ptr(b), _lib.ptr(r), y.data_ptr(), x.numel(), b.numel() if b is not None else 0,
                max(step_b, 1), int(act), int(grad), float(alpha), float(scale), _lib.stream_ptr(x.device),
            )
        _lib.check(rc, fn)
    return y


def fused_leaky_relu(input, bias, negative_slope=0.2, scale=2 ** 0.5):
    return fused_bias_act(input, bias, None, 3, 0, negative_slope, scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
