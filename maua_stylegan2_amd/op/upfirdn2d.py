"""``upfirdn2d`` with the reference's Python signature, executed by the gfx950 HIP kernel.

Mirrors /root/reference/op/upfirdn2d.py:145-156 (``upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0))``) and the
pybind entry it wraps, op/upfirdn2d.cpp:12-22 (``upfirdn2d(input[major,h,w,minor], kernel, up_x, up_y, down_x,
down_y, pad_x0, pad_x1, pad_y0, pad_y1)``).  Inference only (the caller runs under ``set_grad_enabled(False)``,
generate_audiovisual.py:104).  There is no CPU fallback: CPU tensors raise, as CHECK_CUDA does (op/upfirdn2d.cpp:7,13).
"""
import torch

from .. import _lib


def upfirdn2d_native_op(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
    """The native-op boundary: input [major, in_h, in_w, minor] -> [major, out_h, out_w, minor] (fresh tensor)."""
    lib = _lib.load()
    x = _lib.require_cuda_any(input, "input")  # half / float / double, as the reference dispatches (upfirdn2d_kernel.cu:313-359)
    k = _lib.require_cuda_any(kernel, "kernel").to(x.dtype)
    fn = "maua_upfirdn2d_" + _lib.DTYPE_SUFFIX[x.dtype]
    if x.dim() != 4 or k.dim() != 2:
        raise RuntimeError("upfirdn2d expects input [major,h,w,minor] and kernel [kh,kw]")
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h = (in_h * up_y + pad_y0 + pad_y1 - kh) // down_y + 1
    out_w = (in_w * up_x + pad_x0 + pad_x1 - kw) // down_x + 1
    if out_h <= 0 or out_w <= 0:
        raise RuntimeError(f"upfirdn2d: empty output {out_h}x{out_w}")
    y = torch.empty((major, out_h, out_w, minor), dtype=x.dtype, device=x.device)
    if major:
        with torch.cuda.device(x.device):
            rc = getattr(lib, fn)(
                x.data_ptr(), k.data_ptr(), y.data_ptr(), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x, down_y,
                pad_x0, pad_x1, pad_y0, pad_y1, _lib.stream_ptr(x.device),
            )
        _lib.check(rc, fn)
    return y


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    """[N,C,H,W] -> [N,C,H',W'] (reference op/upfirdn2d.py:145-156; tensor viewed [N*C,H,W,1], :99)."""
    n, c, h, w = input.shape
    out = upfirdn2d_native_op(input.reshape(n * c, h, w, 1), kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])
    return out.view(n, c, out.shape[1], out.shape[2])
