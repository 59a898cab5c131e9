"""Oracle (test infrastructure, CPU fp32) for the two native ops of the reference.

  upfirdn2d          <- /root/reference/op/upfirdn2d.py:145-200 (dispatcher + native fallback),
                        kernel semantics op/upfirdn2d_kernel.cu:107-207 (flipped taps = true convolution)
  fused_leaky_relu   <- /root/reference/op/fused_act.py:86-97, op/fused_bias_act_kernel.cu:18-49

Pinned by tests/golden/ops_*.npz (generated from the imported reference by tests/golden/make_golden.py).
"""
import numpy as np
import torch
import torch.nn.functional as F


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, k):
    """op/upfirdn2d.py:102-103 / :197-198."""
    return (in_size * up + pad0 + pad1 - k) // down + 1


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """[N,C,H,W] -> [N,C,H',W'].  Zero-stuff by ``up``, pad (negative = crop), convolve with ``kernel``
    (true convolution, i.e. correlate with the flipped taps), keep every ``down``-th sample."""
    return upfirdn2d_xy(x, kernel, up, up, down, down, pad[0], pad[1], pad[0], pad[1])


def upfirdn2d_xy(x, kernel, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    n, c, h, w = x.shape
    kh, kw = kernel.shape
    planes = x.reshape(n * c, 1, h, w).to(torch.float32)
    # zero-stuffed canvas: sample (y, x) lands on (y*up_y, x*up_x); up-1 trailing zeros after the last one
    stuffed = planes.new_zeros(n * c, 1, h * up_y, w * up_x)
    stuffed[:, :, ::up_y, ::up_x] = planes
    # pad >= 0 adds zeros, pad < 0 crops (op/upfirdn2d.py:171-181)
    stuffed = F.pad(stuffed, [px0, px1, py0, py1])
    taps = torch.flip(kernel.to(torch.float32), [0, 1]).reshape(1, 1, kh, kw)
    full = F.conv2d(stuffed, taps)
    out = full[:, :, ::down_y, ::down_x]
    oh = upfirdn2d_out_size(h, up_y, down_y, py0, py1, kh)
    ow = upfirdn2d_out_size(w, up_x, down_x, px0, px1, kw)
    assert out.shape[-2:] == (oh, ow), (out.shape, oh, ow)
    return out.reshape(n, c, oh, ow).contiguous()


def upfirdn2d_loops(x, kernel, up, down, pad):
    """Independent scalar definition (numpy float64 loops) used to cross-check :func:`upfirdn2d` on
    tiny cases: out[Y,X] = sum_{i,j} k[i,j] * z[Y*down + kh-1-i, X*down + kw-1-j] where z is the
    zero-stuffed, padded plane (op/upfirdn2d_kernel.cu:49-105, the un-tiled gather kernel)."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(kernel, dtype=np.float64)
    n, c, h, w = x.shape
    kh, kw = k.shape
    p0, p1 = pad
    oh = upfirdn2d_out_size(h, up, down, p0, p1, kh)
    ow = upfirdn2d_out_size(w, up, down, p0, p1, kw)
    out = np.zeros((n, c, oh, ow))
    for oy in range(oh):
        for ox in range(ow):
            acc = np.zeros((n, c))
            for i in range(kh):
                for j in range(kw):
                    zy = oy * down + (kh - 1 - i) - p0
                    zx = ox * down + (kw - 1 - j) - p0
                    if zy < 0 or zx < 0 or zy % up or zx % up:
                        continue
                    iy, ix = zy // up, zx // up
                    if iy >= h or ix >= w:
                        continue
                    acc += k[i, j] * x[:, :, iy, ix]
            out[:, :, oy, ox] = acc
    return out


def fused_leaky_relu(x, bias, negative_slope=0.2, scale=2 ** 0.5):
    """leaky_relu(x + bias[channel]) * scale with the channel on dim 1 (op/fused_act.py:86-97).

    The CUDA kernel honours ``negative_slope`` (fused_bias_act_kernel.cu:38); the reference's CPU branch
    hard-codes 0.2 (op/fused_act.py:91).  They agree at the only value the generator uses (0.2); the
    oracle follows the CUDA kernel.
    """
    shape = [1, bias.shape[0]] + [1] * (x.ndim - 2)
    y = x + bias.reshape(shape)
    return torch.where(y > 0, y, y * negative_slope) * scale


def fused_bias_act_kernel_semantics(x, bias, refer, act, grad, alpha, scale):
    """Full act*10+grad switch of op/fused_bias_act_kernel.cu:18-49 on flat data (numpy).  bias index
    is (flat_index // step_b) % size_b with step_b = prod(x.shape[2:]) (:67-71)."""
    x = np.asarray(x, dtype=np.float32)
    flat = x.reshape(-1).copy()
    if bias is not None and bias.size:
        step_b = int(np.prod(x.shape[2:])) if x.ndim > 2 else 1
        idx = (np.arange(flat.size) // step_b) % bias.size
        flat = flat + np.asarray(bias, dtype=np.float32)[idx]
    ref = np.asarray(refer, dtype=np.float32).reshape(-1) if refer is not None and np.size(refer) else np.zeros_like(flat)
    code = act * 10 + grad
    if code == 10:
        y = flat
    elif code == 11:
        y = flat
    elif code == 12:
        y = np.zeros_like(flat)
    elif code == 30:
        y = np.where(flat > 0, flat, flat * np.float32(alpha))
    elif code == 31:
        y = np.where(ref > 0, flat, flat * np.float32(alpha))
    elif code == 32:
        y = np.zeros_like(flat)
    else:  # `default:` falls into case 10 (fused_bias_act_kernel.cu:37-38)
        y = flat
    return (y * np.float32(scale)).astype(np.float32).reshape(x.shape)
