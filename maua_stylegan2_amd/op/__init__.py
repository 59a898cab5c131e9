"""Drop-in for the reference's ``op`` package (/root/reference/op/__init__.py:1-2): same three names."""
from .fused_act import FusedLeakyReLU, fused_bias_act, fused_leaky_relu
from .upfirdn2d import upfirdn2d, upfirdn2d_native_op

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "fused_bias_act", "upfirdn2d", "upfirdn2d_native_op"]
