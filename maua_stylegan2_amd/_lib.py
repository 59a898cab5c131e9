"""ctypes binding of csrc/libmaua_hip.so — the C ABI declared in include/maua_hip.h.

The product path has NO CPU or PyTorch fallback: if the library is missing, or a launcher returns non-zero,
this raises.  torch is used only for device memory (``tensor.data_ptr()``) and the current HIP stream.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmaua_hip.so")
ABI_VERSION = 5


class MauaHipError(RuntimeError):
    pass


class StyleLayer(ctypes.Structure):
    """maua_style_layer_t (include/maua_hip.h)."""

    _fields_ = [
        ("mod_w", c_void_p),
        ("mod_b", c_void_p),
        ("wsq", c_void_p),
        ("cin", c_int),
        ("cout", c_int),
        ("lat_idx", c_int),
        ("s_off", c_int),
        ("d_off", c_int64),
        ("wscale", c_float),
        ("pad_", c_int),
    ]


MAX_NOISE_SLOTS = 32


class FrameSource(ctypes.Structure):
    """maua_frame_source_t (include/maua_hip.h): per-frame sequences resident in HBM + the frame a launch starts at."""

    _fields_ = [
        ("frame0", ctypes.c_int32),
        ("pad_", ctypes.c_int32),
        ("latents", c_void_p),
        ("trunc", c_void_p),
        ("noise", c_void_p * MAX_NOISE_SLOTS),
        ("noise_stride", c_int64 * MAX_NOISE_SLOTS),
    ]


_P = c_void_p
_SIGNATURES = {
    "maua_abi_version": (c_int, []),
    "maua_device_info": (c_int, [POINTER(c_int), POINTER(c_int), c_char_p, c_int]),
    "maua_upfirdn2d_f32": (c_int, [_P, _P, _P] + [c_int] * 14 + [_P]),
    "maua_fused_bias_act_f32": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "maua_fused_bias_act_f16": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "maua_fused_bias_act_f64": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "maua_upfirdn2d_f16": (c_int, [_P, _P, _P] + [c_int] * 14 + [_P]),
    "maua_upfirdn2d_f64": (c_int, [_P, _P, _P] + [c_int] * 14 + [_P]),
    "maua_frame_source_seek": (c_int, [_P, c_int, _P]),
    "maua_blur_noise_act_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P, _P, c_int64, _P, _P, _P, c_int, _P, c_int, _P]),
    "maua_upconv_blur_ok": (c_int, [c_int] * 4),
    "maua_upconv_blur_ws_floats": (c_int64, [c_int] * 5),
    "maua_upconv_blur_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, c_int] + [c_int] * 5 + [c_float, _P, _P]),
    "maua_lowres_ok": (c_int, [c_int] * 5),
    "maua_lowres_ws_floats": (c_int64, [c_int] * 6),
    "maua_upconv_blur_lowres_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int64, _P, _P, _P, c_int] + [c_int] * 6 + [c_float, _P, _P]),
    "maua_styledconv_rgbpart_lowres_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, _P, c_int]
                                           + [c_int] * 6 + [c_float, _P]),
    "maua_const_conv_ok": (c_int, [c_int] * 4),
    "maua_pack_const_conv_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "maua_const_styledconv_f32": (c_int, [_P, _P, c_int, _P, _P, _P, c_int64, _P, _P, _P, _P, c_float, _P, _P, c_int] + [c_int] * 5 + [c_float, _P]),
    "maua_style_affine_f32": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, c_int, c_int, _P, c_int, _P, _P]),
    "maua_demod_f32": (c_int, [_P, c_int, c_int, _P, c_int, _P, c_int, _P]),
    "maua_pack_weight_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "maua_pack_weight_wino_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_pack_weight_wino43_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_pack_weight_upwino_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_pack_weight_wino2d_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_pack_weight_up2d_floats": (c_int64, [c_int] * 2),
    "maua_pack_weight_up2d_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_modconv_up2d_ok": (c_int, [c_int] * 4),
    "maua_pack_weight_sbf16_bytes": (c_int64, [c_int] * 2),
    "maua_pack_weight_sbf16_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "maua_modconv_sbf16_ok": (c_int, [c_int] * 4),
    "maua_modconv_sbf16_up_ok": (c_int, [c_int] * 4),
    "maua_modconv_w2d_ok": (c_int, [c_int] * 4),
    "maua_modconv_w2d_mtiles": (c_int, [c_int] * 4),
    "maua_modconv_ws_floats": (c_int64, [c_int] * 6),
    "maua_modconv_last_instance": (c_int, [c_char_p, c_int]),
    "maua_modconv3x3_f32": (c_int, [_P, _P, _P, c_int, _P, _P] + [c_int] * 6 + [c_float, c_int, _P, c_int64, _P, _P, _P, _P, c_int, _P]),
    "maua_styledconv_torgb_f32": (c_int, [_P, _P, _P, c_int, _P, _P] + [c_int] * 6 + [c_float, _P, c_int64, _P, _P, _P, _P, c_float,
                                          _P, _P, _P, _P, c_int, _P, _P, c_int, _P, _P]),
    "maua_styledconv_torgb_partial_f32": (c_int, [_P, _P, _P, c_int, _P, _P] + [c_int] * 6 + [c_float, _P, c_int64, _P, _P, _P, _P, c_float,
                                                  _P, _P, c_int, _P, _P]),
    "maua_torgb_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "maua_frames_to_u8": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "maua_crop_resize_u8": (c_int, [_P, _P] + [c_int] * 9 + [_P]),
    "maua_sg1_epilogue_f32": (c_int, [_P, _P, _P, c_int64, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "maua_temporal_fir_f32": (c_int, [_P, _P, _P, c_int, c_int64, c_int, _P]),
    "maua_stft_power_f32": (c_int, [_P, c_int64, _P, c_int, c_int, _P, c_int, _P]),
    "maua_stft_complex_f32": (c_int, [_P, c_int64, _P, c_int, c_int, _P, _P, c_int, _P]),
    "maua_istft_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int64, _P]),
    "maua_median_filter_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "maua_softmask_apply_f32": (c_int, [_P, _P, _P, _P, c_float, c_float, c_int, _P, _P, c_int64, _P]),
    "maua_resample_f64": (c_int, [_P, c_int, c_int64, _P, c_int, _P]),
    "maua_cqt_mag_f32": (c_int, [_P, c_int64, _P, _P, c_int, c_int, c_float, _P, c_int, _P]),
    "maua_chroma_cens_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "maua_nn_median_ws_doubles": (c_int64, [c_int] * 3),
    "maua_nn_median_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "maua_filterbank_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "maua_perlin3d_f32": (c_int, [_P, _P] + [c_int] * 6 + [_P]),
    "maua_affine_reflect_warp_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P, _P]),
    "maua_affine_reflect_warp_mapped_f32": (c_int, [_P, _P, _P] + [c_int] * 8 + [_P, _P, _P, _P, _P]),
    "maua_graph_begin_capture": (c_int, [_P]),
    "maua_graph_end_capture": (c_int, [_P, POINTER(c_void_p)]),
    "maua_graph_launch": (c_int, [_P, _P]),
    "maua_graph_destroy": (c_int, [_P]),
    "maua_event_create": (c_int, [POINTER(c_void_p)]),
    "maua_event_record": (c_int, [_P, _P]),
    "maua_event_elapsed_ms": (c_int, [_P, _P, POINTER(c_float)]),
    "maua_event_destroy": (c_int, [_P]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """dlopen the library (once) and type every entry point; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MauaHipError(
            f"{LIB_PATH} is missing: build it with `python -m maua_stylegan2_amd.build` "
            "(there is deliberately no CPU / PyTorch fallback for the native ops)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    ver = lib.maua_abi_version()
    if ver != ABI_VERSION:
        raise MauaHipError(f"libmaua_hip.so ABI {ver} != binding ABI {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise MauaHipError(f"{what} failed with code {rc}" + (" (hipError)" if rc > 0 else " (rejected arguments)"))


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def require_cuda_any(t, name):
    """The two native ops accept half / float / double like the reference's dtype dispatch; everything else is fp32."""
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor")  # mirrors CHECK_CUDA, op/upfirdn2d.cpp:7
    if t.dtype not in (torch.float16, torch.float32, torch.float64):
        raise RuntimeError(f"{name} must be float16, float32 or float64 (got {t.dtype})")
    return t.contiguous()


DTYPE_SUFFIX = {torch.float16: "f16", torch.float32: "f32", torch.float64: "f64"}


def require_cuda(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA (HIP) tensor")  # mirrors CHECK_CUDA, op/upfirdn2d.cpp:7
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32 (got {t.dtype}); the MI355X path computes in fp32")
    return t.contiguous()


def device_info():
    lib = load()
    cu, lds = c_int(0), c_int(0)
    buf = ctypes.create_string_buffer(256)
    check(lib.maua_device_info(ctypes.byref(cu), ctypes.byref(lds), buf, 256), "maua_device_info")
    return {"cu_count": cu.value, "lds_bytes": lds.value, "name": buf.value.decode()}


def last_modconv_instance():
    """rocprofv3 name of the kernel instance the last modulated-conv call launched (key of profiles/*_pmc_traffic.json)."""
    buf = ctypes.create_string_buffer(128)
    check(load().maua_modconv_last_instance(buf, 128), "maua_modconv_last_instance")
    return buf.value.decode()


class HipEvent:
    """HIP event on an explicit stream (bench.py roofline timing)."""

    def __init__(self):
        self._h = c_void_p()
        check(load().maua_event_create(ctypes.byref(self._h)), "maua_event_create")

    def record(self, stream=None):
        check(load().maua_event_record(self._h, stream if stream is not None else stream_ptr()), "maua_event_record")

    def elapsed_ms(self, end):
        ms = c_float(0)
        check(load().maua_event_elapsed_ms(self._h, end._h, ctypes.byref(ms)), "maua_event_elapsed_ms")
        return ms.value

    def __del__(self):
        try:
            if self._h:
                load().maua_event_destroy(self._h)
        except Exception:
            pass


class HipGraph:
    """Capture-and-replay of everything launched on the current stream inside the ``with`` block."""

    def __init__(self):
        self._exec = c_void_p()
        self._stream = None

    def __enter__(self):
        self._stream = stream_ptr()
        check(load().maua_graph_begin_capture(self._stream), "maua_graph_begin_capture")
        return self

    def __exit__(self, et, ev, tb):
        rc = load().maua_graph_end_capture(self._stream, ctypes.byref(self._exec))
        if et is None:
            check(rc, "maua_graph_end_capture")
        return False

    def replay(self, stream=None):
        check(load().maua_graph_launch(self._exec, stream if stream is not None else stream_ptr()), "maua_graph_launch")

    def __del__(self):
        try:
            if self._exec:
                load().maua_graph_destroy(self._exec)
        except Exception:
            pass
