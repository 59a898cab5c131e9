"""StyleGAN2 generator (inference) for MI355X — host-side mirror of /root/reference/models/stylegan2.py:1-576.

Same class names, constructor arguments, ``forward`` signatures and state-dict keys as the reference (so
``th.load(ckpt)["g_ema"]`` drops in, models/stylegan2.py:458-459), but the forward is a different program:

  reference (per StyledConv)                          here
  ------------------------------------------------    ---------------------------------------------------------
  F.linear, mul, pow/sum/rsqrt, mul  (per layer)      2 launches for ALL layers: maua_style_affine_f32, maua_demod_f32
  grouped conv with per-sample weights (cuDNN)        shared-weight implicit GEMM on fp32 MFMA, input-scale/output-demod
  add noise, fused_bias_act                           fused into the conv epilogue (plain) / the blur kernel (upsample)
  conv_transpose2d + upfirdn2d (Blur)                 polyphase MFMA transposed conv + fir_tile_kernel with fused tail
  ToRGB: linear, mul, conv, add, upfirdn2d, add       1 launch (maua_torgb_f32)

All activations live in a per-batch-size cache of static device buffers, so a forward performs no allocation after
its first call and can be captured into a hipGraph (``capture_graph``).  There is no CPU path: CPU tensors raise.
"""
import ctypes
import math

import torch as th
from torch import nn
from torch.nn import functional as F

from .. import _lib
from ..op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d


class PixelNorm(nn.Module):
    def forward(self, inputs):
        return inputs * th.rsqrt(th.mean(inputs ** 2, dim=1, keepdim=True) + 1e-8)


def make_kernel(k):
    k = th.tensor(k, dtype=th.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class Upsample(nn.Module):
    """reference :34-52 — upfirdn2d(up=factor, pad=(pad0, pad1)) with kernel * factor**2."""

    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        self.register_buffer("kernel", make_kernel(kernel) * (factor ** 2))
        p = self.kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, inputs):
        return upfirdn2d(inputs, self.kernel, up=self.factor, down=1, pad=self.pad)


class Blur(nn.Module):
    """reference :76-92."""

    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad

    def forward(self, inputs):
        return upfirdn2d(inputs, self.kernel, pad=self.pad)


import threading  # noqa: E402

_SKIP_INIT = threading.local()  # .on: set only on the thread that is building a Generator from a checkpoint


def _randn(*shape):
    """``th.randn`` for a parameter / buffer of a module under construction.  While a ``Generator`` that was given a checkpoint builds
    itself, every such tensor is about to be overwritten by the strict ``load_state_dict`` of its constructor: it is then left
    uninitialised (the 30 M normal draws of a 1024^2 generator cost 0.4 s of host time, a quarter of a warm generate() of 900 frames;
    the reference draws them and throws them away, models/stylegan2.py:455-459 — the only observable difference is that the CPU
    generator's state is not advanced by the construction)."""
    return th.empty(*shape) if getattr(_SKIP_INIT, "on", False) else th.randn(*shape)


class EqualLinear(nn.Module):
    """reference :123-146 (mapping network / standalone use; the per-layer modulations inside the generator forward
    go through the table-driven style kernel instead)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(_randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(th.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, inputs):
        if self.activation:
            out = F.linear(inputs, self.weight * self.scale)
            return fused_leaky_relu(out, self.bias * self.lr_mul)
        return F.linear(inputs, self.weight * self.scale, bias=self.bias * self.lr_mul)


def _style_table(entries, device):
    """entries: list of dict(mod_w, mod_b, wsq|None, cin, cout, lat_idx, s_off, d_off, wscale) -> device byte tensor."""
    arr = (_lib.StyleLayer * len(entries))()
    for i, e in enumerate(entries):
        arr[i].mod_w = e["mod_w"].data_ptr()
        arr[i].mod_b = e["mod_b"].data_ptr()
        arr[i].wsq = e["wsq"].data_ptr() if e["wsq"] is not None else None
        arr[i].cin, arr[i].cout, arr[i].lat_idx = e["cin"], e["cout"], e["lat_idx"]
        arr[i].s_off, arr[i].d_off, arr[i].wscale = e["s_off"], e["d_off"], e["wscale"]
    raw = th.frombuffer(bytearray(bytes(arr)), dtype=th.uint8).clone()
    return raw.to(device)


class ModulatedConv2d(nn.Module):
    """reference :164-254.  ``forward(inputs, style)`` keeps the reference contract; the generator calls
    ``run(...)`` with styles / demod factors already computed for all layers."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False,
                 downsample=False, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if downsample:
            raise NotImplementedError("downsample=True is discriminator-only (SURVEY.md §2 row 4: out of scope)")
        if kernel_size not in (1, 3):
            raise NotImplementedError("the generator only uses 3x3 and 1x1 modulated convolutions")
        self.eps = 1e-8
        self.kernel_size = kernel_size
        self.in_channel = in_channel
        self.out_channel = out_channel
        self.upsample = upsample
        self.downsample = downsample
        if upsample:
            factor = 2
            p = (len(blur_kernel) - factor) - (kernel_size - 1)
            self.blur = Blur(blur_kernel, pad=((p + 1) // 2 + factor - 1, p // 2 + 1), upsample_factor=factor)
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = kernel_size // 2
        self.weight = nn.Parameter(_randn(1, out_channel, in_channel, kernel_size, kernel_size))
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.demodulate = demodulate
        self._packed = None  # (key, wp, wsq)
        self._packed_wino = None

    def __repr__(self):
        return (f"{self.__class__.__name__}({self.in_channel}, {self.out_channel}, {self.kernel_size}, "
                f"upsample={self.upsample}, downsample={self.downsample})")

    def packed(self):
        """Tap-major repack + per-(o,i) squared-tap sums of the shared weight; rebuilt when the parameter changes
        (model rewriting, render.py:160-167, swaps the Parameter object)."""
        w = self.weight
        key = (w.data_ptr(), w._version, str(w.device))
        if self._packed is None or self._packed[0] != key:
            wd = _lib.require_cuda(w.detach(), "weight")
            k2 = self.kernel_size ** 2
            cpad = (self.out_channel + 31) // 32 * 32
            wp = th.empty((k2, self.in_channel, cpad), dtype=th.float32, device=w.device) if k2 == 9 else None
            wsq = th.empty((self.out_channel, self.in_channel), dtype=th.float32, device=w.device)
            with th.cuda.device(w.device):
                rc = _lib.load().maua_pack_weight_f32(wd.data_ptr(), _lib.ptr(wp), wsq.data_ptr(), self.out_channel,
                                                      self.in_channel, k2, _lib.stream_ptr(w.device))
            _lib.check(rc, "maua_pack_weight_f32")
            self._packed = (key, wp, wsq)
            self._packed_wino = None
        return self._packed[1], self._packed[2]

    # plain 3x3 layers with at least this many output channels run through Winograd F(2,3) (MFMA-bound layers);
    # a huge value turns it off
    winograd_min_cout = 32
    # ... on the maps below 32 x 32 from this many output channels (a huge value turns it off there)
    winograd_small_min_cout = 128
    # ... and from this many output channels, on maps at least this wide, through F(4,3) (6 products per 4 outputs)
    winograd43_min_cout = 32
    winograd43_min_width = 32
    # transposed layers: F(2,2) on the even x-phase (mode 4) where the launch is large enough, see conv_mode
    upconv_winograd = True
    # plain layers whose shape the 2-D Winograd kernel accepts (csrc/modconv_w2d.hip, mode 5: F(2,3) along y on top of
    # F(4,3) along x, 3 instead of 4.5 MFMA products per output) with at least this many output channels.  Measured inside
    # bench.py (profiles/r02_w2d.md): -11..24 % per launch on the 128..512-channel layers; -14 % on the 64-channel and -1 % on
    # the 32-channel layer with the fused ToRGB epilogue (since the skip-image taps of that epilogue are unconditional loads),
    # +1.8 % frames/s for the whole generator.
    winograd2d_min_cout = 32
    # transposed layers whose shape csrc/modconv_up2d.hip accepts (mode 6: F(2,2) on both axes of the polyphase form, 25 instead of
    # 30 (mode 4) / 36 (mode 1) products per 2x2 positions) with at least this many output channels; a huge value turns it off
    upwino2d_min_cout = 32
    # SIDE MEASUREMENT, off by default (a huge value): plain layers with at least this many output channels whose shape
    # csrc/modconv_sbf16.hip accepts run the direct 9-tap form with split-bf16 products on the bf16 matrix cores (mode 7) instead of the
    # fp32 2-D Winograd kernel.  bench.py lowers it for its `split_bf16` side figure; the headline path computes in fp32.
    split_bf16_min_cout = 1 << 30
    # ... and the same for the transposed layers (mode 8: the four polyphase output phases as four launches of the same kernel + the
    # fp32 edge lines)
    split_bf16_up_min_cout = 1 << 30

    def conv_mode(self, h, w):
        """Kernel mode of maua_modconv3x3_f32 for an [*, Cin, h, w] input: 1 / 4 / 6 transposed, 2 Winograd F(2,3), 3 Winograd
        F(4,3), 5 2-D Winograd F(2x4,3x3), 0 direct; 7 / 8 = the split-bf16 side measurement (plain / transposed), off by default."""
        if self.upsample and self.out_channel >= self.split_bf16_up_min_cout and _lib.load().maua_modconv_sbf16_up_ok(
                self.in_channel, self.out_channel, h, w):
            return 8
        if self.upsample and self.out_channel >= self.upwino2d_min_cout and _lib.load().maua_modconv_up2d_ok(
                self.in_channel, self.out_channel, h, w):
            return 6
        if self.upsample:
            # F(2,2) on the even x-phase of the polyphase transposed conv (mode 4: -17 % MFMA work, but 2 instead of 3-4
            # workgroups per CU) once a batch of 8 frames yields at least ~4 rounds of workgroups; smaller grids lose more
            # to the partially filled last round than they gain and keep the plain polyphase kernel (mode 1)
            pairs_per_tile, rows_per_tile = (128, 32) if self.out_channel <= 32 else (64, 64)
            tiles = -(-((h + 1) * (w // 2 + 1)) // pairs_per_tile) * -(-self.out_channel // rows_per_tile)
            if self.upconv_winograd and w % 2 == 0 and tiles >= 256:
                return 4
            return 1
        if self.out_channel >= self.split_bf16_min_cout and _lib.load().maua_modconv_sbf16_ok(self.in_channel, self.out_channel, h, w):
            return 7
        # (the Winograd kernels want at least one full 128-position tile per image; shorter maps would pack several images
        # into a tile, which only the direct mode implements)
        if (self.out_channel >= self.winograd2d_min_cout
                and _lib.load().maua_modconv_w2d_ok(self.in_channel, self.out_channel, h, w)):
            return 5
        if (self.out_channel >= self.winograd43_min_cout and w % 4 == 0 and w >= self.winograd43_min_width
                and h * (w // 4) >= 128):
            return 3
        if self.out_channel >= self.winograd_min_cout and w % 2 == 0 and w >= 32 and h * (w // 2) >= 128:
            return 2
        # the 8^2 / 16^2 layers of a generator (128 and more channels: the 128 x 64 Winograd tile, several images per tile at 8^2): F(2,3) as well —
        # 50 -> 41 us at 8^2, 103 -> 75 us at 16^2 for 512 channels at batch 8 (tools/plain_mode_probe.py, round 6)
        if self.out_channel >= self.winograd_small_min_cout and w % 2 == 0 and 8 <= w < 32 and h >= 8:
            return 2
        return 0

    def blur_is_separable(self):
        """True when the Blur's tap matrix is 4 x 4 and an outer product (make_kernel of 1-D taps always is; a checkpoint could carry
        anything): what maua_upconv_blur_f32 requires.  One host read per (buffer, version), made by the eager warm-up forward — never
        inside a capture."""
        k = self.blur.kernel
        key = (k.data_ptr(), k._version, str(k.device))
        cached = self.__dict__.get("_blur_sep")
        if cached is None or cached[0] != key:
            kk = k.detach().double().cpu()
            ok = tuple(kk.shape) == (4, 4) and float(kk.sum()) != 0.0
            if ok:
                outer = kk.sum(1)[:, None] * kk.sum(0)[None, :] / kk.sum()
                ok = bool((outer - kk).abs().max() <= 1e-6 * kk.abs().max())
            self.__dict__["_blur_sep"] = cached = (key, ok)
        return cached[1]

    def packed_wino(self, mode=2):
        """Winograd-domain weight [(ky*F+xi), Cin, Cout_pad], F = 4 for mode 2 (maua_pack_weight_wino_f32) and mode 4
        (maua_pack_weight_upwino_f32), 6 for mode 3 (maua_pack_weight_wino43_f32); cached like ``packed()``."""
        self.packed()  # refreshes / invalidates on weight change
        if self._packed_wino is None:
            self._packed_wino = {}
        if mode == 6 and mode not in self._packed_wino:
            w = self.weight
            wd = _lib.require_cuda(w.detach(), "weight")
            wq = th.empty(_lib.load().maua_pack_weight_up2d_floats(self.out_channel, self.in_channel), dtype=th.float32,
                          device=w.device)
            with th.cuda.device(w.device):
                _lib.check(_lib.load().maua_pack_weight_up2d_f32(wd.data_ptr(), wq.data_ptr(), self.out_channel,
                                                                 self.in_channel, _lib.stream_ptr(w.device)),
                           "maua_pack_weight_up2d_f32")
            self._packed_wino[mode] = wq
        if mode == 8:
            mode = 7  # (one packed weight serves the plain and the transposed form)
        if mode == 7 and mode not in self._packed_wino:
            w = self.weight
            wd = _lib.require_cuda(w.detach(), "weight")
            wq = th.empty(_lib.load().maua_pack_weight_sbf16_bytes(self.out_channel, self.in_channel), dtype=th.uint8, device=w.device)
            with th.cuda.device(w.device):
                _lib.check(_lib.load().maua_pack_weight_sbf16_f32(wd.data_ptr(), wq.data_ptr(), self.out_channel, self.in_channel,
                                                                  _lib.stream_ptr(w.device)), "maua_pack_weight_sbf16_f32")
            self._packed_wino[mode] = wq
        if mode == 5 and mode not in self._packed_wino:
            w = self.weight
            wd = _lib.require_cuda(w.detach(), "weight")
            wq = th.empty(24 * self.in_channel * self.out_channel, dtype=th.float32, device=w.device)
            with th.cuda.device(w.device):
                _lib.check(_lib.load().maua_pack_weight_wino2d_f32(wd.data_ptr(), wq.data_ptr(), self.out_channel,
                                                                   self.in_channel, _lib.stream_ptr(w.device)),
                           "maua_pack_weight_wino2d_f32")
            self._packed_wino[mode] = wq
        if mode not in self._packed_wino:
            w = self.weight
            wd = _lib.require_cuda(w.detach(), "weight")
            # modes 2 / 3 above 32 channels: columns padded to 64 and interleaved [lane][m-tile] (see pack_weight_wino_kernel)
            pad = 64 if (mode in (2, 3) and self.out_channel > 32) else 32
            cpad = (self.out_channel + pad - 1) // pad * pad
            wq = th.empty((18 if mode == 3 else 12, self.in_channel, cpad), dtype=th.float32, device=w.device)
            fn = {2: "maua_pack_weight_wino_f32", 3: "maua_pack_weight_wino43_f32", 4: "maua_pack_weight_upwino_f32"}[mode]
            with th.cuda.device(w.device):
                rc = getattr(_lib.load(), fn)(wd.data_ptr(), wq.data_ptr(), self.out_channel, self.in_channel,
                                              _lib.stream_ptr(w.device))
            _lib.check(rc, fn)
            self._packed_wino[mode] = wq
        return self._packed_wino[mode]

    def table_entry(self, lat_idx, s_off, d_off):
        wp, wsq = self.packed()
        return dict(mod_w=self.modulation.weight, mod_b=self.modulation.bias, wsq=wsq if self.demodulate else None,
                    cin=self.in_channel, cout=self.out_channel, lat_idx=lat_idx, s_off=s_off, d_off=d_off,
                    wscale=self.scale)

    def run(self, x, s, s_off, d, out, ws, fuse_act=False, noise=None, noise_w=None, bias=None, src=None, slot=0, prescaled=False):
        """3x3 only. x [B,Cin,H,W]; s [B,S] (this layer's slice at s_off); d [B,Cout] or None.
        Writes ``out`` ([B,Cout,H,W] or [B,Cout,2H+1,2W+1] when upsample).  ``src`` (device pointer of a frame source,
        include/maua_hip.h): the noise map comes from its slot ``slot`` instead of ``noise``.  ``prescaled``: x arrives multiplied by
        this layer's styles (the style fold, include/maua_hip.h; modes 5 and 6 only)."""
        lib = _lib.load()
        b, cin, h, w = x.shape
        mode = self.conv_mode(h, w)
        wp = self.packed_wino(mode) if mode >= 2 else self.packed()[0]
        nstride = 0 if noise is None or noise.shape[0] == 1 else noise.shape[-1] * noise.shape[-2]
        rc = lib.maua_modconv3x3_f32(
            x.data_ptr(), wp.data_ptr(), None if prescaled else s.data_ptr() + 4 * s_off, s.shape[1], _lib.ptr(d), out.data_ptr(), b, cin,
            self.out_channel, h, w, mode, float(self.scale), int(fuse_act), _lib.ptr(noise), nstride,
            _lib.ptr(noise_w), _lib.ptr(bias), _lib.ptr(ws), src if fuse_act else None, slot, _lib.stream_ptr(x.device),
        )
        _lib.check(rc, "maua_modconv3x3_f32")
        return out

    def forward(self, inputs, style):
        x = _lib.require_cuda(inputs, "inputs")
        style = _lib.require_cuda(style, "style")
        b, cin, h, w = x.shape
        dev = x.device
        lib = _lib.load()
        with th.cuda.device(dev):
            s = th.empty((b, cin), dtype=th.float32, device=dev)
            d = th.empty((b, self.out_channel), dtype=th.float32, device=dev) if self.demodulate else None
            table = _style_table([self.table_entry(0, 0, 0)], dev)
            lat = style.reshape(b, 1, -1)
            st = _lib.stream_ptr(dev)
            _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), b, 1, lat.shape[-1], None, None, table.data_ptr(), 1,
                                                 cin, s.data_ptr(), cin, None, st), "maua_style_affine_f32")
            if d is not None:
                _lib.check(lib.maua_demod_f32(table.data_ptr(), 1, self.out_channel, s.data_ptr(), cin, d.data_ptr(), b,
                                              st), "maua_demod_f32")
            if self.kernel_size == 1:
                if self.out_channel != 3 or self.demodulate:
                    raise NotImplementedError("1x1 modulated conv is only built for ToRGB (3 channels, no demodulation)")
                out = th.empty((b, 3, h, w), dtype=th.float32, device=dev)
                _lib.check(lib.maua_torgb_f32(x.data_ptr(), self.weight.data_ptr(), s.data_ptr(), cin, None, None, None,
                                              out.data_ptr(), b, cin, h, w, float(self.scale), st), "maua_torgb_f32")
                return out
            oh, ow = (2 * h + 1, 2 * w + 1) if self.upsample else (h, w)
            out = th.empty((b, self.out_channel, oh, ow), dtype=th.float32, device=dev)
            n_ws = lib.maua_modconv_ws_floats(b, cin, self.out_channel, h, w, self.conv_mode(h, w))
            ws = th.empty(n_ws, dtype=th.float32, device=dev) if n_ws else None
            self.run(x, s, 0, d, out, ws)
            if self.upsample:
                out = self.blur(out)
        return out


class NoiseInjection(nn.Module):
    """reference :257-266 (standalone; inside StyledConv the add is fused into the conv / blur epilogue)."""

    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(th.zeros(1))

    def forward(self, image, noise=None):
        if noise is None:
            batch, _, height, width = image.shape
            noise = image.new_empty(batch, 1, height, width).normal_()
        return image + self.weight * noise.to(image.device)


class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(_randn(1, channel, size, size))

    def forward(self, inputs):
        return self.input.repeat(inputs.shape[0], 1, 1, 1)


class LatentInput(nn.Module):
    """reference :281-294 (``--noconst`` checkpoints): the 4x4 input is an affine function of the first latent row,
    ``activate(fused_lrelu(EqualLinear(latent[:, 0])))`` reshaped to [B, C, 4, 4].  State-dict keys as the reference:
    input.linear.{weight,bias}, input.activate.bias, input.input (an unused scalar Parameter that only carries the device)."""

    def __init__(self, latent_dim, channel, size=4):
        super().__init__()
        self.channel = channel
        self.size = size
        self.linear = EqualLinear(latent_dim, channel * size * size, activation="fused_lrelu")
        self.activate = FusedLeakyReLU(channel * size * size)
        self.input = nn.Parameter(th.randn(1))

    def forward(self, inputs):
        out = self.activate(self.linear(inputs[:, 0]))
        return out.reshape((inputs.shape[0], self.channel, self.size, self.size))

    def run(self, latent, trunc, tl, out, src=None, n_latent=None, style_dim=None):
        """Device path on static buffers (capturable): the table-driven affine kernel (truncation lerp included) writes
        W x / sqrt(dim) + b into ``out`` [B, C*16]; two in-place bias-act launches apply lrelu*sqrt2 and the second
        bias + lrelu*sqrt2.  ``src``: the latents / truncation come from the frame source instead of ``latent`` / ``trunc``."""
        lib = _lib.load()
        dev = out.device
        n_out = self.channel * self.size * self.size
        batch = out.shape[0]
        if latent is not None:
            n_latent, style_dim = latent.shape[1], latent.shape[2]
        key = (self.linear.weight.data_ptr(), self.linear.bias.data_ptr(), str(dev))
        if getattr(self, "_table", None) is None or self._table[0] != key:
            entry = dict(mod_w=self.linear.weight, mod_b=self.linear.bias, wsq=None, cin=n_out, cout=0, lat_idx=0, s_off=0,
                         d_off=0, wscale=1.0)
            self._table = (key, _style_table([entry], dev))
        st = _lib.stream_ptr(dev)
        flat = out.view(batch, n_out)
        _lib.check(lib.maua_style_affine_f32(_lib.ptr(latent), batch, n_latent, style_dim, _lib.ptr(trunc),
                                             _lib.ptr(tl), self._table[1].data_ptr(), 1, n_out, flat.data_ptr(), n_out, src, st),
                   "maua_style_affine_f32")
        _lib.check(lib.maua_fused_bias_act_f32(flat.data_ptr(), None, None, flat.data_ptr(), flat.numel(), 0, 1, 3, 0, 0.2,
                                               2 ** 0.5, st), "maua_fused_bias_act_f32")
        _lib.check(lib.maua_fused_bias_act_f32(flat.data_ptr(), self.activate.bias.data_ptr(), None, flat.data_ptr(),
                                               flat.numel(), n_out, 1, 3, 0, self.activate.negative_slope,
                                               self.activate.scale, st), "maua_fused_bias_act_f32")
        return out


class ManipulationLayer(nn.Module):
    """reference :297-307 — applies every transform whose "layer" id matches."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, input, tranforms_dict_list):
        out = input
        for transform_dict in tranforms_dict_list:
            if transform_dict["layer"] == self.layer:
                out = transform_dict["transform"].to(out.device)(out)
        return out

    def run(self, x, bends, bufs, tag, src=None):
        """Device path.  Eager (``src`` None): as ``forward``.  Inside a captured forward every bend is a module with the
        ``run_static(x, out, src)`` protocol (audioreactive/bend.py): it writes into a static buffer and reads its per-frame
        parameters through the frame source, so one captured graph serves every batch of the render."""
        if src is None:
            return self.forward(x, bends)
        for i, bd in enumerate(bends):
            if bd["layer"] == self.layer:
                x = bd["transform"].run_static(x, bufs(f"{tag}.bend{i}", tuple(x.shape)), src)
        return x


class StyledConv(nn.Module):
    """reference :310-343: ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU -> ManipulationLayer."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, blur_kernel=[1, 3, 3, 1],
                 demodulate=True, layerID=-1):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = FusedLeakyReLU(out_channel)
        self.manipulation = ManipulationLayer(layerID)

    # layers wider than one weight tile on the 2-D Winograd kernel leave per-tile partial ToRGB sums (A/B switch)
    partial_rgb_fusion = True
    # up-sampling layers whose INPUT is at least this wide run transposed convolution + blur + noise + bias + activation as one kernel
    # (maua_upconv_blur_f32, round 5): convs.14 (0.82 ms against the pair's 1.01) and convs.12 (level alone, +0.8 % frames/s inside the
    # overlapped pipeline, where the raw map's round trip competes for HBM); below, the extra tiles of its overlapped tiling cost more than
    # that round trip saves (convs.10: 0.70 against 0.56 ms; profiles/r05_fused_upconv_blur.md).  A huge value = always the two-launch path.
    fused_blur_min_width = 256
    # the 4^2 .. 32^2 layers (split-K direct / polyphase kernels): the convolution leaves its split-K slabs and ONE more launch reduces them
    # together with what follows — blur + noise + bias + activation of an up-sampling layer (maua_upconv_blur_lowres_f32), tail + per-group
    # partial ToRGB sums of a plain one (maua_styledconv_rgbpart_lowres_f32) — instead of reduce, then blur / ToRGB (A/B switch)
    lowres_fusion = True
    # ... and the 16-wide up-sampling layer among them on the F(2,2)^2 kernel (16 x 16-position tiles, K split) instead of the polyphase one (A/B switch)
    lowres_up2d = True

    def accepts_prescaled(self, h, w):
        """True when this layer's convolution has a kernel instance without the style multiplies for an [*, Cin, h, w] input (the style
        fold, include/maua_hip.h: the 2-D Winograd kernels, mode 5, and the F(2,2)^2 transposed kernel, mode 6, fused with its blur or not)."""
        return self.conv.conv_mode(h, w) in (5, 6)

    def run(self, x, s, s_off, d, noise, bufs, tag, rgb=None, src=None, slot=0, prescaled=False, post_off=None):
        """Fused forward on precomputed styles. ``bufs(name, shape)`` hands out static device buffers.
        The style fold (include/maua_hip.h): ``prescaled`` = x arrives multiplied by this layer's styles (the producer applied them);
        ``post_off`` = offset inside ``s`` of the styles of the layer that consumes this layer's output: where this layer's path can, it
        stores its map multiplied by them and sets ``self.posted`` (the consumer is then run with ``prescaled``).
        ``rgb`` (plain layers only): dict(module=ToRGB, s_off, skip, out, store) — fold the following ToRGB into the conv
        epilogue when the layer qualifies; on success ``rgb["done"]`` is set and ``rgb["out"]`` holds the image.
        ``src`` / ``slot``: the noise map is read through the frame source (``noise`` is then ignored and may be None)."""
        lib = _lib.load()
        conv = self.conv
        b, cin, h, w = x.shape
        self.posted = False
        s_ptr = None if prescaled else s.data_ptr() + 4 * s_off
        post_ptr = None if post_off is None else s.data_ptr() + 4 * post_off
        n_ws = lib.maua_modconv_ws_floats(b, cin, conv.out_channel, h, w, conv.conv_mode(h, w))
        # one split-K workspace PER LAYER: a shared name would be re-allocated whenever the size changes, and a captured
        # hipGraph keeps writing through the pointer of the buffer that was freed
        low_mode = conv.conv_mode(h, w)
        low = (self.lowres_fusion and not prescaled and low_mode in ((1,) if conv.upsample else (0, 2, 3))
               and lib.maua_lowres_ok(cin, conv.out_channel, h, w, low_mode))
        # (up-sampling layers: 6 = the F(2,2)^2 kernel on 16-wide inputs where the shape allows it, else 1 = the polyphase kernel)
        low_up = 6 if (low and conv.upsample and self.lowres_up2d and lib.maua_lowres_ok(cin, conv.out_channel, h, w, 6)) else 1
        ws = bufs(tag + ".ws", (n_ws,)) if (n_ws and not low) else None
        if src is not None:
            noise = None
        if noise is not None:
            noise = _lib.require_cuda(noise, "noise")
            # the kernels read oh*ow floats per sample (b samples unless the map is shared): a wrongly sized map would be
            # a silent out-of-bounds read where the reference raises a broadcast error (models/stylegan2.py:266)
            oh, ow = (2 * h, 2 * w) if conv.upsample else (h, w)
            if noise.dim() != 4 or noise.shape[1] != 1 or tuple(noise.shape[-2:]) != (oh, ow) or noise.shape[0] not in (1, b):
                raise RuntimeError(f"noise {tuple(noise.shape)} does not match feature map [{b}, 1, {oh}, {ow}] "
                                   f"(batch must be 1 or {b})")
            noise = noise.contiguous()
        if not conv.upsample:
            out = bufs(tag, (b, conv.out_channel, h, w))
            if rgb is not None and conv.out_channel <= 64 and n_ws == 0:
                t = rgb["module"]
                skip = rgb["skip"]
                fusable = skip is None or (tuple(t.upsample.kernel.shape) == (4, 4) and t.upsample.factor == 2
                                           and skip.shape[2] * 2 == h and skip.shape[3] * 2 == w)
                if fusable:
                    mode = conv.conv_mode(h, w)
                    wp = conv.packed_wino(mode) if mode >= 2 else conv.packed()[0]
                    nstride = 0 if noise is None or noise.shape[0] == 1 else noise.shape[-1] * noise.shape[-2]
                    post = post_ptr if mode == 5 else None
                    rc = lib.maua_styledconv_torgb_f32(
                        x.data_ptr(), wp.data_ptr(), s_ptr, s.shape[1], _lib.ptr(d), out.data_ptr(), b,
                        cin, conv.out_channel, h, w, mode, float(conv.scale), _lib.ptr(noise), nstride,
                        self.noise.weight.data_ptr(), self.activate.bias.data_ptr(), t.conv.weight.data_ptr(),
                        s.data_ptr() + 4 * rgb["s_off"], float(t.conv.scale), t.bias.data_ptr(), _lib.ptr(skip),
                        _lib.ptr(t.upsample.kernel) if skip is not None else None,
                        rgb["out"].data_ptr() if (rgb.get("u8") is None or rgb.get("tap")) else None,
                        int(rgb.get("store", True)), _lib.ptr(rgb.get("u8")), src, slot, post, _lib.stream_ptr(x.device))
                    if rc == 0:
                        rgb["done"] = True
                        self.posted = post is not None
                        return out
                    if rc != -38:  # MAUA_ENOSYS = layer shape not fusable -> two launches below
                        _lib.check(rc, "maua_styledconv_torgb_f32")
            elif rgb is not None and self.partial_rgb_fusion and n_ws == 0 and conv.conv_mode(h, w) == 5:
                # wider layers on the 2-D Winograd kernel: every output-channel tile leaves its share of the ToRGB sum (3 planes);
                # the ToRGB pass then adds 3 * m_tiles planes (+ bias, + up-sampled skip) instead of reading all `cout` feature planes
                t = rgb["module"]
                skip = rgb["skip"]
                m_tiles = lib.maua_modconv_w2d_mtiles(cin, conv.out_channel, h, w)
                fusable = m_tiles > 1 and (skip is None or (tuple(t.upsample.kernel.shape) == (4, 4) and t.upsample.factor == 2
                                                            and skip.shape[2] * 2 == h and skip.shape[3] * 2 == w))
                if fusable:
                    part = bufs(tag + ".rgb_partial", (b, 3 * m_tiles, h, w))
                    nstride = 0 if noise is None or noise.shape[0] == 1 else noise.shape[-1] * noise.shape[-2]
                    _lib.check(lib.maua_styledconv_torgb_partial_f32(
                        x.data_ptr(), conv.packed_wino(5).data_ptr(), s_ptr, s.shape[1], _lib.ptr(d),
                        out.data_ptr(), b, cin, conv.out_channel, h, w, 5, float(conv.scale), _lib.ptr(noise), nstride,
                        self.noise.weight.data_ptr(), self.activate.bias.data_ptr(), t.conv.weight.data_ptr(),
                        s.data_ptr() + 4 * rgb["s_off"], float(t.conv.scale), part.data_ptr(), src, slot, post_ptr,
                        _lib.stream_ptr(x.device)), "maua_styledconv_torgb_partial_f32")
                    self.posted = post_ptr is not None
                    _lib.check(lib.maua_torgb_f32(part.data_ptr(), None, None, 0, t.bias.data_ptr(),
                                                  _lib.ptr(skip), _lib.ptr(t.upsample.kernel) if skip is not None else None,
                                                  rgb["out"].data_ptr(), b, 3 * m_tiles, h, w, 1.0, _lib.stream_ptr(x.device)),
                               "maua_torgb_f32")  # (w = s = NULL: the plane sum + bias + up-sampled skip)
                    rgb["done"] = True
                    rgb["u8_done"] = False  # the image is in rgb["out"] as fp32 planes: a last layer still needs the frame epilogue
                    return out
            if low and rgb is not None and self.partial_rgb_fusion and (w % 4 == 0) and post_ptr is None:
                t = rgb["module"]
                skip = rgb["skip"]
                if skip is None or (tuple(t.upsample.kernel.shape) == (4, 4) and t.upsample.factor == 2
                                    and skip.shape[2] * 2 == h and skip.shape[3] * 2 == w):
                    groups = conv.out_channel // 32
                    lws = bufs(tag + ".lws", (lib.maua_lowres_ws_floats(b, cin, conv.out_channel, h, w, low_mode),))
                    part = bufs(tag + ".rgb_partial", (b, 3 * groups, h, w))
                    nstride = 0 if noise is None or noise.shape[0] == 1 else noise.shape[-1] * noise.shape[-2]
                    wpk = conv.packed_wino(low_mode) if low_mode >= 2 else conv.packed()[0]
                    _lib.check(lib.maua_styledconv_rgbpart_lowres_f32(
                        x.data_ptr(), wpk.data_ptr(), s_ptr, s.shape[1], _lib.ptr(d), out.data_ptr(), lws.data_ptr(),
                        _lib.ptr(noise), nstride, self.noise.weight.data_ptr(), self.activate.bias.data_ptr(), t.conv.weight.data_ptr(),
                        s.data_ptr() + 4 * rgb["s_off"], float(t.conv.scale), part.data_ptr(), src, slot, b, cin, conv.out_channel, h, w,
                        low_mode, float(conv.scale), _lib.stream_ptr(x.device)), "maua_styledconv_rgbpart_lowres_f32")
                    _lib.check(lib.maua_torgb_f32(part.data_ptr(), None, None, 0, t.bias.data_ptr(), _lib.ptr(skip),
                                                  _lib.ptr(t.upsample.kernel) if skip is not None else None, rgb["out"].data_ptr(), b,
                                                  3 * groups, h, w, 1.0, _lib.stream_ptr(x.device)), "maua_torgb_f32")
                    rgb["done"] = True
                    rgb["u8_done"] = False
                    self.last_path = "lowres"
                    return out
            if ws is None and n_ws:
                ws = bufs(tag + ".ws", (n_ws,))
            self.last_path = "plain"
            return conv.run(x, s, s_off, d, out, ws, fuse_act=True, noise=noise, noise_w=self.noise.weight,
                            bias=self.activate.bias, src=src, slot=slot, prescaled=prescaled)
        k = conv.blur.kernel
        pad0, pad1 = conv.blur.pad
        if (self.fused_blur_min_width <= w and conv.conv_mode(h, w) == 6 and (pad0, pad1) == (1, 1) and conv.blur_is_separable()
                and lib.maua_upconv_blur_ok(cin, conv.out_channel, h, w)):
            # the whole layer in one pass over the transposed convolution's accumulators: no raw (2H+1) x (2W+1) map (csrc/modconv_up2d.hip)
            self.last_path = "fused"
            out = bufs(tag, (b, conv.out_channel, 2 * h, 2 * w))
            n_seam = lib.maua_upconv_blur_ws_floats(b, cin, conv.out_channel, h, w)
            seam = bufs(tag + ".seam", (n_seam,)) if n_seam else None
            nstride = 0 if noise is None or noise.shape[0] == 1 else 4 * h * w
            _lib.check(lib.maua_upconv_blur_f32(
                x.data_ptr(), conv.packed_wino(6).data_ptr(), s_ptr, s.shape[1], _lib.ptr(d), out.data_ptr(),
                _lib.ptr(seam), k.data_ptr(), _lib.ptr(noise), nstride, self.noise.weight.data_ptr(), self.activate.bias.data_ptr(),
                src, slot, b, cin, conv.out_channel, h, w, float(conv.scale), post_ptr, _lib.stream_ptr(x.device)), "maua_upconv_blur_f32")
            self.posted = post_ptr is not None
            return out
        if low and tuple(k.shape) == (4, 4) and (pad0, pad1) == (1, 1):
            # transposed convolution -> split-K slabs, then ONE launch: slab sum, demodulation, blur, noise, bias, leaky ReLU (+ the style fold's scale)
            self.last_path = "lowres"
            out = bufs(tag, (b, conv.out_channel, 2 * h, 2 * w))
            lws = bufs(tag + ".lws", (lib.maua_lowres_ws_floats(b, cin, conv.out_channel, h, w, low_up),))
            nstride = 0 if noise is None or noise.shape[0] == 1 else 4 * h * w
            wpk = conv.packed_wino(6) if low_up == 6 else conv.packed()[0]
            _lib.check(lib.maua_upconv_blur_lowres_f32(
                x.data_ptr(), wpk.data_ptr(), s_ptr, s.shape[1], _lib.ptr(d), out.data_ptr(), lws.data_ptr(), k.data_ptr(),
                _lib.ptr(noise), nstride, self.noise.weight.data_ptr(), self.activate.bias.data_ptr(), src, slot, b, cin, conv.out_channel,
                h, w, low_up, float(conv.scale), post_ptr, _lib.stream_ptr(x.device)), "maua_upconv_blur_lowres_f32")
            self.posted = post_ptr is not None
            return out
        if ws is None and n_ws:
            ws = bufs(tag + ".ws", (n_ws,))
        self.last_path = "pair"
        raw = bufs(tag + ".raw", (b, conv.out_channel, 2 * h + 1, 2 * w + 1))
        conv.run(x, s, s_off, d, raw, ws, prescaled=prescaled)
        oh, ow = raw.shape[2] + pad0 + pad1 - k.shape[0] + 1, raw.shape[3] + pad0 + pad1 - k.shape[1] + 1
        out = bufs(tag, (b, conv.out_channel, oh, ow))
        nstride = 0 if noise is None or noise.shape[0] == 1 else oh * ow
        rc = lib.maua_blur_noise_act_f32(raw.data_ptr(), k.data_ptr(), out.data_ptr(), b, conv.out_channel, raw.shape[2],
                                         raw.shape[3], k.shape[0], k.shape[1], pad0, pad1, None, _lib.ptr(noise), nstride,
                                         self.noise.weight.data_ptr(), self.activate.bias.data_ptr(), src, slot,
                                         post_ptr, s.shape[1], _lib.stream_ptr(x.device))
        _lib.check(rc, "maua_blur_noise_act_f32")
        self.posted = post_ptr is not None
        return out

    def forward(self, inputs, style, noise=None, transform_dict_list=[]):
        x = _lib.require_cuda(inputs, "inputs")
        style = _lib.require_cuda(style, "style")
        b, cin = x.shape[:2]
        dev = x.device
        lib = _lib.load()
        with th.cuda.device(dev):
            s = th.empty((b, cin), dtype=th.float32, device=dev)
            d = th.empty((b, self.conv.out_channel), dtype=th.float32, device=dev)
            table = _style_table([self.conv.table_entry(0, 0, 0)], dev)
            lat = style.reshape(b, 1, -1)
            st = _lib.stream_ptr(dev)
            _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), b, 1, lat.shape[-1], None, None, table.data_ptr(), 1,
                                                 cin, s.data_ptr(), cin, None, st), "maua_style_affine_f32")
            _lib.check(lib.maua_demod_f32(table.data_ptr(), 1, self.conv.out_channel, s.data_ptr(), cin, d.data_ptr(), b,
                                          st), "maua_demod_f32")
            if noise is None:
                up = 2 if self.conv.upsample else 1
                noise = th.randn(b, 1, x.shape[2] * up, x.shape[3] * up, device=dev)
            out = self.run(x, s, 0, d if self.conv.demodulate else None, noise,
                           lambda name, shape: th.empty(shape, dtype=th.float32, device=dev), "out")
        return self.manipulation(out, transform_dict_list)


class ToRGB(nn.Module):
    """reference :346-365."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=[1, 3, 3, 1]):
        super().__init__()
        if upsample:
            self.upsample = Upsample(blur_kernel)
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(th.zeros(1, 3, 1, 1))

    def run(self, x, s, s_off, skip, out):
        lib = _lib.load()
        b, cin, h, w = x.shape
        k4 = None
        if skip is not None:
            k4 = self.upsample.kernel
            if tuple(k4.shape) != (4, 4) or self.upsample.factor != 2:
                raise NotImplementedError("fused ToRGB skip path is built for the 4-tap / factor-2 Upsample")
            if skip.shape[2] * 2 != h or skip.shape[3] * 2 != w:
                raise RuntimeError(f"skip {tuple(skip.shape)} is not half of {h}x{w}")
        rc = lib.maua_torgb_f32(x.data_ptr(), self.conv.weight.data_ptr(), s.data_ptr() + 4 * s_off, s.shape[1],
                                self.bias.data_ptr(), _lib.ptr(skip), _lib.ptr(k4), out.data_ptr(), b, cin, h, w,
                                float(self.conv.scale), _lib.stream_ptr(x.device))
        _lib.check(rc, "maua_torgb_f32")
        return out

    def forward(self, inputs, style, skip=None):
        x = _lib.require_cuda(inputs, "inputs")
        style = _lib.require_cuda(style, "style")
        b, cin, h, w = x.shape
        dev = x.device
        lib = _lib.load()
        with th.cuda.device(dev):
            s = th.empty((b, cin), dtype=th.float32, device=dev)
            table = _style_table([self.conv.table_entry(0, 0, 0)], dev)
            lat = style.reshape(b, 1, -1)
            _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), b, 1, lat.shape[-1], None, None, table.data_ptr(), 1,
                                                 cin, s.data_ptr(), cin, None, _lib.stream_ptr(dev)), "maua_style_affine_f32")
            out = th.empty((b, 3, h, w), dtype=th.float32, device=dev)
            if skip is not None:
                skip = _lib.require_cuda(skip, "skip")
            return self.run(x, s, 0, skip, out)


class Generator(nn.Module):
    """reference :368-576.  ``constant_input=True`` is what load_generator passes by default
    (generate_audiovisual.py:49); ``--noconst`` checkpoints use LatentInput (:281-294,409-412)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01,
                 constant_input=False, checkpoint=None, output_size=None, min_rgb_size=4, base_res_factor=1):
        super().__init__()
        _SKIP_INIT.on = checkpoint is not None  # (see _randn; thread-local; reset below, also when construction fails)
        try:
            self._build(size, style_dim, n_mlp, channel_multiplier, blur_kernel, lr_mlp, constant_input, min_rgb_size)
        finally:
            _SKIP_INIT.on = False
        self.truncation_latent = None
        if checkpoint is not None:
            try:  # zip-format checkpoints are mapped instead of read (0.1 s for the 120 MB of a 1024^2 generator)
                state = th.load(checkpoint, map_location="cpu", mmap=True)
            except (RuntimeError, ValueError, TypeError):
                state = th.load(checkpoint)
            # the tensors above were left uninitialised: EVERY parameter and buffer must come from the checkpoint (strict load,
            # and the key sets are compared explicitly so that a later relaxation to strict=False cannot leave garbage weights)
            missing = set(self.state_dict().keys()) - set(state["g_ema"].keys())
            if missing:
                raise RuntimeError(f"checkpoint {checkpoint!r} lacks {sorted(missing)[:5]} ... ({len(missing)} tensors): the generator "
                                   "was built without initial values and cannot be completed from it")
            self.load_state_dict(state["g_ema"], strict=True)
        self._random_noise_buffers = size != output_size or base_res_factor != 1
        if self._random_noise_buffers:  # reference :461-470 (resizes only the noise buffers)
            for layer_idx in range(self.num_layers):
                res = (layer_idx + 5) // 2
                shape = [1, 1, int(base_res_factor * 2 ** res * (2 if output_size == 1080 else 1)),
                         int(base_res_factor * 2 ** res * (2 if output_size == 1920 else 1))]
                setattr(self.noises, f"noise_{layer_idx}", th.randn(*shape))
        self._bufs = {}
        self._arena = {}  # (lane, device) -> (current chunk, bytes used): see _arena_tensor
        self._retired = []  # replaced static buffers, kept alive for graphs that still reference them
        self._captured = False
        self._lane = 0  # static-buffer namespace: concurrent hipGraphs of one generator each own a lane
        self._tables = {}

    # parity-test tap: a forward that writes uint8 frames from the last layer's epilogue ALSO leaves the fp32 image of the last
    # resolution (``GraphLane.image``) — the same kernel instance writes both, so the float comparison sees exactly what became the frame
    tap_float_image = False
    # the style fold (include/maua_hip.h THE STYLE FOLD; A/B switch): producers store their map multiplied by the consumer's styles
    style_fold = True

    def _build(self, size, style_dim, n_mlp, channel_multiplier, blur_kernel, lr_mlp, constant_input, min_rgb_size):
        self.size = size
        self.style_dim = style_dim
        layers = [PixelNorm()]
        for _ in range(n_mlp):
            layers.append(EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu"))
        self.style = nn.Sequential(*layers)
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * channel_multiplier, 128: 128 * channel_multiplier,
                         256: 64 * channel_multiplier, 512: 32 * channel_multiplier, 1024: 16 * channel_multiplier}
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2
        self.min_rgb_size = min_rgb_size
        self.input = ConstantInput(self.channels[4]) if constant_input else LatentInput(style_dim, self.channels[4])
        self.const_manipulation = ManipulationLayer(0)
        layerID = 1
        self.conv1 = StyledConv(self.channels[4], self.channels[4], 3, style_dim, blur_kernel=blur_kernel, layerID=layerID)
        self.to_rgb1 = ToRGB(self.channels[4], style_dim, upsample=False)
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        in_channel = self.channels[4]
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", _randn(1, 1, 2 ** res, 2 ** res))
        for i in range(3, self.log_size + 1):
            out_channel = self.channels[2 ** i]
            layerID += 1
            self.convs.append(StyledConv(in_channel, out_channel, 3, style_dim, upsample=True, blur_kernel=blur_kernel,
                                         layerID=layerID))
            layerID += 1
            self.convs.append(StyledConv(out_channel, out_channel, 3, style_dim, blur_kernel=blur_kernel, layerID=layerID))
            self.to_rgbs.append(ToRGB(out_channel, style_dim))
            in_channel = out_channel

    # ------------------------------------------------------------------ helpers shared with the reference API
    def make_noise(self):
        device = self.input.input.device
        noises = [th.randn(1, 1, 4, 4, device=device)]
        for i in range(3, self.log_size + 1):
            for _ in range(2):
                noises.append(th.randn(1, 1, 2 ** i, 2 ** i, device=device))
        return noises

    def mean_latent(self, n_latent):
        latent_in = th.randn(n_latent, self.style_dim, device=self.input.input.device)
        return self.style(latent_in).mean(0, keepdim=True)

    def get_latent(self, inputs):
        return self.style(inputs)

    # ------------------------------------------------------------------ static buffers / tables
    def _buf(self, batch, name, shape, dtype=th.float32):
        """Static device buffer keyed by (batch, name, lane).  A buffer that has to change shape is RETIRED, not freed:
        hipGraphs captured earlier still launch kernels on its address."""
        key = (batch, name, self._lane)
        t = self._bufs.get(key)
        shape = tuple(int(v) for v in shape)
        if t is None or tuple(t.shape) != shape or t.device != self.input.input.device:
            if t is not None and self._captured:
                self._retired.append(t)
            t = self._arena_tensor(shape, dtype)
            self._bufs[key] = t
        return t

    _ARENA_CHUNK = 256 << 20

    def _arena_tensor(self, shape, dtype):
        """Static buffers are carved out of a few large device allocations per lane instead of one allocation each: a 1024^2
        generator owns ~240 buffers per lane, and 720 hipMalloc calls were 0.3-0.4 s of every generate() (the reference's
        `empty_cache()` before the render hands the cached blocks of the previous run back to the driver, so they are real
        allocations each time).  Buffers of half a chunk and more get an allocation of their own; nothing is ever returned to
        an arena (a retired buffer keeps its bytes, hipGraphs captured earlier still write there)."""
        dev = self.input.input.device
        n = 1
        for v in shape:
            n *= v
        nbytes = n * th.empty((), dtype=dtype).element_size()
        aligned = max(256, (nbytes + 255) & ~255)
        if aligned >= self._ARENA_CHUNK // 2:
            raw = th.empty(aligned, dtype=th.uint8, device=dev)
        else:
            key = (self._lane, str(dev))
            chunk, used = self._arena.get(key, (None, 0))
            if chunk is None or used + aligned > chunk.numel():
                chunk, used = th.empty(self._ARENA_CHUNK, dtype=th.uint8, device=dev), 0
            raw = chunk[used: used + aligned]
            self._arena[key] = (chunk, used + aligned)
        return raw[:nbytes].view(dtype).view(shape)

    def _const_conv_operand(self):
        """T [Cout / 32, 16, Cin / 8, 32, 8] = conv1's weight applied to the learned constant (csrc/constconv.hip): a function of the checkpoint,
        rebuilt when either parameter changes."""
        w, c = self.conv1.conv.weight, self.input.input
        key = (w.data_ptr(), w._version, c.data_ptr(), c._version, str(w.device))
        cached = self.__dict__.get("_const_T")
        if cached is None or cached[0] != key:
            cout, cin = self.conv1.conv.out_channel, self.conv1.conv.in_channel
            T = th.empty(cout * 16 * cin, dtype=th.float32, device=w.device)
            with th.cuda.device(w.device):
                _lib.check(_lib.load().maua_pack_const_conv_f32(_lib.require_cuda(w.detach(), "weight").data_ptr(),
                                                                _lib.require_cuda(c.detach(), "input").data_ptr(), T.data_ptr(), cout, cin, 4, 4,
                                                                _lib.stream_ptr(w.device)), "maua_pack_const_conv_f32")
            self.__dict__["_const_T"] = cached = (key, T)
        return cached[1]

    def _style_layers(self):
        """(module ModulatedConv2d, latent index) in forward order: conv1, to_rgb1, then per resolution
        conv_up, conv, to_rgb with latent indices i, i+1, i+2 (reference :549-569)."""
        seq = [(self.conv1.conv, 0), (self.to_rgb1.conv, 1)]
        i = 1
        for n in range(self.log_size - 2):
            seq += [(self.convs[2 * n].conv, i), (self.convs[2 * n + 1].conv, i + 1), (self.to_rgbs[n].conv, i + 2)]
            i += 2
        return seq

    def _table(self, batch):
        seq = self._style_layers()
        key = (batch,) + tuple((m.weight.data_ptr(), m.weight._version, m.modulation.weight.data_ptr()) for m, _ in seq)
        cached = self._tables.get(batch)
        if cached is not None and cached["key"] == key:
            return cached
        entries, s_off, d_off = [], 0, 0
        for m, lat_idx in seq:
            entries.append(m.table_entry(lat_idx, s_off, d_off))
            s_off += m.in_channel
            d_off += batch * m.out_channel if m.demodulate else 0
        info = dict(key=key, table=_style_table(entries, self.input.input.device), entries=entries, s_total=s_off,
                    d_total=max(d_off, 1), max_cin=max(e["cin"] for e in entries),
                    max_cout=max(e["cout"] for e in entries))
        self._tables[batch] = info
        return info

    # ------------------------------------------------------------------ forward
    def forward(self, styles, return_latents=False, return_activation_maps=False, inject_index=None, truncation=1.0,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True, transform_dict_list=[],
                map_latents=False):
        dev = self.input.input.device
        if dev.type != "cuda":
            raise RuntimeError("Generator must live on a HIP device (.cuda()); the MI355X path has no CPU fallback")
        if map_latents:
            # evident intent of the reference branch (:506-509), whose literal code normalises over a singleton dim
            # (SURVEY.md §8a quirks): map z [N,512] through the mapping network and repeat to n_latent.
            return self.style(styles)[:, None, :].repeat(1, self.n_latent, 1)
        if not input_is_latent:
            styles = [self.style(s) for s in styles]
            if len(styles) < 2:
                inject_index = self.n_latent
                latent = styles[0].unsqueeze(1).repeat(1, inject_index, 1) if styles[0].ndim < 3 else styles[0]
            else:
                if inject_index is None:
                    import random
                    inject_index = random.randint(1, self.n_latent - 1)
                latent = th.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                 styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        else:
            latent = styles
            if latent.dim() == 2:
                latent = latent[:, None, :].repeat(1, self.n_latent, 1)
        latent = _lib.require_cuda(latent.to(dev), "styles")
        batch = latent.shape[0]
        # the reference only indexes latent[:, i] for i < n_latent (:549-569): surplus rows — an 18-layer latent file fed to a
        # 256-px generator — are legal and ignored; the kernels take the row count as a stride
        if latent.dim() != 3 or latent.shape[1] < self.n_latent or latent.shape[2] != self.style_dim:
            raise RuntimeError(f"styles {tuple(latent.shape)} do not match [batch, >= {self.n_latent}, {self.style_dim}]")
        latent = latent.contiguous()

        noise = list(noise) if noise is not None else [None] * self.num_layers
        for ns in range(self.num_layers):
            if noise[ns] is None and not randomize_noise:
                noise[ns] = getattr(self.noises, f"noise_{ns}")
        # truncation (:537-543).  A float 1.0 is an exact identity and skips the lazy random mean_latent.
        trunc = None
        if not (isinstance(truncation, float) and truncation == 1.0 and truncation_latent is None
                and self.truncation_latent is None):
            if isinstance(truncation, float):
                truncation = th.full((1,), truncation, device=dev)
            if self.truncation_latent is None:
                self.truncation_latent = truncation_latent if truncation_latent is not None else self.mean_latent(2 ** 14)
            trunc = _lib.require_cuda(truncation.to(dev).float().reshape(-1), "truncation")
            if trunc.numel() == 1 and batch > 1:
                trunc = trunc.expand(batch).contiguous()
            if trunc.numel() != batch:
                raise RuntimeError(f"truncation has {trunc.numel()} entries for a batch of {batch}")
        tl = _lib.require_cuda(self.truncation_latent.to(dev).reshape(-1), "truncation_latent") if trunc is not None else None

        with th.cuda.device(dev):
            image, acts, lat_out = self._forward_device(latent, noise, trunc, tl, transform_dict_list,
                                                        return_activation_maps, return_latents)
        # the device forward writes into static per-(batch, layer, lane) buffers that the next call overwrites; the public
        # call hands out private copies, like the reference's freshly allocated outputs (render / hipGraph replay read the
        # static buffers directly and never pass through here)
        image = image.clone() if image is not None else None  # None: min_rgb_size above the output size (reference :553-568)
        if return_activation_maps:
            return image, [a.clone() for a in acts]
        if return_latents:
            return image, lat_out
        return image, None

    def _forward_device(self, latent, noise, trunc, tl, bends, want_acts=False, want_latents=False, frames_u8=None, src=None,
                        batch=None):
        """``frames_u8`` ([B, H, W, 3] uint8): the frame epilogue of render.py:40-43 (clamp, scale, NHWC, uint8) is folded into
        the last layer's fused ToRGB epilogue — the fp32 image of the last resolution is then never written (``image`` returned
        is None) — or, when that layer cannot take the fused path, applied by maua_frames_to_u8 right behind it.
        ``src`` (device pointer of a maua_frame_source_t) + ``batch``: latents, truncation and every noise map are read through
        the frame source (``latent`` / ``noise`` / ``trunc`` are ignored); this is the form ``capture_graph`` records."""
        lib = _lib.load()
        dev = self.input.input.device
        if src is None:
            batch = latent.shape[0]
        st = _lib.stream_ptr(dev)
        info = self._table(batch)
        bufs = lambda name, shape: self._buf(batch, name, shape)  # noqa: E731
        s = bufs("styles", (batch, info["s_total"]))
        d = bufs("demod", (info["d_total"],))
        lat_rows = self.n_latent if latent is None else latent.shape[1]  # row stride of the latents (surplus rows are skipped)
        _lib.check(lib.maua_style_affine_f32(_lib.ptr(latent), batch, lat_rows, self.style_dim, _lib.ptr(trunc),
                                             _lib.ptr(tl), info["table"].data_ptr(), len(info["entries"]),
                                             info["max_cin"], s.data_ptr(), info["s_total"], src, st), "maua_style_affine_f32")
        _lib.check(lib.maua_demod_f32(info["table"].data_ptr(), len(info["entries"]), info["max_cout"], s.data_ptr(),
                                      info["s_total"], d.data_ptr(), batch, st), "maua_demod_f32")
        ent = info["entries"]

        def demod_of(e):
            if e["wsq"] is None:
                return None
            return d[e["d_off"]: e["d_off"] + batch * e["cout"]].view(batch, e["cout"])

        def noise_for(i, h, w):
            if src is not None:
                return None  # slot i of the frame source
            nz = noise[i]
            if nz is None:  # randomize_noise=True: fresh N(0,1) per call (:262-265)
                nz = bufs(f"rand_noise_{i}", (batch, 1, h, w)).normal_()
            return nz.to(dev)

        # the style fold (include/maua_hip.h): a layer's epilogue multiplies the map it stores by the NEXT convolution's styles, which then
        # runs without the multiplies in its K loop.  Only where nothing else reads the map: no bend on the producer's layer id, no
        # activation maps handed out (the ToRGB of a plain layer is computed in its own epilogue from the un-scaled value).
        fold = self.style_fold and not want_acts

        def post_for(layer_id, consumer, h, w, consumer_entry):
            """s offset of ``consumer``'s styles if the producer with ``layer_id`` may store its map pre-multiplied by them, else None."""
            if not fold or any(bd["layer"] == layer_id for bd in bends) or not consumer.accepts_prescaled(h, w):
                return None
            return consumer_entry["s_off"]

        acts = []
        # conv1 on a ConstantInput nobody bends: y = T s with T = W * const precomputed (csrc/constconv.hip) — no [B, C, 4, 4] copy of the
        # constant, no convolution
        c1 = self.conv1.conv
        const_conv = (isinstance(self.input, ConstantInput) and self.conv1.lowres_fusion and not any(bd["layer"] == 0 for bd in bends)
                      and tuple(self.input.input.shape[2:]) == (4, 4) and c1.kernel_size == 3 and not c1.upsample
                      and lib.maua_const_conv_ok(c1.in_channel, c1.out_channel, 4, 4))
        x = None
        if isinstance(self.input, LatentInput):
            x = self.input.run(latent, trunc, tl, bufs("const", (batch, self.input.channel, self.input.size, self.input.size)),
                               src=src, n_latent=self.n_latent, style_dim=self.style_dim)
        elif not const_conv:
            x = bufs("const", (batch,) + tuple(self.input.input.shape[1:]))
            x.copy_(self.input.input.expand(batch, -1, -1, -1))
        if x is not None:
            x = self.const_manipulation.run(x, bends, bufs, "const", src)
        li = 0
        # min_rgb_size (reference :553,567): resolutions below it contribute no ToRGB, the skip chain starts later
        current_size = 4
        image = None
        fuse1 = None
        hw0 = tuple(self.input.input.shape[2:]) if x is None else tuple(x.shape[2:])
        if (self.min_rgb_size <= current_size and not any(bd["layer"] == 1 for bd in bends)
                and not getattr(self, "disable_rgb_fusion", False)):
            fuse1 = dict(module=self.to_rgb1, s_off=ent[li + 1]["s_off"], skip=None, out=bufs("rgb1", (batch, 3) + hw0), store=True)
        if const_conv:
            out = self._run_const_conv(s, ent[li]["s_off"], demod_of(ent[li]), noise_for(0, 4, 4), bufs, fuse1, src, batch)
        else:
            out = self.conv1.run(x, s, ent[li]["s_off"], demod_of(ent[li]), noise_for(0, x.shape[2], x.shape[3]), bufs, "conv1",
                                 rgb=fuse1, src=src, slot=0)  # (conv1 never posts: its consumer is a polyphase layer without a pre-scaled instance)
        posted = False  # whether `out` carries the next convolution's styles already
        out = self.conv1.manipulation.run(out, bends, bufs, "conv1", src)
        acts.append(out)
        li += 1
        if fuse1 is not None and fuse1.get("done"):
            image = fuse1["out"]
        elif self.min_rgb_size <= current_size:
            image = self.to_rgb1.run(out, s, ent[li]["s_off"], None, bufs("rgb1", (batch, 3) + tuple(out.shape[2:])))
        li += 1
        for n in range(self.log_size - 2):
            up, plain, rgb = self.convs[2 * n], self.convs[2 * n + 1], self.to_rgbs[n]
            out = up.run(out, s, ent[li]["s_off"], demod_of(ent[li]), noise_for(2 * n + 1, out.shape[2] * 2, out.shape[3] * 2),
                         bufs, f"convs.{2 * n}", src=src, slot=2 * n + 1, prescaled=posted,
                         post_off=post_for(2 * n + 2, plain, out.shape[2] * 2, out.shape[3] * 2, ent[li + 1]))
            posted = up.posted
            out = up.manipulation.run(out, bends, bufs, f"convs.{2 * n}", src)
            acts.append(out)
            li += 1
            current_size *= 2
            # fold ToRGB into the conv epilogue where the layer qualifies (<= 64 channels) and nothing needs the feature
            # map in between (a bend on this layer id would); the last layer then never writes its feature map at all
            layer_id = 2 * n + 3
            bent = any(bd["layer"] == layer_id for bd in bends)
            rgb_buf = bufs(f"rgbs.{n}", (batch, 3, out.shape[2], out.shape[3]))
            fuse = None
            wants_rgb = self.min_rgb_size <= current_size
            is_last = n == self.log_size - 3
            if wants_rgb and not bent and not getattr(self, "disable_rgb_fusion", False):
                fuse = dict(module=rgb, s_off=ent[li + 1]["s_off"], skip=image, out=rgb_buf,
                            store=(not is_last) or want_acts, u8=frames_u8 if is_last else None, tap=self.tap_float_image)
            nxt = self.convs[2 * n + 2] if not is_last else None
            out = plain.run(out, s, ent[li]["s_off"], demod_of(ent[li]), noise_for(2 * n + 2, out.shape[2], out.shape[3]),
                            bufs, f"convs.{2 * n + 1}", rgb=fuse, src=src, slot=2 * n + 2, prescaled=posted,
                            post_off=None if nxt is None else post_for(layer_id, nxt, out.shape[2], out.shape[3], ent[li + 2]))
            posted = plain.posted
            out = plain.manipulation.run(out, bends, bufs, f"convs.{2 * n + 1}", src)
            acts.append(out)
            li += 1
            if fuse is not None and fuse.get("done"):
                image = rgb_buf
                if is_last and frames_u8 is not None and fuse.get("u8_done", True) and not self.tap_float_image:
                    image = None  # left the device path as uint8 frames
            elif wants_rgb:
                assert not posted  # (a separate ToRGB pass reads the un-scaled map: plain.run only posts from the ToRGB-fused paths)
                image = rgb.run(out, s, ent[li]["s_off"], image, rgb_buf)
            li += 1
        if frames_u8 is not None and image is not None:  # last layer not fusable (bend on it, > 64 channels, ...)
            _lib.check(lib.maua_frames_to_u8(image.data_ptr(), frames_u8.data_ptr(), batch, image.shape[2], image.shape[3], st),
                       "maua_frames_to_u8")
        lat_out = None
        if want_latents:
            lat_out = latent if trunc is None else tl[None, None, :] + trunc[:, None, None] * (latent - tl[None, None, :])
        return image, acts, lat_out

    def _run_const_conv(self, s, s_off, d, noise, bufs, rgb, src, batch):
        """conv1 + noise + bias + activation (+ the partial ToRGB sums of to_rgb1 and their plane sum) on the constant input: one launch of
        maua_const_styledconv_f32 (+ one of maua_torgb_f32's plane-sum form)."""
        lib = _lib.load()
        m, c1 = self.conv1, self.conv1.conv
        dev = self.input.input.device
        cin, cout = c1.in_channel, c1.out_channel
        out = bufs("conv1", (batch, cout, 4, 4))
        if src is not None:
            noise = None
        if noise is not None:
            noise = _lib.require_cuda(noise, "noise")
            if noise.dim() != 4 or noise.shape[1] != 1 or tuple(noise.shape[-2:]) != (4, 4) or noise.shape[0] not in (1, batch):
                raise RuntimeError(f"noise {tuple(noise.shape)} does not match feature map [{batch}, 1, 4, 4] (batch must be 1 or {batch})")
            noise = noise.contiguous()
        nstride = 0 if noise is None or noise.shape[0] == 1 else 16
        part, t = None, None
        if rgb is not None:
            t = rgb["module"]
            part = bufs("conv1.rgb_partial", (batch, 3 * (cout // 32), 4, 4))
        st = _lib.stream_ptr(dev)
        _lib.check(lib.maua_const_styledconv_f32(
            self._const_conv_operand().data_ptr(), s.data_ptr() + 4 * s_off, s.shape[1], _lib.ptr(d), out.data_ptr(), _lib.ptr(noise), nstride,
            m.noise.weight.data_ptr(), m.activate.bias.data_ptr(), None if t is None else t.conv.weight.data_ptr(),
            None if t is None else s.data_ptr() + 4 * rgb["s_off"], 0.0 if t is None else float(t.conv.scale), _lib.ptr(part), src, 0, batch,
            cin, cout, 4, 4, float(c1.scale), st), "maua_const_styledconv_f32")
        if rgb is not None:
            _lib.check(lib.maua_torgb_f32(part.data_ptr(), None, None, 0, t.bias.data_ptr(), None, None, rgb["out"].data_ptr(), batch,
                                          3 * (cout // 32), 4, 4, 1.0, st), "maua_torgb_f32")
            rgb["done"] = True
        m.posted = False
        m.last_path = "const"
        return out

    # ------------------------------------------------------------------ hipGraph
    def weights_key(self):
        """Identity of everything a captured graph has baked in as pointers: parameters and buffers (a swapped or in-place
        modified tensor invalidates the packed weights, style tables and therefore the graph)."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def capture_graph(self, batch, lane=0, frames_u8=False, bends=()):
        """Capture one forward of ``batch`` frames into a hipGraph and return its ``GraphLane``.  The captured kernels read
        their per-frame inputs (latents, truncation, noise maps, bend parameters) THROUGH a frame source in device memory
        (include/maua_hip.h): ``lane.bind(latents, noise, truncation)`` points it at sequences resident in HBM — any render,
        any mix of per-frame maps and checkpoint noise buffers — and ``lane.replay(frame0)`` moves only the frame index, so a
        graph is captured once per (batch, lane, bend set) and outlives the render (the reference re-uploads every slice per
        batch, render.py:140-149).  ``bends``: [{"layer", "transform"}] whose transforms implement ``run_static`` (audioreactive
        /bend.py).  Lanes share the weights but no activation buffer: they may be replayed concurrently on different streams."""
        dev = self.input.input.device
        if th.cuda.current_stream(dev).cuda_stream == 0:
            raise RuntimeError("capture_graph must run on a non-default stream (with torch.cuda.stream(s): ...): HIP cannot "
                               "capture the legacy default stream")
        bends = list(bends)
        for bd in bends:
            if not hasattr(bd["transform"], "run_static"):
                raise RuntimeError(f"bend on layer {bd['layer']} is not capturable: {type(bd['transform']).__name__} has no run_static")
        self._lane = lane
        self._captured = True
        try:
            with th.cuda.device(dev):
                source = FrameSource(self, batch, lane)
                u8 = None
                if frames_u8:
                    hw = getattr(self.noises, f"noise_{self.num_layers - 1}").shape[-2:]
                    u8 = self._buf(batch, "g.frames_u8", (batch, int(hw[0]), int(hw[1]), 3), dtype=th.uint8)
                tl = self._buf(0, "g.trunc_latent", (self.style_dim,))
                run = lambda: self._forward_device(None, None, None, tl, bends, frames_u8=u8, src=source.ptr, batch=batch)  # noqa: E731
                run()  # warm-up: allocates every static buffer, packs the weights
                th.cuda.synchronize(dev)
                graph = _lib.HipGraph()
                with graph:
                    image, _, _ = run()
            return GraphLane(self, graph, source, image, u8, tl, batch, lane, self.weights_key())
        finally:
            self._lane = 0


class FrameSource:
    """Host handle of a maua_frame_source_t in device memory (include/maua_hip.h): the pointers of the HBM-resident per-frame
    sequences a captured forward reads, plus the frame it starts at."""

    def __init__(self, generator, batch, lane):
        self.generator = generator
        self.batch = batch
        dev = generator.input.input.device
        self.dev = generator._buf(batch, "g.frame_source", (ctypes.sizeof(_lib.FrameSource),), dtype=th.uint8)
        self.ptr = self.dev.data_ptr()
        self._keep = None
        # until bound: zero latents of one batch, the checkpoint's noise buffers (stride 0), no truncation
        zeros = generator._buf(batch, "g.latents0", (batch, generator.n_latent, generator.style_dim))
        zeros.zero_()
        self.noise_hw = [tuple(getattr(generator.noises, f"noise_{i}").shape[-2:]) for i in range(generator.num_layers)]
        self.bind(zeros, [None] * generator.num_layers, None, _n_frames=batch)
        del dev

    def bind(self, latents, noise, trunc, _n_frames=None):
        """latents [n_frames, n_latent, style_dim]; noise: per slot None (the checkpoint's buffer for every frame), a
        [n_frames, 1, h, w] sequence or one shared [1, 1, h, w] map; trunc [n_frames] or None — all fp32, on the device,
        contiguous; they are kept alive by this handle until the next bind."""
        g = self.generator
        dev = g.input.input.device
        n_frames = latents.shape[0] if _n_frames is None else _n_frames
        host = _lib.FrameSource()
        latents = _lib.require_cuda(latents, "latents")
        if latents.dim() != 3 or latents.shape[1] < g.n_latent or latents.shape[2] != g.style_dim:
            raise RuntimeError(f"latents {tuple(latents.shape)} do not match [n_frames, >= {g.n_latent}, {g.style_dim}]")
        if latents.shape[1] != g.n_latent or not latents.is_contiguous():
            # surplus rows (the reference only indexes latent[:, i], i < n_latent): the captured kernels have the row stride
            # n_latent baked in, so the sequence is cut once per render
            latents = latents[:, :g.n_latent].contiguous()
        host.latents = latents.data_ptr()
        keep = [latents]
        if trunc is not None:
            trunc = _lib.require_cuda(trunc, "truncation").reshape(-1)
            if trunc.numel() != n_frames:
                raise RuntimeError(f"truncation has {trunc.numel()} entries for {n_frames} frames")
            host.trunc = trunc.data_ptr()
            keep.append(trunc)
        if len(noise) != g.num_layers or g.num_layers > _lib.MAX_NOISE_SLOTS:
            raise RuntimeError(f"{len(noise)} noise entries for {g.num_layers} layers")
        for i, nz in enumerate(noise):
            if nz is None:
                nz = getattr(g.noises, f"noise_{i}")
            nz = _lib.require_cuda(nz.to(dev), f"noise[{i}]")
            if nz.dim() != 4 or nz.shape[1] != 1 or tuple(nz.shape[-2:]) != self.noise_hw[i] or nz.shape[0] not in (1, n_frames):
                raise RuntimeError(f"noise[{i}] {tuple(nz.shape)} does not match [{n_frames} or 1, 1, {self.noise_hw[i][0]}, "
                                   f"{self.noise_hw[i][1]}]")
            host.noise[i] = nz.data_ptr()
            host.noise_stride[i] = 0 if nz.shape[0] == 1 else nz.shape[-1] * nz.shape[-2]
            keep.append(nz)
        self.n_frames = n_frames
        self._keep = keep
        raw = th.frombuffer(bytearray(bytes(host)), dtype=th.uint8)
        th.cuda.synchronize(dev)  # no replay of this lane may still be reading the previous pointers (bind is once per render)
        self.dev.copy_(raw.to(dev), non_blocking=False)
        th.cuda.synchronize(dev)

    def release(self):
        """Drop the references to the bound sequences (the end of a render): a replay before the next ``bind`` raises instead of
        reading memory the allocator may have handed out again."""
        self._keep = None
        self.n_frames = 0

    def seek(self, frame0, stream=None):
        if frame0 < 0 or frame0 + self.batch > self.n_frames:
            raise RuntimeError(f"frames [{frame0}, {frame0 + self.batch}) are outside the bound sequences ({self.n_frames} frames)")
        _lib.check(_lib.load().maua_frame_source_seek(self.ptr, int(frame0), stream if stream is not None else _lib.stream_ptr()),
                   "maua_frame_source_seek")


class GraphLane:
    """One captured forward of a Generator: hipGraph + the frame source it reads + its output buffers."""

    def __init__(self, generator, graph, source, image, u8, trunc_latent, batch, lane, weights_key):
        self.generator, self.graph, self.source = generator, graph, source
        self.image, self.u8, self.batch, self.lane = image, u8, batch, lane
        self._trunc_latent = trunc_latent
        self.weights_key = weights_key

    def bind(self, latents, noise, truncation=None):
        """Point the lane at the sequences of a render (see FrameSource.bind).  ``truncation`` [n_frames] switches the
        truncation lerp on; its centre is the generator's truncation latent (drawn lazily as in the reference :539-540)."""
        g = self.generator
        if truncation is not None:
            if g.truncation_latent is None:
                g.truncation_latent = g.mean_latent(2 ** 14)
            self._trunc_latent.copy_(g.truncation_latent.to(self._trunc_latent.device).reshape(-1))
        self.source.bind(latents, noise, truncation)

    def release(self):
        self.source.release()

    def replay(self, frame0, stream=None):
        """Frames [frame0, frame0 + batch) of the bound sequences: one 4-byte device write + the graph launch."""
        self.source.seek(frame0, stream)
        self.graph.replay(stream)
