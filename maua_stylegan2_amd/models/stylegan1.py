"""StyleGAN1 generator (``--stylegan1``) for MI355X — host-side mirror of /root/reference/models/stylegan1.py:1-617.

Same module tree (state-dict keys ``g_mapping.dense{0..7}.*``, ``g_synthesis.blocks.{4x4..}.{const,bias,conv,conv0_up,conv1,
epi1,epi2}.*``, ``g_synthesis.torgb.*``, ``noise_{i}``), same ``G_style(output_size, checkpoint)`` constructor behaviour
(resolution probing 1024 -> 128 by ``load_state_dict``, constant widened for 1920-wide output, one noise buffer per block, the
truncation latent drawn at construction) and the same ``forward(styles, noise, truncation, ...) -> (image, None)`` contract that
render.py:175-182 relies on.  The forward itself is a different program:

  reference (per layer)                                   here
  ----------------------------------------------------    ------------------------------------------------------------------
  F.conv2d / F.conv_transpose2d with the 4-shift summed    ONE shared-weight 3x3 convolution on the MFMA kernels of the StyleGAN2
  4x4 kernel (>= 128 px), nearest upscale + conv below      path (csrc/modconv*.hip, unit styles, no demodulation): the "fused
                                                            upscale" of :83-93 is exactly nearest-upscale + the SAME kernel
                                                            flipped (shown in oracle/stylegan1_oracle.py, pinned by the golden)
  Upscale2d (view/expand/contiguous), BlurLayer (conv2d)    maua_upfirdn2d_f32 (zero-insert x 2x2 box = nearest; [1,2,1]^2/16 blur)
  bias add, NoiseLayer, LeakyReLU, InstanceNorm2d,          one launch: maua_sg1_epilogue_f32 (csrc/stylegan1.hip)
  StyleMod (5 modules, 18 F.linear)                         + one table-driven launch for all style vectors (maua_style_affine_f32)
  1x1 torgb conv                                            maua_torgb_f32

There is no CPU path: CPU tensors raise.  Bends are not applied (the reference ignores ``transform_dict_list`` here too).
"""
from collections import OrderedDict

import numpy as np
import torch as th
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib
from ..op import upfirdn2d
from .stylegan2 import ModulatedConv2d, _style_table


class MyLinear(nn.Module):
    """reference :12-37 (equalised learning rate; only the parameters and multipliers are used by the device forward)."""

    def __init__(self, input_size, output_size, gain=2 ** 0.5, use_wscale=False, lrmul=1, bias=True):
        super().__init__()
        he_std = gain * input_size ** (-0.5)
        init_std, self.w_mul = (1.0 / lrmul, he_std * lrmul) if use_wscale else (he_std / lrmul, lrmul)
        self.weight = nn.Parameter(th.randn(output_size, input_size) * init_std)
        self.bias = nn.Parameter(th.zeros(output_size)) if bias else None
        self.b_mul = lrmul

    def forward(self, x):
        return F.linear(x, self.weight * self.w_mul, None if self.bias is None else self.bias * self.b_mul)


class BlurLayer(nn.Module):
    """reference :148-167: fixed [1,2,1] x [1,2,1] / 16 depthwise blur (the constructor ignores its ``kernel`` argument)."""

    def __init__(self, kernel=[1, 2, 1], normalize=True, flip=False, stride=1):
        super().__init__()
        k = th.tensor([1.0, 2.0, 1.0])
        k = (k[:, None] * k[None, :])[None, None]
        if normalize:
            k = k / k.sum()
        self.register_buffer("kernel", k)
        self.stride = stride

    def forward(self, x):
        if self.stride != 1:
            raise NotImplementedError("BlurLayer stride != 1 is discriminator-only")
        return upfirdn2d(x, self.kernel[0, 0].contiguous(), pad=(1, 1))


class Upscale2d(nn.Module):
    """reference :170-189: nearest-neighbour upscaling, as zero insertion + 2x2 box FIR on the HIP upfirdn2d kernel."""

    def __init__(self, factor=2, gain=1):
        super().__init__()
        self.gain, self.factor = gain, factor

    def forward(self, x):
        if self.factor == 1:
            return x * self.gain if self.gain != 1 else x
        box = th.full((self.factor, self.factor), float(self.gain), device=x.device)
        return upfirdn2d(x, box, up=self.factor, pad=(self.factor - 1, 0))


class MyConv2d(nn.Module):
    """reference :40-103.  ``run`` is the device path: [nearest upscale ->] shared-weight 3x3 conv on the MFMA kernels
    [-> blur]; the bias is left to the layer epilogue (the reference adds it after the blur, :99-102)."""

    def __init__(self, input_channels, output_channels, kernel_size, gain=2 ** 0.5, use_wscale=False, lrmul=1, bias=True,
                 intermediate=None, upscale=False):
        super().__init__()
        self.upscale = Upscale2d() if upscale else None
        he_std = gain * (input_channels * kernel_size ** 2) ** (-0.5)
        self.kernel_size = kernel_size
        init_std, self.w_mul = (1.0 / lrmul, he_std * lrmul) if use_wscale else (he_std / lrmul, lrmul)
        self.weight = nn.Parameter(th.randn(output_channels, input_channels, kernel_size, kernel_size) * init_std)
        self.bias = nn.Parameter(th.zeros(output_channels)) if bias else None
        self.b_mul = lrmul
        self.intermediate = intermediate
        self._engines = {}  # flipped? -> (weight key, ModulatedConv2d shell that owns the packed weights)

    def _engine(self, flipped):
        """The StyleGAN2 path's modulated-conv machinery with unit styles and no demodulation = a plain shared-weight conv."""
        w = self.weight
        key = (w.data_ptr(), w._version, str(w.device))
        cached = self._engines.get(flipped)
        if cached is None or cached[0] != key:
            shell = ModulatedConv2d(w.shape[1], w.shape[0], 3, 64, demodulate=False)
            shell.modulation = None
            data = w.detach().flip(-1, -2).contiguous() if flipped else w.detach()
            shell.weight = nn.Parameter(data[None], requires_grad=False)
            shell.scale = float(self.w_mul)
            self._engines[flipped] = cached = (key, shell)
        return cached[1]

    def bias_vector(self):
        if self.bias is None:
            return None
        return self.bias if self.b_mul == 1 else self.bias * self.b_mul

    def run(self, x, ones):
        """x [B, Cin, H, W] -> raw conv output (no bias): [B, Cout, H, W], or [B, Cout, 2H, 2W] when ``upscale``."""
        if self.kernel_size != 3:
            raise NotImplementedError("device path of MyConv2d is the 3x3 convolution (torgb goes through maua_torgb_f32)")
        lib = _lib.load()
        flipped = False
        if self.upscale is not None:
            # >= 128 px the reference runs conv_transpose2d with the 4-shift-summed 4x4 kernel (:83-93) = nearest upscale
            # followed by the same 3x3 kernel FLIPPED; below that, nearest upscale + the kernel as stored (:94-95)
            flipped = min(x.shape[2:]) * 2 >= 128
            x = self.upscale(x)
        eng = self._engine(flipped)
        b, cin, h, w = x.shape
        out = th.empty((b, eng.out_channel, h, w), dtype=th.float32, device=x.device)
        n_ws = lib.maua_modconv_ws_floats(b, cin, eng.out_channel, h, w, eng.conv_mode(h, w))
        ws = th.empty(n_ws, dtype=th.float32, device=x.device) if n_ws else None
        eng.run(x, ones, 0, None, out, ws)
        if self.intermediate is not None:
            out = self.intermediate(out)
        return out

    def forward(self, x):
        x = _lib.require_cuda(x, "x")
        with th.cuda.device(x.device):
            ones = th.ones((x.shape[0], x.shape[1]), dtype=th.float32, device=x.device)
            out = self.run(x, ones)
        b = self.bias_vector()
        return out if b is None else out + b.view(1, -1, 1, 1)


class NoiseLayer(nn.Module):
    """reference :106-123 (parameter holder; the add is fused into maua_sg1_epilogue_f32)."""

    def __init__(self, channels):
        super().__init__()
        self.weight = nn.Parameter(th.zeros(channels))
        self.noise = None


class StyleMod(nn.Module):
    def __init__(self, latent_size, channels, use_wscale):
        super().__init__()
        self.lin = MyLinear(latent_size, channels * 2, gain=1.0, use_wscale=use_wscale)


class PixelNormLayer(nn.Module):
    def __init__(self, epsilon=1e-8):
        super().__init__()
        self.epsilon = epsilon

    def forward(self, x):
        return x * th.rsqrt(th.mean(x ** 2, dim=1, keepdim=True) + self.epsilon)


class G_mapping(nn.Sequential):
    """reference :192-223: PixelNorm + 8 x (MyLinear lrmul 0.01, LeakyReLU 0.2), broadcast to 18 layers."""

    def __init__(self, nonlinearity="lrelu", use_wscale=True):
        act, gain = {"relu": (nn.ReLU(), np.sqrt(2)), "lrelu": (nn.LeakyReLU(negative_slope=0.2), np.sqrt(2))}[nonlinearity]
        layers = [("pixel_norm", PixelNormLayer())]
        for i in range(8):
            layers.append((f"dense{i}", MyLinear(512, 512, gain=gain, lrmul=0.01, use_wscale=use_wscale)))
            layers.append((f"dense{i}_act", act))
        super().__init__(OrderedDict(layers))

    def forward(self, x):
        return super().forward(x).unsqueeze(1).expand(-1, 18, -1)


class LayerEpilogue(nn.Module):
    """reference :241-318: noise -> activation -> [pixel norm] -> [instance norm] -> style modulation.  ``run`` is one launch."""

    def __init__(self, channels, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm, use_styles,
                 activation_layer):
        super().__init__()
        if use_pixel_norm:
            raise NotImplementedError("use_pixel_norm: G_style never enables it (reference :415)")
        layers = []
        if use_noise:
            layers.append(("noise", NoiseLayer(channels)))
        layers.append(("activation", activation_layer))
        if use_instance_norm:
            layers.append(("instance_norm", nn.InstanceNorm2d(channels)))
        self.top_epi = nn.Sequential(OrderedDict(layers))
        self.use_noise, self.use_instance_norm = use_noise, use_instance_norm
        self.style_mod = StyleMod(dlatent_size, channels, use_wscale=use_wscale) if use_styles else None

    def run(self, x, bias, noise, styles, s_off, s_stride):
        """In place on ``x`` [B, C, H, W]: y = style(instance_norm(lrelu(x + bias + w_c * noise)))."""
        lib = _lib.load()
        b, c, h, w = x.shape
        nw = None
        if self.use_noise:
            if noise is None:  # NoiseLayer without a stored tensor draws fresh noise (:117-118)
                noise = th.randn(b, 1, h, w, device=x.device)
            noise = _lib.require_cuda(noise.to(x.device), "noise")
            if noise.dim() != 4 or noise.shape[1] != 1 or tuple(noise.shape[-2:]) != (h, w) or noise.shape[0] not in (1, b):
                raise RuntimeError(f"noise {tuple(noise.shape)} does not match feature map [{b}, 1, {h}, {w}] (batch 1 or {b})")
            nw = self.top_epi.noise.weight
        nstride = 0 if noise is None or noise.shape[0] == 1 else h * w
        style_ptr = styles.data_ptr() + 4 * s_off if self.style_mod is not None else None
        _lib.check(lib.maua_sg1_epilogue_f32(x.data_ptr(), _lib.ptr(bias), _lib.ptr(noise) if self.use_noise else None, nstride,
                                             _lib.ptr(nw), style_ptr, s_stride, x.data_ptr(), b, c, h, w,
                                             int(self.use_instance_norm), _lib.stream_ptr(x.device)), "maua_sg1_epilogue_f32")
        return x


class InputBlock(nn.Module):
    """reference :321-362."""

    def __init__(self, nf, dlatent_size, const_input_layer, gain, use_wscale, use_noise, use_pixel_norm, use_instance_norm,
                 use_styles, activation_layer):
        super().__init__()
        if not const_input_layer:
            raise NotImplementedError("G_style always uses the learned constant input (reference :412)")
        self.const_input_layer, self.nf = const_input_layer, nf
        self.const = nn.Parameter(th.ones(1, nf, 4, 4))
        self.bias = nn.Parameter(th.ones(nf))
        self.epi1 = LayerEpilogue(nf, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm, use_styles,
                                  activation_layer)
        self.conv = MyConv2d(nf, nf, 3, gain=gain, use_wscale=use_wscale)
        self.epi2 = LayerEpilogue(nf, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm, use_styles,
                                  activation_layer)


class GSynthesisBlock(nn.Module):
    """reference :365-410."""

    def __init__(self, in_channels, out_channels, blur_filter, dlatent_size, gain, use_wscale, use_noise, use_pixel_norm,
                 use_instance_norm, use_styles, activation_layer):
        super().__init__()
        blur = BlurLayer(blur_filter) if blur_filter else None
        self.conv0_up = MyConv2d(in_channels, out_channels, kernel_size=3, gain=gain, use_wscale=use_wscale, intermediate=blur,
                                 upscale=True)
        self.epi1 = LayerEpilogue(out_channels, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm,
                                  use_styles, activation_layer)
        self.conv1 = MyConv2d(out_channels, out_channels, kernel_size=3, gain=gain, use_wscale=use_wscale)
        self.epi2 = LayerEpilogue(out_channels, dlatent_size, use_wscale, use_noise, use_pixel_norm, use_instance_norm,
                                  use_styles, activation_layer)


class G_synthesis(nn.Module):
    """reference :413-500."""

    def __init__(self, dlatent_size=512, num_channels=3, resolution=1024, fmap_base=8192, fmap_decay=1.0, fmap_max=512,
                 use_styles=True, const_input_layer=True, use_noise=True, randomize_noise=False, nonlinearity="lrelu",
                 use_wscale=True, use_pixel_norm=False, use_instance_norm=True, dtype=th.float32, blur_filter=[1, 2, 1]):
        super().__init__()

        def nf(stage):
            return min(int(fmap_base / (2.0 ** (stage * fmap_decay))), fmap_max)

        self.dlatent_size = dlatent_size
        resolution_log2 = int(np.log2(resolution))
        assert resolution == 2 ** resolution_log2 and resolution >= 4
        act, gain = {"relu": (nn.ReLU(), np.sqrt(2)), "lrelu": (nn.LeakyReLU(negative_slope=0.2), np.sqrt(2))}[nonlinearity]
        if nonlinearity != "lrelu":
            raise NotImplementedError("the fused epilogue implements LeakyReLU(0.2), G_style's activation")
        blocks = []
        last_channels = None
        for res in range(2, resolution_log2 + 1):
            channels = nf(res - 1)
            name = "{s}x{s}".format(s=2 ** res)
            if res == 2:
                blocks.append((name, InputBlock(channels, dlatent_size, const_input_layer, gain, use_wscale, use_noise,
                                                use_pixel_norm, use_instance_norm, use_styles, act)))
            else:
                blocks.append((name, GSynthesisBlock(last_channels, channels, blur_filter, dlatent_size, gain, use_wscale,
                                                     use_noise, use_pixel_norm, use_instance_norm, use_styles, act)))
            last_channels = channels
        self.torgb = MyConv2d(channels, num_channels, 1, gain=1, use_wscale=use_wscale)
        self.blocks = nn.ModuleDict(OrderedDict(blocks))
        self._table = None

    # ------------------------------------------------------------------ device forward
    def _style_layers(self):
        out = []
        for i, block in enumerate(self.blocks.values()):
            out += [(block.epi1, 2 * i), (block.epi2, 2 * i + 1)]
        return out

    def _styles(self, dlatents):
        """All 2 * n_blocks style vectors [B, sum 2C] in one table-driven launch (StyleMod.lin: W x / sqrt(512) + b, :129-132)."""
        lib = _lib.load()
        dev = dlatents.device
        layers = [(e, i) for e, i in self._style_layers() if e.style_mod is not None]
        key = tuple((e.style_mod.lin.weight.data_ptr(), e.style_mod.lin.weight._version) for e, _ in layers) + (str(dev),)
        if self._table is None or self._table["key"] != key:
            entries, off = [], 0
            for e, lat_idx in layers:
                lin = e.style_mod.lin
                if abs(lin.w_mul - dlatents.shape[-1] ** -0.5) > 1e-9 or lin.b_mul != 1:
                    raise NotImplementedError("StyleMod.lin is expected to be an equalised-lr layer with gain 1 (reference :129)")
                entries.append(dict(mod_w=lin.weight, mod_b=lin.bias, wsq=None, cin=lin.weight.shape[0], cout=0, lat_idx=lat_idx,
                                    s_off=off, d_off=0, wscale=1.0))
                e._s_off = off
                off += lin.weight.shape[0]
            self._table = dict(key=key, table=_style_table(entries, dev), n=len(entries), total=off,
                               max_rows=max(en["cin"] for en in entries))
        t = self._table
        s = th.empty((dlatents.shape[0], t["total"]), dtype=th.float32, device=dev)
        _lib.check(lib.maua_style_affine_f32(dlatents.data_ptr(), dlatents.shape[0], dlatents.shape[1], dlatents.shape[2], None,
                                             None, t["table"].data_ptr(), t["n"], t["max_rows"], s.data_ptr(), t["total"],
                                             None, _lib.stream_ptr(dev)), "maua_style_affine_f32")
        return s, t["total"]

    def run(self, dlatents, noise):
        """dlatents [B, >= 2 n_blocks, 512] on the device; noise: one tensor ([B or 1, 1, h, w]) or None per block."""
        lib = _lib.load()
        dev = dlatents.device
        b = dlatents.shape[0]
        styles, s_stride = self._styles(dlatents)
        ones_cache = {}

        def ones(c):
            if c not in ones_cache:
                ones_cache[c] = th.ones((b, c), dtype=th.float32, device=dev)
            return ones_cache[c]

        x = None
        for i, block in enumerate(self.blocks.values()):
            nz = noise[i] if noise is not None else None
            if i == 0:
                x = block.const.expand(b, -1, -1, -1).contiguous()
                x = block.epi1.run(x, block.bias, nz, styles, getattr(block.epi1, "_s_off", 0), s_stride)
                x = block.conv.run(x, ones(x.shape[1]))
                x = block.epi2.run(x, block.conv.bias_vector(), nz, styles, getattr(block.epi2, "_s_off", 0), s_stride)
            else:
                x = block.conv0_up.run(x, ones(x.shape[1]))
                x = block.epi1.run(x, block.conv0_up.bias_vector(), nz, styles, getattr(block.epi1, "_s_off", 0), s_stride)
                x = block.conv1.run(x, ones(x.shape[1]))
                x = block.epi2.run(x, block.conv1.bias_vector(), nz, styles, getattr(block.epi2, "_s_off", 0), s_stride)
        t = self.torgb
        rgb = th.empty((b, t.weight.shape[0], x.shape[2], x.shape[3]), dtype=th.float32, device=dev)
        if t.weight.shape[0] != 3:
            raise NotImplementedError("torgb is built for 3 colour channels")
        _lib.check(lib.maua_torgb_f32(x.data_ptr(), t.weight.data_ptr(), ones(x.shape[1]).data_ptr(), x.shape[1],
                                      _lib.ptr(t.bias_vector()), None, None, rgb.data_ptr(), b, x.shape[1], x.shape[2], x.shape[3],
                                      float(t.w_mul), _lib.stream_ptr(dev)), "maua_torgb_f32")
        return rgb

    def forward(self, dlatents_in, noise):
        dl = _lib.require_cuda(dlatents_in, "dlatents_in")
        with th.cuda.device(dl.device):
            return self.run(dl, [noise] * len(self.blocks) if not isinstance(noise, (list, tuple)) else list(noise))


class G_style(nn.Sequential):
    """reference :503-617 — what ``load_generator(..., is_stylegan1=True)`` builds (generate_audiovisual.py:41-42)."""

    def __init__(self, output_size=1920, checkpoint=None, network_resolution=None):
        """``network_resolution`` (not in the reference): build the synthesis network at this resolution without probing a
        checkpoint — the ranks of a multi-GPU job that do not read the checkpoint get it from rank 0, so that every rank has
        the same blocks, constant and noise-buffer shapes before the weights are broadcast."""
        super().__init__()
        self.g_mapping = G_mapping()
        state = th.load(checkpoint, map_location="cpu") if checkpoint is not None else None
        candidates = (1024, 512, 256, 128) if network_resolution is None else (int(network_resolution),)
        network_resolution = None
        for resolution in candidates:  # the checkpoint decides: the first resolution whose shapes fit (:509-537)
            self.g_synthesis = G_synthesis(resolution=resolution)
            try:
                if state is not None:
                    self.load_state_dict(state, strict=False)
                network_resolution = resolution
                break
            except RuntimeError:
                print(f"Trying {resolution // 2}px generator resolution..." if resolution > 128 else
                      "ERROR: Network too small or state_dict mismatch")
        if network_resolution is None:
            raise SystemExit(1)
        self.network_resolution = network_resolution
        block0 = getattr(self.g_synthesis.blocks, "4x4")
        const = block0.const
        if network_resolution != 1024:  # a smaller network still renders 1024 px: larger random constant (:540-542)
            side = int(4 * 1024 / network_resolution)
            means = th.zeros(size=(1, 512, side, side))
            const = th.normal(mean=means, std=th.ones_like(means) * const.std())
        _, _, ch, cw = const.shape
        if output_size == 1920:
            layer0 = th.cat([const[:, :, :, [0]], const[:, :, :, [0]], const, const[:, :, :, [-1]], const[:, :, :, [-1]]], axis=3)
        elif output_size == 512:
            layer0 = const[:, :, ch // 4: 3 * ch // 4, cw // 4: 3 * cw // 4]
        else:
            layer0 = const
        block0.const = nn.Parameter(layer0 + th.normal(0, const.std() / 2.0))
        _, _, height, width = block0.const.shape
        for i in range(len(self.g_synthesis.blocks)):
            self.register_buffer(f"noise_{i}", th.randn(1, 1, height * 2 ** i, width * 2 ** i))
        self.truncation_latent = self.mean_latent(2 ** 14)

    def mean_latent(self, n_latent):
        dev = self.g_mapping.dense0.weight.device
        return self.g_mapping(th.randn(n_latent, 512, device=dev)).mean(0, keepdim=True)

    capturable_bends = False  # G_style takes no transform_dict_list (reference :584-617): bends keep the eager path

    def weights_key(self):
        """Identity of everything a captured forward has baked in as pointers (see Generator.weights_key)."""
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def capture_graph(self, batch, lane=0, frames_u8=True, bends=()):
        """One forward of ``batch`` frames (+ the uint8 frame epilogue) captured into a hipGraph: the render loop replays it per
        batch instead of issuing the ~100 launches of the eager synthesis (render.synthesize; same lane protocol as the StyleGAN2
        generator's ``capture_graph``)."""
        if bends:
            raise NotImplementedError("G_style has no network-bending hook")
        return StyleGAN1Lane(self, batch, lane)

    def forward(self, styles, noise=None, truncation=1, map_latents=False, randomize_noise=False, input_is_latent=True,
                transform_dict_list=None):
        if map_latents:
            return self.g_mapping(styles)
        dev = self.g_mapping.dense0.weight.device
        if dev.type != "cuda":
            raise RuntimeError("G_style must live on a HIP device (.cuda()); the MI355X path has no CPU fallback")
        n_blocks = len(self.g_synthesis.blocks)
        noise = list(noise) if noise is not None else [None] * n_blocks
        for ns in range(len(noise)):
            if noise[ns] is None and hasattr(self, f"noise_{ns}"):
                noise[ns] = getattr(self, f"noise_{ns}")
        styles = _lib.require_cuda(styles.to(dev), "styles")
        is_one = isinstance(truncation, (int, float)) and truncation == 1
        if not is_one:  # lerp toward the truncation latent on the first 8 layers (:593-596)
            weight = truncation if isinstance(truncation, (int, float)) else truncation.to(dev).reshape(-1, 1, 1)
            interp = th.lerp(self.truncation_latent.to(dev).expand_as(styles), styles, weight)
            do_trunc = (th.arange(styles.size(1), device=dev) < 8).view(1, -1, 1)
            styles = th.where(do_trunc, interp, styles).contiguous()
        with th.cuda.device(dev):
            img = self.g_synthesis.run(styles.contiguous(), noise)
        return img, None


class StyleGAN1Lane:
    """A captured G_style forward.  Unlike the StyleGAN2 lanes (whose kernels read the HBM-resident sequences through a frame
    source), the StyleGAN1 synthesis keeps static input tensors: ``replay`` copies the batch's slices into them (StyleGAN1 is the
    compatibility path of generate(), not the measured one) and launches the graph; the capture itself is a hipGraph stream
    capture driven through torch.cuda.CUDAGraph, so the few torch ops of G_style.forward (truncation lerp, constant expand) and
    their allocations are part of it."""

    def __init__(self, g, batch, lane):
        from ..render import frames_to_uint8

        dev = g.g_mapping.dense0.weight.device
        self.generator, self.batch, self.lane = g, batch, lane
        g.truncation_latent = g.truncation_latent.to(dev)  # (a plain attribute: module.cuda() leaves it on the host, and an upload cannot be captured)
        self.weights_key = g.weights_key()
        n_blocks = len(g.g_synthesis.blocks)
        # (G_mapping broadcasts to 18 latents whatever the network resolution, :357-362; the synthesis reads the first 2 n_blocks)
        self._latents = th.zeros(batch, g.truncation_latent.shape[1], g.truncation_latent.shape[2], device=dev)
        self._trunc = th.ones(batch, device=dev)
        self._noise = []
        for i in range(n_blocks):
            buf = getattr(g, f"noise_{i}")
            self._noise.append(buf.to(dev, th.float32).expand(batch, -1, -1, -1).contiguous())
        self._bound = None
        stream = th.cuda.current_stream(dev)

        def forward():
            images, _ = g(styles=self._latents, noise=list(self._noise), truncation=self._trunc, input_is_latent=True)
            return images

        images = forward()  # warm-up: weight packs, style tables, function attributes — nothing of it may happen under capture
        self.u8 = th.empty((batch, images.shape[2], images.shape[3], 3), dtype=th.uint8, device=dev)
        frames_to_uint8(images, self.u8)
        stream.synchronize()
        self.graph = th.cuda.CUDAGraph()
        with th.cuda.graph(self.graph, stream=stream):
            self.image = forward()
            frames_to_uint8(self.image, self.u8)

    def bind(self, latents, noise, truncation=None):
        """``latents`` [N, 18, 512], ``noise`` one [N or 1, 1, h, w] sequence or None (= the generator's buffer) per block,
        ``truncation`` [N] or None (= 1): the sequences ``replay`` slices."""
        noise = list(noise) + [None] * (len(self._noise) - len(noise))
        for i, (static, seq) in enumerate(zip(self._noise, noise)):
            if seq is None:
                static.copy_(getattr(self.generator, f"noise_{i}").to(static.device).expand_as(static))
            elif seq.dim() != 4 or tuple(seq.shape[1:]) != tuple(static.shape[1:]):
                raise RuntimeError(f"noise {tuple(seq.shape)} does not match feature map {tuple(static.shape)} of block {i}")
            elif seq.shape[0] == 1:
                static.copy_(seq.expand_as(static))
        if truncation is None:
            self._trunc.fill_(1.0)
        self._bound = (latents, noise, truncation)

    def release(self):
        self._bound = None

    def replay(self, frame0, stream=None):
        latents, noise, truncation = self._bound
        hi = frame0 + self.batch
        self._latents.copy_(latents[frame0:hi])
        for static, seq in zip(self._noise, noise):
            if seq is not None and seq.shape[0] != 1:
                static.copy_(seq[frame0:hi])
        if truncation is not None:
            self._trunc.copy_(truncation[frame0:hi])
        self.graph.replay()
