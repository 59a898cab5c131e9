"""Network bending with the reference's ``audioreactive.bend`` surface, on one HIP warp kernel instead of kornia.

Mirrors /root/reference/audioreactive/bend.py: NetworkBend :12-26 · AddNoise :29-41 · Print :44-49 · Translate :52-72 ·
Zoom :75-87 · Rotate :90-102.  The reference composes ReflectionPad2d -> kornia Translate/Scale/Rotate -> kornia
CenterCrop (three full-size intermediates); here the composition is evaluated per OUTPUT pixel by
maua_affine_reflect_warp_f32: inverse affine map into the padded canvas, bilinear taps, reflection folded into the
index.  kornia is an un-pinned, un-vendored dependency (requirements.txt:8) -> "parity unpinned" (DESIGN.md); the
conventions implemented are kornia's documented ones: pixel-unit translation, transforms about the canvas centre
((W-1)/2, (H-1)/2), bilinear, zeros outside the canvas, anticlockwise-positive angles in degrees.
"""
import math

import torch as th

from .. import _lib


class NetworkBend(th.nn.Module):
    """reference :12-26.  ``modulation`` is the batch slice in the reference's render loop (render.py:151-158); the MI355X
    render loop hands over the modulation of the WHOLE sequence once and a captured forward picks the frame's row on the
    device (``run_static``), which is what makes bends hipGraph-capturable."""

    def __init__(self, sequential_fn, modulation):
        super().__init__()
        self.sequential = sequential_fn(modulation)

    def forward(self, x):
        return self.sequential(x)

    def run_static(self, x, out, src):
        if not hasattr(self.sequential, "run_static"):
            raise RuntimeError(f"{type(self.sequential).__name__} cannot run inside a captured forward")
        return self.sequential.run_static(x, out, src)

    @property
    def capturable(self):
        return hasattr(self.sequential, "run_static")

    @property
    def sequence_rows(self):
        """Rows of per-frame parameters the captured form indexes with frame0 + b (None: nothing per-frame); the render loop
        requires 1 or the number of frames of the sequence before it captures (a shorter table would be read out of bounds)."""
        return getattr(self.sequential, "sequence_rows", None)


class AddNoise(th.nn.Module):
    def __init__(self, noise):
        super().__init__()
        self.noise = noise

    def forward(self, x):
        return x + self.noise.to(x.device)

    def run_static(self, x, out, src):
        """Capturable form: writes into the static buffer ``out`` (no allocation; the noise is static for the whole render)."""
        if self.noise.device != x.device or self.noise.dtype != x.dtype:
            self.noise = self.noise.to(x.device, x.dtype)
        return th.add(x, self.noise, out=out)


class Print(th.nn.Module):
    def forward(self, x):
        print(x.shape, [x.min().item(), x.mean().item(), x.max().item()], th.std(x).item())
        return x


def reflection_chain_index(n, pad_pairs):
    """Source index of every element of an axis of length ``n`` after a CHAIN of reflection pads [(before, after), ...]
    applied one after the other, as stacked ``ReflectionPad2d`` modules do: each pad mirrors the canvas built so far, so
    the composition is in general not one triangular fold of the source (reference bend.py:60-64 stacks three)."""
    idx = list(range(n))
    for before, after in pad_pairs:
        if before >= len(idx) or after >= len(idx):
            raise ValueError(f"reflection pad ({before}, {after}) must be smaller than the axis ({len(idx)})")
        idx = idx[before:0:-1] + idx + idx[-2:-2 - after:-1] if before or after else idx
    return idx


class AffineReflectWarp(th.nn.Module):
    """y = CenterCrop(h, w)( Affine( ReflectionPad(x) [+ noise] ) ) with a per-sample inverse map ``m`` [B, 6].
    ``pads`` is one (left, right, top, bottom) tuple or a list of them (pads stacked in that order)."""

    def __init__(self, inv_maps, pads, noise=None):
        super().__init__()
        self.inv_maps = inv_maps
        chain = [tuple(pads)] if isinstance(pads[0], int) else [tuple(p) for p in pads]
        self.chain = chain
        self.pads = tuple(sum(p[i] for p in chain) for i in range(4))  # total (left, right, top, bottom)
        self.noise = noise
        self._maps = None  # (h, w, device) -> int32 index tables, only for a real chain
        self._dev = None   # (device, batch) -> device-resident operands of the launch

    @property
    def sequence_rows(self):
        return int(self.inv_maps.shape[0])

    def _operands(self, x, per_frame):
        """Device-resident operands, built on the first call for a (device, batch) and reused (so that a captured forward
        performs no allocation): inverse maps (the whole sequence when ``per_frame``, else expanded to the batch), the canvas
        noise plane, the reflection-chain index tables."""
        b, c, h, w = x.shape
        key = (str(x.device), b, per_frame, h, w)
        if self._dev is None or self._dev[0] != key:
            m = self.inv_maps.to(x.device, th.float32).contiguous()
            if not per_frame:
                if m.shape[0] == 1 and b > 1:
                    m = m.expand(b, 6).contiguous()
                if m.shape != (b, 6):
                    raise RuntimeError(f"expected {b} inverse affine maps, got {tuple(m.shape)}")
            nz = None
            if self.noise is not None:
                nz = _lib.require_cuda(self.noise.to(x.device).float(), "noise")
                pl, pr, pt, pb = self.pads
                if nz.numel() != (h + pt + pb) * (w + pl + pr):
                    raise RuntimeError("bend noise must have the size of one padded canvas plane")
            xmap = ymap = None
            if len(self.chain) > 1:
                xs = reflection_chain_index(w, [(p[0], p[1]) for p in self.chain])
                ys = reflection_chain_index(h, [(p[2], p[3]) for p in self.chain])
                xmap, ymap = th.tensor(xs, dtype=th.int32, device=x.device), th.tensor(ys, dtype=th.int32, device=x.device)
            self._dev = (key, m, nz, xmap, ymap)
        return self._dev[1:]

    def _launch(self, x, y, m, nz, xmap, ymap, src):
        b, c, h, w = x.shape
        with th.cuda.device(x.device):
            _lib.check(_lib.load().maua_affine_reflect_warp_mapped_f32(
                x.data_ptr(), m.data_ptr(), y.data_ptr(), b, c, h, w, self.pads[0], self.pads[1], self.pads[2], self.pads[3],
                _lib.ptr(nz), _lib.ptr(xmap), _lib.ptr(ymap), src, _lib.stream_ptr(x.device)), "maua_affine_reflect_warp_mapped_f32")
        return y

    def forward(self, x):
        x = _lib.require_cuda(x, "x")
        m, nz, xmap, ymap = self._operands(x, per_frame=False)
        return self._launch(x, th.empty_like(x), m, nz, xmap, ymap, None)

    def run_static(self, x, out, src):
        """Inside a captured forward: ``inv_maps`` holds one row per frame of the render (or a single static row) and sample b
        uses row frame0 + b, read on the device through the frame source ``src`` — nothing is rebuilt per batch."""
        per_frame = self.inv_maps.shape[0] != 1
        m, nz, xmap, ymap = self._operands(x, per_frame=per_frame)
        return self._launch(x, out, m, nz, xmap, ymap, src if per_frame else None)


def _inverse_maps_translate(t):
    """dst = src + t  ->  src = dst - t (pixels)."""
    t = t.reshape(-1, 2).float()
    m = th.zeros(t.shape[0], 6, device=t.device)  # stays where the modulation lives (HBM during a render)
    m[:, 0] = 1.0
    m[:, 4] = 1.0
    m[:, 2] = -t[:, 0]
    m[:, 5] = -t[:, 1]
    return m


def _inverse_maps_scale(s, cw, ch):
    s = s.float()
    if s.dim() == 1:
        s = s[:, None].expand(-1, 2)
    cx, cy = (cw - 1) / 2.0, (ch - 1) / 2.0
    m = th.zeros(s.shape[0], 6, device=s.device)
    m[:, 0] = 1.0 / s[:, 0]
    m[:, 4] = 1.0 / s[:, 1]
    m[:, 2] = cx - cx / s[:, 0]
    m[:, 5] = cy - cy / s[:, 1]
    return m


def _inverse_maps_rotate(angle_deg, cw, ch):
    a = th.deg2rad(angle_deg.float().reshape(-1))
    cx, cy = (cw - 1) / 2.0, (ch - 1) / 2.0
    cos, sin = th.cos(a), th.sin(a)
    # forward (OpenCV/kornia get_rotation_matrix2d): dst = R(src - c) + c with R = [[cos, sin], [-sin, cos]];
    # inverse: src = R^T (dst - c) + c
    m = th.zeros(a.shape[0], 6, device=a.device)
    m[:, 0], m[:, 1] = cos, -sin
    m[:, 3], m[:, 4] = sin, cos
    m[:, 2] = cx - cos * cx + sin * cy
    m[:, 5] = cy - sin * cx - cos * cy
    return m


class Translate(NetworkBend):
    """Horizontal scrolling (reference :52-72): three STACKED reflection pads (w/2 | w/2, then w | w, then w | 0 — each
    mirrors the canvas built so far, which is what makes a scroll of w pixels land on the same features), add noise,
    translate, centre crop."""

    def __init__(self, modulation, h, w, noise):
        pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]
        sequential_fn = lambda b: AffineReflectWarp(_inverse_maps_translate(b), pads, noise)  # noqa: E731
        super().__init__(sequential_fn, modulation)


class Zoom(NetworkBend):
    def __init__(self, modulation, h, w):
        padding = int(max(h, w)) - 1
        pads = (padding,) * 4
        sequential_fn = lambda b: AffineReflectWarp(_inverse_maps_scale(b, w + 2 * padding, h + 2 * padding), pads)  # noqa: E731
        super().__init__(sequential_fn, modulation)


class Rotate(NetworkBend):
    def __init__(self, modulation, h, w):
        padding = int(max(h, w) * (1 - math.sqrt(2) / 2))
        pads = (padding,) * 4
        sequential_fn = lambda b: AffineReflectWarp(_inverse_maps_rotate(b, w + 2 * padding, h + 2 * padding), pads)  # noqa: E731
        super().__init__(sequential_fn, modulation)
