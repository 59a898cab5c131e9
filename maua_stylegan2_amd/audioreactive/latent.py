"""Latent / noise helpers with the reference's ``audioreactive.latent`` surface.

Mirrors /root/reference/audioreactive/latent.py: chroma_weight_latents :15-26 · slerp :29-45 · slerp_loops :48-83 ·
spline_loops :86-107 · wrapping_slice :110-133 · generate_latents :136-159 · save/load_latents :162-181 ·
perlin_noise :188-246.  perlin_noise runs as one HIP kernel (csrc/signal.hip perlin3d_kernel) from the same numpy-RNG
gradient angles the reference draws; generate_latents maps z through the mapping network of the MI355X generator
(the evident intent — the reference's map_latents branch normalises over a singleton axis).
"""
import numpy as np
import torch as th
from scipy import interpolate

from .. import _lib
from .signal import gaussian_filter


def chroma_weight_latents(chroma, latents):
    """[n_frames, notes] x [notes, n_latent, 512] -> [n_frames, n_latent, 512]."""
    return th.einsum("tn,nld->tld", chroma.to(latents.device, latents.dtype), latents)


def slerp(val, low, high):
    """Great-circle interpolation between two vectors (reference :29-45).  ``val`` may be a scalar or an array of
    fractions: all of them are evaluated at once, one row per fraction."""
    low, high = np.asarray(low), np.asarray(high)
    frac = np.asarray(val, dtype=np.float64)
    if frac.ndim:
        frac = frac.reshape(frac.shape + (1,) * low.ndim)
    cosine = np.dot(low.ravel() / np.linalg.norm(low), high.ravel() / np.linalg.norm(high))
    angle = np.arccos(np.clip(cosine, -1.0, 1.0))
    sine = np.sin(angle)
    if sine == 0:  # parallel vectors: the geodesic degenerates to the straight line
        return low + frac * (high - low)
    return (np.sin((1.0 - frac) * angle) * low + np.sin(frac * angle) * high) / sine


def slerp_loops(latent_selection, n_frames, n_loops, smoothing=1, loop=True):
    """Looping latent sequence along great circles between consecutive selection entries (reference :48-83): every key
    gets n_frames // n_loops // n_keys frames, the loop is Gaussian-smoothed along time and tiled to ``n_frames``.
    Differences from the reference, both needed for it to run at all or off 1024 px: the float64 interpolant is cast to
    float32 before the filter (the reference's conv1d rejects it) and the layer axis is
    the selection's own instead of a hard-coded 18."""
    keys = np.asarray(latent_selection)
    n_layers = keys.shape[1]
    if loop:
        keys = np.concatenate([keys, keys[:1]])
    fractions = np.linspace(0.0, 1.0, int(n_frames // max(1, n_loops) // len(keys)))
    legs = [slerp(fractions, keys[k][0], keys[(k + 1) % len(keys)][0]) for k in range(len(keys))]
    cycle = gaussian_filter(th.from_numpy(np.concatenate(legs)).float(), smoothing)
    frames = cycle.repeat(int(n_frames / len(cycle)), 1)
    if len(frames) != n_frames:
        frames = th.cat([frames, frames[: n_frames - len(frames)]])
    return frames[:, None, :].repeat(1, n_layers, 1)


def spline_loops(latent_selection, n_frames, n_loops, loop=True):
    latent_selection = np.asarray(latent_selection)
    if loop:
        latent_selection = np.concatenate([latent_selection, latent_selection[[0]]])
    x = np.linspace(0, 1, int(n_frames // max(1, n_loops)))
    knots = np.linspace(0, 1, latent_selection.shape[0])
    flat = latent_selection.reshape(latent_selection.shape[0], -1)
    base = np.zeros((len(x), flat.shape[1]))
    for j in range(flat.shape[1]):
        base[:, j] = interpolate.splev(x, interpolate.splrep(knots, flat[:, j]))
    base = th.from_numpy(base.reshape((len(x),) + latent_selection.shape[1:]))
    out = th.cat([base] * int(n_frames / len(base)), axis=0)
    if n_frames - len(out) > 0:
        out = th.cat([out, out[0: n_frames - len(out)]])
    return out[:n_frames]


def wrapping_slice(tensor, start, length, return_indices=False):
    """``length`` entries of ``tensor`` from ``start``, continuing at the beginning when the end is reached (reference
    :110-133).  Like the reference the slice wraps at most once — a request that would lap the tensor again ends at
    (start + length) mod n — and a slice that starts AT the end has no head: it is the first (start + length) mod n entries.
    A start beyond the end raises, as the reference's ``arange(start, n)`` does."""
    n = tensor.shape[0]
    end = start + length
    if start > n and end > n:
        raise RuntimeError(f"wrapping_slice: start {start} lies beyond the tensor ({n} entries)")
    head = length if end <= n else max(n - start, 0)   # entries taken before the wrap
    tail = 0 if end <= n else end % n                  # entries taken from the beginning
    k = th.arange(head + tail)
    indices = th.where(k < head, k + start, k - head)
    if n == 1:
        indices = indices.new_zeros(1)
    return indices if return_indices else tensor[indices]


_MAPPING_CACHE = {}  # (checkpoint path, mtime, size, latent_dim, n_mlp, device) -> mapping network on the device (8 MB; one entry)


def _mapping_network(ckpt, latent_dim, n_mlp):
    """The checkpoint's mapping network (``style.*``) on the current device.  Built ON the device without initial values when a checkpoint
    will overwrite every tensor (eight 512 x 512 normal draws + scalings on the CPU were 0.05 s of a 0.9 s job), and kept for the next job on
    the same checkpoint FILE (path, mtime, size) — the generator itself is kept the same way (generate_audiovisual._cached_generator)."""
    import os

    from ..models import stylegan2 as sg2

    dev = th.device("cuda", th.cuda.current_device())
    key = None
    if ckpt is not None:
        try:
            st = os.stat(ckpt)
            key = (os.path.realpath(ckpt), st.st_mtime_ns, st.st_size, latent_dim, n_mlp, dev.index)
        except OSError:
            key = None
        if key is not None and _MAPPING_CACHE.get("key") == key:
            return _MAPPING_CACHE["net"]
    sg2._SKIP_INIT.on = ckpt is not None
    try:
        with th.device(dev):
            mapping = th.nn.Sequential(sg2.PixelNorm(), *[sg2.EqualLinear(latent_dim, latent_dim, lr_mul=0.01, activation="fused_lrelu")
                                                          for _ in range(n_mlp)])
    finally:
        sg2._SKIP_INIT.on = False
    if ckpt is not None:
        try:  # zip-format checkpoints are mapped: only the pages of the eight style.* matrices are ever read
            weights = th.load(ckpt, map_location="cpu", mmap=True)["g_ema"]
        except (RuntimeError, ValueError, TypeError):
            weights = th.load(ckpt, map_location="cpu")["g_ema"]
        mapping.load_state_dict({k[len("style."):]: v for k, v in weights.items() if k.startswith("style.")}, strict=True)
        del weights
    mapping = mapping.to(dev)
    if key is not None:
        _MAPPING_CACHE.clear()
        _MAPPING_CACHE.update(key=key, net=mapping)
    return mapping


def generate_latents(n_latents, ckpt, G_res, noconst=False, latent_dim=512, n_mlp=8, channel_multiplier=2):
    """``n_latents`` random w vectors, each repeated over the generator's layers: [n_latents, n_latent, latent_dim] on the
    CPU (reference :136-159).  Only the mapping network is needed for that, so only its ``style.*`` tensors are read from
    the checkpoint and put on the device (the reference builds and uploads a second full generator).  z is mapped on
    [N, latent_dim] — the evident intent; the reference's map_latents branch normalises over a singleton axis."""
    mapping = _mapping_network(ckpt, latent_dim, n_mlp)
    w = mapping(th.randn((n_latents, latent_dim), device="cuda"))
    n_layers = 2 * int(np.log2(G_res)) - 2
    return w[:, None, :].repeat(1, n_layers, 1).cpu()


def save_latents(latents, filename):
    np.save(filename, latents.numpy() if isinstance(latents, th.Tensor) else np.asarray(latents))


def load_latents(filename):
    return th.from_numpy(np.load(filename))


def perlin_gradients(res, tileable=(True, False, False), rng=None):
    """Unit gradient lattice [r0+1, r1+1, r2+1, 3] from two uniform angle draws (reference :209-218)."""
    rand = np.random.rand if rng is None else rng.random
    theta = 2 * np.pi * rand(res[0] + 1, res[1] + 1, res[2] + 1)
    phi = 2 * np.pi * rand(res[0] + 1, res[1] + 1, res[2] + 1)
    g = np.stack((np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)), axis=3)
    if tileable[0]:
        g[-1, :, :] = g[0, :, :]
    if tileable[1]:
        g[:, -1, :] = g[:, 0, :]
    if tileable[2]:
        g[:, :, -1] = g[:, :, 0]
    return g.astype(np.float32)


def perlin_noise(shape, res, tileable=(True, False, False), interpolant=None, gradients=None):
    """3-D Perlin noise tensor of ``shape`` on the current HIP device (reference :188-246; quintic fade only)."""
    if interpolant is not None:
        raise NotImplementedError("the HIP Perlin kernel implements the default quintic interpolant")
    if any(s % r for s, r in zip(shape, res)):
        raise ValueError("shape must be a multiple of res")
    lib = _lib.load()
    dev = th.device("cuda", th.cuda.current_device())
    g = perlin_gradients(res, tileable) if gradients is None else np.asarray(gradients, dtype=np.float32)
    gd = th.from_numpy(np.ascontiguousarray(g)).to(dev)
    out = th.empty(tuple(shape), dtype=th.float32, device=dev)
    with th.cuda.device(dev):
        _lib.check(lib.maua_perlin3d_f32(gd.data_ptr(), out.data_ptr(), shape[0], shape[1], shape[2], res[0], res[1],
                                         res[2], _lib.stream_ptr(dev)), "maua_perlin3d_f32")
    return out
