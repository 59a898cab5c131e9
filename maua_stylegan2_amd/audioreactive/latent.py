"""Latent / noise helpers with the reference's ``audioreactive.latent`` surface.

Mirrors /root/reference/audioreactive/latent.py: chroma_weight_latents :15-26 · slerp :29-45 · slerp_loops :48-83 ·
spline_loops :86-107 · wrapping_slice :110-133 · generate_latents :136-159 · save/load_latents :162-181 ·
perlin_noise :188-246.  perlin_noise runs as one HIP kernel (csrc/signal.hip perlin3d_kernel) from the same numpy-RNG
gradient angles the reference draws; generate_latents maps z through the mapping network of the MI355X generator
(the evident intent — the reference's map_latents branch normalises over a singleton axis).
"""
import gc

import numpy as np
import torch as th
from scipy import interpolate

from .. import _lib
from .signal import gaussian_filter


def chroma_weight_latents(chroma, latents):
    """[n_frames, notes] x [notes, n_latent, 512] -> [n_frames, n_latent, 512]."""
    return th.einsum("tn,nld->tld", chroma.to(latents.device, latents.dtype), latents)


def slerp(val, low, high):
    omega = np.arccos(np.clip(np.dot(low / np.linalg.norm(low), high / np.linalg.norm(high)), -1, 1))
    so = np.sin(omega)
    if so == 0:
        return (1.0 - val) * low + val * high
    return np.sin((1.0 - val) * omega) / so * low + np.sin(val * omega) / so * high


def slerp_loops(latent_selection, n_frames, n_loops, smoothing=1, loop=True):
    """Reference :48-83, with the float64 -> float32 cast it needs to get through gaussian_filter."""
    latent_selection = np.asarray(latent_selection)
    n_lat = latent_selection.shape[1]
    if loop:
        latent_selection = np.concatenate([latent_selection, latent_selection[[0]]])
    base = []
    for n in range(len(latent_selection)):
        for val in np.linspace(0.0, 1.0, int(n_frames // max(1, n_loops) // len(latent_selection))):
            base.append(th.from_numpy(slerp(val, latent_selection[n % len(latent_selection)][0],
                                            latent_selection[(n + 1) % len(latent_selection)][0])))
    base = th.stack(base).float()
    base = gaussian_filter(base, smoothing)
    base = th.cat([base] * int(n_frames / len(base)), axis=0)
    base = th.cat([base[:, None, :]] * n_lat, axis=1)
    if n_frames - len(base) != 0:
        base = th.cat([base, base[0: n_frames - len(base)]])
    return base


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
    if start + length <= tensor.shape[0]:
        indices = th.arange(start, start + length)
    else:
        indices = th.cat((th.arange(start, tensor.shape[0]), th.arange(0, (start + length) % tensor.shape[0])))
    if tensor.shape[0] == 1:
        indices = th.zeros(1, dtype=th.int64)
    if return_indices:
        return indices
    return tensor[indices]


def generate_latents(n_latents, ckpt, G_res, noconst=False, latent_dim=512, n_mlp=8, channel_multiplier=2):
    from ..models.stylegan2 import Generator

    generator = Generator(G_res, latent_dim, n_mlp, channel_multiplier=channel_multiplier, constant_input=not noconst,
                          checkpoint=ckpt).cuda()
    zs = th.randn((n_latents, latent_dim), device="cuda")
    latent_selection = generator(zs, map_latents=True).cpu()
    del generator, zs
    gc.collect()
    th.cuda.empty_cache()
    return latent_selection


def save_latents(latents, filename):
    np.save(filename, latents)


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
