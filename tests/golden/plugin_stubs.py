"""Deterministic stand-ins for the parts of the audio-reactive path that cannot be pinned to the reference here
(librosa / madmom are absent, torch's device RNG differs between builds), shared by tests/golden/make_golden.py — which
runs the REFERENCE's default plugin and ``generate()`` with them — and by the GPU tests, which run this repo's plugin and
``generate()`` with the very same stand-ins and compare against what the reference produced.

Nothing here is reference code: seeded envelopes, a chromagram, an audio loader that returns noise of the right length,
and a ``torch.randn`` whose values depend only on (seed, shape, how many times that shape was asked for).
"""
import zlib

import numpy as np
import torch


def envelopes(n_frames, seed=40):
    """(lo_onsets, hi_onsets) float32 [n_frames] in [0, 1] with a few sharp peaks, chroma float32 [n_frames, 12] whose rows
    sum to 1 — the shapes/ranges ``ar.onsets`` / ``ar.chroma`` return (reference audioreactive/signal.py:31-73,136-156)."""
    rng = np.random.default_rng([seed, n_frames])
    t = np.arange(n_frames)

    def peaks(period, width, phase):
        d = ((t + phase) % period).astype(np.float64)
        env = np.exp(-d / width) * (0.6 + 0.4 * rng.random(n_frames))
        env = np.clip(env / env.max(), 0, 1) ** 2
        return env.astype(np.float32)

    lo = peaks(max(n_frames // 6, 4), 2.5, 1)
    hi = peaks(max(n_frames // 11, 3), 1.2, 0)
    walk = np.cumsum(rng.standard_normal((n_frames, 12)) * 0.15, axis=0)
    ch = np.exp(walk - walk.max(axis=1, keepdims=True))
    ch = (ch / ch.sum(axis=1, keepdims=True)).astype(np.float32)
    return lo, hi, ch


class Features:
    """``onsets`` / ``chroma`` / ``load_audio`` replacements bound to one clip length."""

    def __init__(self, n_frames, fps=30, seed=40, sr=22050):
        self.n_frames, self.fps, self.sr = n_frames, fps, sr
        self.lo, self.hi, self.ch = envelopes(n_frames, seed)
        self.calls = []

    def onsets(self, audio, sr, n_frames, margin=8, fmin=20, fmax=8000, smooth=1, clip=100, power=1, type="mm", **kw):
        assert n_frames == self.n_frames
        self.calls.append(("onsets", fmin, fmax, smooth, clip, power))
        return torch.from_numpy((self.lo if fmax < 1000 else self.hi).copy())

    def chroma(self, audio, sr, n_frames, margin=16, type="cens", notes=12, **kw):
        assert n_frames == self.n_frames
        self.calls.append(("chroma", margin, type, notes))
        return torch.from_numpy(self.ch[:, :notes].copy())

    def load_audio(self, audio_file, offset=0, duration=-1, cache=True):
        dur = self.n_frames / self.fps
        audio = np.random.default_rng(5).standard_normal(int(round(dur * self.sr))).astype(np.float32) * 0.1
        return audio, self.sr, dur


class SeededRandn:
    """Replacement for ``torch.randn``: the values of the k-th request for a given shape come from
    numpy default_rng([seed, crc32(shape), k]) — independent of the device, the torch build and of unrelated draws
    (module constructors) in between.  ``device="cuda"`` is honoured when a GPU is present, ignored otherwise (the
    reference's plugins hard-code it, examples/default.py:32-36)."""

    def __init__(self, seed=41):
        self.seed = seed
        self.count = {}

    def __call__(self, *size, device=None, dtype=None, **kw):
        if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)):
            size = tuple(size[0])
        size = tuple(int(v) for v in size)
        k = self.count.get(size, 0)
        self.count[size] = k + 1
        rng = np.random.default_rng([self.seed, zlib.crc32(repr(size).encode()), k])
        out = torch.from_numpy(rng.standard_normal(size, dtype=np.float32))
        if device is not None and torch.device(device).type == "cuda" and torch.cuda.is_available():
            out = out.to(device)
        return out


def summary(t):
    """Small, comparable description of a big tensor: moments + a strided subsample (first dim dense-ish)."""
    a = t.detach().float().cpu().numpy()
    flat = a.reshape(a.shape[0], -1)
    step = max(flat.shape[1] // 97, 1)
    return {
        "stats": np.array([a.mean(), a.std(), a.min(), a.max()], dtype=np.float64),
        "frame_mean": flat.mean(axis=1).astype(np.float64),
        "sub": flat[:: max(a.shape[0] // 40, 1), ::step][:, :97].astype(np.float32).copy(),
    }
