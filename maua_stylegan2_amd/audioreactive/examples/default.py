"""Default audio-reactive plugin for the MI355X path.

Behaviour contract = /root/reference/audioreactive/examples/default.py:6-45 (what ``--audioreactive_file`` defaults
to): two band-limited onset envelopes (bass <= 150 Hz, treble >= 500 Hz) computed once in ``initialize``; latents = the
12-bin chromagram weighting of the latent selection, smoothed, with onset-driven jumps toward two fixed selection
entries; noise (scales up to 256 px) = a fast and a slow temporally-filtered Gaussian field cross-faded by the onsets.
Expressed here through a few small helpers, and everything that is large (the [n_frames,1,h,w] noise fields) is created,
filtered and left on the HIP device — the reference round-trips it through host memory (``.cpu()`` at :45).
"""
import torch

import maua_stylegan2_amd.audioreactive as ar

# band name -> onset-analysis settings (shared: smooth 5, power 2)
ONSET_BANDS = {
    "lo_onsets": {"fmax": 150, "clip": 97},
    "hi_onsets": {"fmin": 500, "clip": 99},
}
JUMP_TARGETS = {"hi_onsets": -4, "lo_onsets": -7}  # selection index each band pulls the latents toward
NOISE_MAX_WIDTH = 256  # wider noise maps stay on the checkpoint's static buffers (get_noise -> None)
FAST_SIGMA, SLOW_SIGMA = 5, 128


def _crossfade(envelope, toward, base):
    """envelope in [0,1] (broadcast over trailing dims): 1 -> ``toward``, 0 -> ``base``."""
    return envelope * toward + (1 - envelope) * base


def initialize(args):
    for name, band in ONSET_BANDS.items():
        setattr(args, name, ar.onsets(args.audio, args.sr, args.n_frames, smooth=5, power=2, **band))
    return args


def get_latents(selection, args):
    weights = ar.chroma(args.audio, args.sr, args.n_frames)
    latents = ar.gaussian_filter(ar.chroma_weight_latents(weights, selection), 4)
    for band in ("hi_onsets", "lo_onsets"):  # treble first, bass on top of it
        envelope = getattr(args, band)[:, None, None]
        latents = _crossfade(envelope, selection[[JUMP_TARGETS[band]]], latents)
    return ar.gaussian_filter(latents, 2, causal=0.2)


def _filtered_field(n_frames, height, width, sigma):
    return ar.gaussian_filter(torch.randn((n_frames, 1, height, width), device="cuda"), sigma)


def get_noise(height, width, scale, num_scales, args):
    if width > NOISE_MAX_WIDTH:
        return None
    bass = args.lo_onsets[:, None, None, None].cuda()
    treble = args.hi_onsets[:, None, None, None].cuda()
    jittery = _filtered_field(args.n_frames, height, width, FAST_SIGMA)
    field = _filtered_field(args.n_frames, height, width, SLOW_SIGMA)
    if width < 128:  # coarse scales follow the bass
        field = _crossfade(bass, jittery, field)
    if width > 32:  # fine scales follow the treble
        field = _crossfade(treble, jittery, field)
    return field / (field.std() * 2.5)
