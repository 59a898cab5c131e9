"""Bit-stable synthetic inputs for the StyleGAN2 inference path.

Everything here is drawn from ``numpy.random.default_rng`` streams keyed by (seed, crc32(name)) so
that the container that imports the reference (to write ``tests/golden``) and the MI355X box (which
never sees the reference) regenerate *identical* weights, latents and noise maps without shipping
them.  torch RNG is never used: its streams differ between CPU and GPU builds.

State-dict key names and shapes follow the reference checkpoint layout
(/root/reference/models/stylegan2.py:388-454, SURVEY.md §8b "Checkpoint").
"""
import math
import zlib
from collections import OrderedDict

import numpy as np
import torch

CHANNELS_BASE = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256, 128: 128, 256: 64, 512: 32, 1024: 16}


def channels_for(channel_multiplier=2):
    """Feature-map width per resolution (reference models/stylegan2.py:395-405)."""
    ch = {}
    for res, c in CHANNELS_BASE.items():
        ch[res] = c if res <= 32 else c * channel_multiplier
    return ch


def fir_kernel_2d(taps=(1, 3, 3, 1), gain=1.0):
    """Normalised outer-product FIR (reference make_kernel, models/stylegan2.py:23-31)."""
    k = np.asarray(taps, dtype=np.float32)
    k2 = np.outer(k, k).astype(np.float32)
    k2 /= k2.sum()
    return (k2 * np.float32(gain)).astype(np.float32)


def generator_tensor_shapes(size, style_dim=512, n_mlp=8, channel_multiplier=2, constant_input=True):
    """Ordered {state_dict key: shape} of the reference ``Generator`` (g_ema) for ``size``."""
    ch = channels_for(channel_multiplier)
    log_size = int(math.log2(size))
    num_layers = (log_size - 2) * 2 + 1
    shapes = OrderedDict()
    for i in range(1, n_mlp + 1):
        shapes[f"style.{i}.weight"] = (style_dim, style_dim)
        shapes[f"style.{i}.bias"] = (style_dim,)
    if constant_input:
        shapes["input.input"] = (1, ch[4], 4, 4)
    else:  # LatentInput (reference models/stylegan2.py:281-294), the --noconst checkpoints
        shapes["input.input"] = (1,)
        shapes["input.linear.weight"] = (ch[4] * 16, style_dim)
        shapes["input.linear.bias"] = (ch[4] * 16,)
        shapes["input.activate.bias"] = (ch[4] * 16,)

    def styled(prefix, cin, cout, up):
        shapes[f"{prefix}.conv.weight"] = (1, cout, cin, 3, 3)
        if up:
            shapes[f"{prefix}.conv.blur.kernel"] = (4, 4)
        shapes[f"{prefix}.conv.modulation.weight"] = (cin, style_dim)
        shapes[f"{prefix}.conv.modulation.bias"] = (cin,)
        shapes[f"{prefix}.noise.weight"] = (1,)
        shapes[f"{prefix}.activate.bias"] = (cout,)

    def torgb(prefix, cin, up):
        shapes[f"{prefix}.bias"] = (1, 3, 1, 1)
        if up:
            shapes[f"{prefix}.upsample.kernel"] = (4, 4)
        shapes[f"{prefix}.conv.weight"] = (1, 3, cin, 1, 1)
        shapes[f"{prefix}.conv.modulation.weight"] = (cin, style_dim)
        shapes[f"{prefix}.conv.modulation.bias"] = (cin,)

    styled("conv1", ch[4], ch[4], False)
    torgb("to_rgb1", ch[4], False)
    cin = ch[4]
    for n, i in enumerate(range(3, log_size + 1)):
        cout = ch[2 ** i]
        styled(f"convs.{2 * n}", cin, cout, True)
        styled(f"convs.{2 * n + 1}", cout, cout, False)
        cin = cout
    cin = ch[4]
    for n, i in enumerate(range(3, log_size + 1)):
        torgb(f"to_rgbs.{n}", ch[2 ** i], True)
    for layer_idx in range(num_layers):
        res = 2 ** ((layer_idx + 5) // 2)
        shapes[f"noises.noise_{layer_idx}"] = (1, 1, res, res)
    return shapes


def _rng(seed, name):
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def seeded_array(seed, name, shape, std=1.0, mean=0.0):
    a = _rng(seed, name).standard_normal(size=tuple(shape), dtype=np.float32)
    if std != 1.0:
        a *= np.float32(std)
    if mean != 0.0:
        a += np.float32(mean)
    return a


def seeded_state_dict(size, seed=0, style_dim=512, n_mlp=8, channel_multiplier=2, constant_input=True, rgb_gain=1.0):
    """Random-init-like checkpoint with *non-trivial* noise strengths and biases.

    The reference's random init has noise.weight = activate.bias = ToRGB.bias = 0
    (models/stylegan2.py:260,354; op/fused_act.py:78), which would leave those paths untested
    (SURVEY.md §7 "Hard parts"); here they are N(0, 0.1) around their init value.

    ``rgb_gain``: factor on every ToRGB weight and bias.  The image is linear in them (ToRGB is not demodulated,
    models/stylegan2.py:352,356-365), so the generator's output is exactly ``rgb_gain`` times the plain checkpoint's.  With N(0,1)
    ToRGB weights a 1024^2 image has a standard deviation of ~3: three quarters of its pixels clamp to 0 / 255 in the uint8
    frame (render.py:40-43), where a comparison of frames sees nothing.  The whole-frame parity tests use ``UNSATURATED_RGB_GAIN``.
    """
    sd = OrderedDict()
    for key, shape in generator_tensor_shapes(size, style_dim, n_mlp, channel_multiplier, constant_input).items():
        if key.endswith("blur.kernel") or key.endswith("upsample.kernel"):
            arr = fir_kernel_2d((1, 3, 3, 1), gain=4.0)
        elif key.endswith("modulation.bias"):
            arr = seeded_array(seed, key, shape, std=0.1, mean=1.0)
        elif key.endswith("noise.weight"):
            arr = seeded_array(seed, key, shape, std=0.1)
        elif key.endswith("activate.bias") or (key.startswith("to_rgb") and key.endswith(".bias") and len(shape) == 4):
            arr = seeded_array(seed, key, shape, std=0.1)
        elif (key.startswith("style.") and key.endswith(".bias")) or key in ("input.linear.bias", "input.activate.bias"):
            arr = seeded_array(seed, key, shape, std=0.1)
        else:
            arr = seeded_array(seed, key, shape)
        if rgb_gain != 1.0 and key.startswith("to_rgb") and (key.endswith(".conv.weight") or (key.endswith(".bias") and len(shape) == 4)):
            arr = arr * np.float32(rgb_gain)
        sd[key] = torch.from_numpy(np.ascontiguousarray(arr))
    return sd


def unsaturated_rgb_gain(size):
    """ToRGB gain of the whole-frame parity tests: the seeded generator's image then has a standard deviation of ~0.4 (measured std
    at gain 1: 0.93 / 1.69 / 3.0 at 64 / 256 / 1024 px), i.e. the uint8 frame uses the whole grey range and < 5 % of it clamps."""
    return min(1.0, 0.4 / (0.93 * (size / 64.0) ** 0.42))


def clamped_fraction(image):
    """Share of the values of a float image (or of a uint8 frame) that sit on the clamp of render.py:40-43."""
    a = image.detach().cpu().numpy() if hasattr(image, "detach") else np.asarray(image)
    if a.dtype == np.uint8:
        return float(((a == 0) | (a == 255)).mean())
    return float((np.abs(a) >= 1.0).mean())


def seeded_latents(n, n_latent, seed=1, style_dim=512):
    return torch.from_numpy(seeded_array(seed, "latents", (n, n_latent, style_dim)))


def noise_sizes(size):
    """Spatial side of each of the num_layers noise maps: 4, 8, 8, 16, 16, ... size, size."""
    log_size = int(math.log2(size))
    return [2 ** ((i + 5) // 2) for i in range((log_size - 2) * 2 + 1)]


def seeded_noise(n, size, seed=2):
    return [torch.from_numpy(seeded_array(seed, f"noise_{i}", (n, 1, r, r))) for i, r in enumerate(noise_sizes(size))]


def synthetic_audio(duration_s, sr=22050, seed=3, bpm=120.0):
    """Seeded test track: 120-BPM decaying kick + 3-note sine chord progression + -30 dB white noise
    (SURVEY.md §8d config 2). Returns float32 mono in [-1, 1]."""
    n = int(round(duration_s * sr))
    t = np.arange(n, dtype=np.float64) / sr
    beat = 60.0 / bpm
    phase = np.mod(t, beat)
    kick = np.sin(2 * np.pi * (55.0 + 60.0 * np.exp(-phase * 30.0)) * phase) * np.exp(-phase * 12.0)
    chords = [(220.0, 277.18, 329.63), (196.0, 246.94, 293.66), (174.61, 220.0, 261.63), (146.83, 185.0, 220.0)]
    bar = np.floor(t / (4 * beat)).astype(np.int64) % len(chords)
    harm = np.zeros(n)
    for ci, notes in enumerate(chords):
        m = bar == ci
        for f in notes:
            harm[m] += np.sin(2 * np.pi * f * t[m])
    harm /= 3.0
    hiss = _rng(seed, "audio_noise").standard_normal(n) * 10 ** (-30 / 20)
    y = 0.6 * kick + 0.3 * harm + hiss
    y /= np.max(np.abs(y)) + 1e-9
    return y.astype(np.float32)
