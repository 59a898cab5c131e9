"""Oracle (test infrastructure, CPU) for the audio-feature / latent / noise stages.

PINNED against the imported reference (tests/golden/audioreactive_torch.npz):
  gaussian_filter       <- /root/reference/audioreactive/signal.py:319-368
  percentile / percentile_clip / normalize / compress <- signal.py:243-316
  chroma_weight_latents <- /root/reference/audioreactive/latent.py:15-26
  noise_side_lengths    <- /root/reference/generate_audiovisual.py:22-34,147-151
  wrapping_slice        <- latent.py:110-133
  perlin_noise          <- latent.py:188-246 (tests/golden/perlin.npz: the reference itself, run on the CPU by
                           tests/golden/make_golden.py with Tensor.cuda replaced by the identity and numpy's global
                           generator seeded; the oracle takes the gradient angles as inputs and agrees to 0.0)

PARITY UNPINNED (stated in DESIGN.md): stft_power / mel_filterbank / onset_strength / the band-filtered onset
functions / chroma_filterbank / chroma_stft / constant-Q / CENS / nn_filter / hpss restate the *published* librosa and
madmom algorithms that the reference calls at signal.py:49-67,115-131,150 (un-pinned, un-vendored dependencies,
requirements.txt:4,6, not installed in this image, so no golden vector can be produced), and affine_reflect_warp
restates the kornia composition of bend.py:60-102.  They define what the HIP kernels must reproduce; they are NOT claimed
to be bit-identical to any librosa / madmom / kornia release.  Their building blocks are cross-checked against the
third-party code that is installed (scipy.signal.stft, scipy.signal.resample, scipy.ndimage.median_filter,
scikit-learn NearestNeighbors, torch grid_sample) in tests/test_oracle_golden.py.
"""

import numpy as np
import scipy.signal
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------------- pinned stages


def gaussian_filter(x, sigma, causal=None, smf=1.0):
    """Circular Gaussian FIR along dim 0 (signal.py:319-368). ``smf`` is the module-global SMF (:21-23,335)."""
    nd = x.dim()
    n = x.shape[0]
    flat = x.reshape(n, -1).to(torch.float32)  # [T, F]
    radius = min(int(sigma * 4 * smf), 3 * n)
    taps = torch.arange(-radius, radius + 1, dtype=torch.float32)
    taps = torch.exp(-0.5 / sigma ** 2 * taps ** 2)
    if causal is not None:
        taps[radius + 1:] *= causal if isinstance(causal, float) else 0
    taps = taps / taps.sum()
    seq = flat.t()[:, None, :]  # [F,1,T]
    if radius > n:  # :350-355 — one circular wrap of n on both sides, then zeros
        seq = F.pad(seq, (n, n), mode="circular")
        seq = F.pad(seq, (radius - n, radius - n))
    else:
        seq = F.pad(seq, (radius, radius), mode="circular")
    y = F.conv1d(seq, taps.reshape(1, 1, -1))[:, 0, :].t()
    if nd < 3:  # the reference lifts to 3-D and .squeeze()s back (:331-332,365-366): size-1 dims vanish
        return y.reshape(list(x.shape) + [1] * (3 - nd)).squeeze()
    return y.reshape(x.shape)


def percentile(sig, p):
    k = 1 + round(0.01 * float(p) * (sig.numel() - 1))
    return sig.reshape(-1).kthvalue(k).values.item()


def percentile_clip(sig, p):
    """signal.py:271-292: percentile taken over strict local maxima only."""
    n = sig.shape[0]
    idx = torch.arange(n)
    nxt = sig[(idx + 1).clamp(0, n - 1)]
    prv = sig[(idx - 1).clamp(0, n - 1)]
    peaks = (sig > nxt) & (sig > prv)
    out = sig.clamp(0, percentile(sig[peaks], p))
    return out / out.max()


def normalize(sig):
    sig = sig - sig.min()
    return sig / sig.max()


def compress(sig, threshold, ratio, invert=False):
    sig = sig.clone()
    mask = sig < threshold if invert else sig > threshold
    sig[mask] = sig[mask] * ratio
    return normalize(sig)


def chroma_weight_latents(chroma, latents):
    return torch.einsum("tn,nld->tld", chroma, latents)


def noise_side_lengths(out_size, g_res):
    """Side length 2**int(scale/2) for each StyleGAN2 noise scale (generate_audiovisual.py:22-34,147-151)."""
    log_max = int(np.log2(out_size))
    log_min = 2 + (log_max - int(np.log2(g_res)))
    return [2 ** int(s / 2) for s in range(2 * log_min + 1, 2 * (log_max + 1))]


def resample(x, num):
    """Fourier resampling along axis 0 — scipy.signal.resample is the third-party routine the reference
    calls (signal.py:68,152) and IS installed here, so it is used directly as the anchor."""
    return scipy.signal.resample(np.asarray(x, dtype=np.float64), num, axis=0)


# --------------------------------------------------------------------------------------------- unpinned stages


def hann_periodic(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_power(y, n_fft=2048, hop=512, power=2.0):
    """|STFT|**power, centred frames with reflect padding, periodic Hann (librosa.stft defaults).
    Returns [1 + n_fft/2, n_frames] float64 with n_frames = 1 + len(y)//hop."""
    y = np.asarray(y, dtype=np.float64)
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(ypad) - n_fft) // hop
    win = hann_periodic(n_fft)
    frames = np.stack([ypad[t * hop: t * hop + n_fft] * win for t in range(n_frames)], axis=1)
    spec = np.fft.rfft(frames, axis=0)
    mag2 = spec.real ** 2 + spec.imag ** 2
    return mag2 if power == 2.0 else mag2 ** (power / 2.0)


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mel = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mel)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft=2048, n_mels=128, fmin=0.0, fmax=None):
    """Slaney-scale triangular filters with area normalisation (librosa.filters.mel defaults)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fft_f = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2))
    fb = np.zeros((n_mels, fft_f.size))
    for m in range(n_mels):
        lo, ce, hi = mel_f[m], mel_f[m + 1], mel_f[m + 2]
        up = (fft_f - lo) / (ce - lo)
        dn = (hi - fft_f) / (hi - ce)
        fb[m] = np.maximum(0.0, np.minimum(up, dn)) * (2.0 / (hi - lo))
    return fb


def power_to_db(s, amin=1e-10, top_db=80.0):
    db = 10.0 * np.log10(np.maximum(amin, s))
    return np.maximum(db, db.max() - top_db)


def onset_strength(y, sr, fmin=0.0, fmax=None, n_fft=2048, hop=512, n_mels=128):
    """Spectral-flux onset envelope (librosa.onset.onset_strength defaults: lag 1, max_size 1, mean
    aggregate, centre compensation of lag + n_fft//(2*hop) frames).  Returns [n_frames]."""
    fmax = 11025.0 if fmax is None else fmax
    mel = mel_filterbank(sr, n_fft, n_mels, fmin, fmax) @ stft_power(y, n_fft, hop)
    db = power_to_db(mel)
    flux = np.maximum(0.0, db[:, 1:] - db[:, :-1]).mean(axis=0)
    pad = 1 + n_fft // (2 * hop)
    env = np.concatenate([np.zeros(pad), flux])
    return env[: db.shape[1]]


def chroma_filterbank(sr, n_fft=2048, n_chroma=12, ctroct=5.0, octwidth=2.0, tuning=0.0):
    """librosa.filters.chroma (L2-normalised columns, Gaussian octave weighting, C-based); ``tuning`` = deviation of A440 in
    fractions of a chroma bin."""
    freqs = np.linspace(0, sr, n_fft, endpoint=False)[1:]
    bins = n_chroma * np.log2(freqs / (440.0 * 2.0 ** (tuning / n_chroma) / 16))
    bins = np.concatenate([[bins[0] - 1.5 * n_chroma], bins])
    width = np.concatenate([np.maximum(bins[1:] - bins[:-1], 1.0), [1.0]])
    d = bins[None, :] - np.arange(n_chroma, dtype=np.float64)[:, None]
    half = np.round(n_chroma / 2.0)
    d = np.remainder(d + half + 10 * n_chroma, n_chroma) - half
    w = np.exp(-0.5 * (2 * d / width[None, :]) ** 2)
    w /= np.maximum(np.sqrt((w ** 2).sum(axis=0, keepdims=True)), np.finfo(np.float64).tiny)
    w *= np.exp(-0.5 * ((bins / n_chroma - ctroct) / octwidth) ** 2)[None, :]
    w = np.roll(w, -3 * (n_chroma // 12), axis=0)
    return w[:, : 1 + n_fft // 2]


def localmax_rows(x):
    """librosa.util.localmax along axis 0: strictly greater than the element before, not smaller than the one after
    (edge-padded)."""
    xp = np.pad(x, ((1, 1), (0, 0)), mode="edge")
    return (x > xp[:-2]) & (x >= xp[2:])


def piptrack(S, sr, n_fft=2048, fmin=150.0, fmax=4000.0, threshold=0.1):
    """librosa.piptrack (published algorithm, librosa/core/pitch.py): parabolic interpolation of the spectrogram ``S``
    [1 + n_fft/2, T] around every local maximum above ``threshold`` x the frame maximum inside [fmin, fmax) ->
    (pitches [Hz], magnitudes), zero elsewhere.  **parity unpinned** (librosa absent)."""
    S = np.abs(np.asarray(S, dtype=np.float64))
    fmax = min(fmax, sr / 2.0)
    freqs = np.arange(S.shape[0]) * (float(sr) / n_fft)
    avg = 0.5 * (S[2:] - S[:-2])
    shift = 2 * S[1:-1] - S[2:] - S[:-2]
    shift = avg / (shift + (np.abs(shift) < np.finfo(np.float64).tiny))
    avg = np.pad(avg, ((1, 1), (0, 0)))
    shift = np.pad(shift, ((1, 1), (0, 0)))
    dskew = 0.5 * avg * shift
    ref = threshold * S.max(axis=0, keepdims=True)
    peak = ((max(fmin, 0.0) <= freqs) & (freqs < fmax))[:, None] & localmax_rows(S * (S > ref))
    bins = np.arange(S.shape[0], dtype=np.float64)[:, None]
    return np.where(peak, (bins + shift) * float(sr) / n_fft, 0.0), np.where(peak, S + dskew, 0.0)


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    """librosa.pitch_tuning: histogram peak of the frequencies' deviation from the bins of an A440 scale, in fractions of a bin
    in [-0.5, 0.5)."""
    f = np.asarray(frequencies, dtype=np.float64).ravel()
    f = f[f > 0]
    if f.size == 0:
        return 0.0
    residual = np.mod(bins_per_octave * np.log2(f / (440.0 / 16)), 1.0)
    residual[residual >= 0.5] -= 1.0
    edges = np.linspace(-0.5, 0.5, int(np.ceil(1.0 / resolution)) + 1)
    counts, edges = np.histogram(residual, edges)
    return float(edges[np.argmax(counts)])


def estimate_tuning(y=None, sr=22050, S=None, n_fft=2048, resolution=0.01, bins_per_octave=12):
    """librosa.estimate_tuning: pitches from piptrack (on |STFT| of ``y``, hop n_fft/4, or on a given spectrogram ``S``), those at
    least as strong as the median peak -> pitch_tuning."""
    if S is None:
        S = np.sqrt(stft_power(y, n_fft, n_fft // 4))
    pitch, mag = piptrack(S, sr, n_fft)
    mask = pitch > 0
    thr = np.median(mag[mask]) if mask.any() else 0.0
    return pitch_tuning(pitch[(mag >= thr) & mask], resolution, bins_per_octave)


def chroma_stft(y, sr, n_fft=2048, hop=512, tuning=None):
    """Power-spectrogram chroma, each frame normalised by its max (librosa.feature.chroma_stft; ``tuning`` None = estimated
    from the power spectrogram, as librosa does)."""
    P = stft_power(y, n_fft, hop)
    if tuning is None:
        tuning = estimate_tuning(S=P, sr=sr, n_fft=n_fft, bins_per_octave=12)
    raw = chroma_filterbank(sr, n_fft, tuning=tuning) @ P
    peak = raw.max(axis=0, keepdims=True)
    return raw / np.where(peak > np.finfo(np.float64).tiny, peak, 1.0)


def stft_complex(y, n_fft=2048, hop=512):
    y = np.asarray(y, dtype=np.float64)
    ypad = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(ypad) - n_fft) // hop
    win = hann_periodic(n_fft)
    frames = np.stack([ypad[t * hop: t * hop + n_fft] * win for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames, axis=0)


def istft(spec, length, n_fft=2048, hop=512):
    """Window-sum-square normalised overlap-add, centre padding removed (librosa.istft)."""
    n_frames = spec.shape[1]
    win = hann_periodic(n_fft)
    frames = np.fft.irfft(spec, n=n_fft, axis=0) * win[:, None]
    total = n_fft + hop * (n_frames - 1)
    y = np.zeros(total)
    wss = np.zeros(total)
    for t in range(n_frames):
        y[t * hop: t * hop + n_fft] += frames[:, t]
        wss[t * hop: t * hop + n_fft] += win ** 2
    ok = wss > np.finfo(np.float32).tiny
    y[ok] /= wss[ok]
    return y[n_fft // 2: n_fft // 2 + length]


def softmask(x, x_ref, power=2.0, split_zeros=False):
    z = np.maximum(x, x_ref)
    bad = z < np.finfo(np.float32).tiny
    z = np.where(bad, 1.0, z)
    m, r = (x / z) ** power, (x_ref / z) ** power
    out = np.where(bad, 0.5 if split_zeros else 0.0, m / np.where(bad, 1.0, m + r))
    return out


def hpss(y, margin=1.0, kernel_size=31, power=2.0):
    """librosa.decompose.hpss + effects.harmonic/percussive: median filters (scipy.ndimage, mode 'reflect') along time
    and frequency of |STFT|, soft masks with margin, inverse STFT.  Returns (y_harmonic, y_percussive)."""
    import scipy.ndimage

    d = stft_complex(y)
    s = np.abs(d)
    harm = scipy.ndimage.median_filter(s, size=(1, kernel_size), mode="reflect")
    perc = scipy.ndimage.median_filter(s, size=(kernel_size, 1), mode="reflect")
    split = margin == 1
    mask_h = softmask(harm, perc * margin, power, split)
    mask_p = softmask(perc, harm * margin, power, split)
    return istft(d * mask_h, len(y)), istft(d * mask_p, len(y))


def log_filterbank(sr, n_fft=2048, num_bands=24, fmin=20.0, fmax=8000.0, fref=440.0):
    """Logarithmically spaced triangular filterbank in the style of madmom.audio.filters.LogarithmicFilterbank (the
    FilteredSpectrogram of signal.py:57): centre frequencies fref * 2**(k/num_bands) inside [fmin, fmax] snapped to the
    nearest FFT bin, duplicates removed; filter i rises from centre i-1 to centre i and falls to centre i+1; every filter
    sums to one.  Returns [n_filters, n_fft/2] (the Nyquist bin is not used, as in madmom).  **parity unpinned.**"""
    n_bins = n_fft // 2
    bin_freqs = np.arange(n_bins) * (sr / n_fft)
    lo = int(np.floor(np.log2(fmin / fref) * num_bands))
    hi = int(np.ceil(np.log2(fmax / fref) * num_bands))
    freqs = fref * 2.0 ** (np.arange(lo, hi + 1) / num_bands)
    freqs = freqs[(freqs >= fmin) & (freqs <= fmax)]
    centres = np.unique(np.clip(np.round(freqs / (sr / n_fft)).astype(np.int64), 0, n_bins - 1))
    if len(centres) < 3:
        raise ValueError("log_filterbank: fewer than three distinct FFT bins between fmin and fmax")
    fb = np.zeros((len(centres) - 2, n_bins))
    for i, (start, centre, stop) in enumerate(zip(centres[:-2], centres[1:-1], centres[2:])):
        fb[i, start:centre] = np.linspace(0.0, 1.0, centre - start, endpoint=False)
        fb[i, centre:stop] = np.linspace(1.0, 0.0, stop - centre, endpoint=False)
        fb[i] /= fb[i].sum()
    del bin_freqs
    return fb


def madmom_like_onset_functions(filt):
    """The four phase-free onset detection functions signal.py:58-66 sums (madmom.features.onsets, frame lag 1), on a
    filtered magnitude spectrogram ``filt`` [n_frames, n_bands]: spectral_diff = sum of squared positive differences,
    spectral_flux = sum of positive differences, superflux = positive differences against the 3-bin frequency-maximum of the
    previous frame, modified_kullback_leibler = mean log(1 + S[t] / (S[t-1] + eps)).  (The fifth, complex_flux, needs the phase
    spectrogram: :func:`complex_flux`.)  Returns a dict of [n_frames] arrays."""
    filt = np.asarray(filt, dtype=np.float64)
    prev = np.concatenate([filt[:1], filt[:-1]], axis=0)
    pos = np.maximum(filt - prev, 0.0)
    padded = np.pad(prev, ((0, 0), (1, 1)), mode="edge")
    prev_max = np.maximum(np.maximum(padded[:, :-2], padded[:, 1:-1]), padded[:, 2:])
    sup = np.maximum(filt - prev_max, 0.0)
    sup[0] = 0.0
    mkl = np.log1p(filt / (prev + np.finfo(np.float64).eps)).mean(axis=1)
    mkl[0] = 0.0
    return {"spectral_diff": (pos ** 2).sum(axis=1), "spectral_flux": pos.sum(axis=1), "superflux": sup.sum(axis=1),
            "modified_kullback_leibler": mkl}


def local_group_delay(spec):
    """|local group delay| / pi of a complex STFT ``spec`` [bins, frames] whose frames were circularly shifted by half a
    window (madmom ShortTimeFourierTransform(circular_shift=True), signal.py:55: the phase is then measured against the frame
    centre): bin k of the un-shifted transform is multiplied by (-1)^k, the phase unwrapped along frequency, differenced
    (phase[k] - phase[k+1], last bin 0 — madmom.audio.stft.Phase.local_group_delay), absolute value, / pi."""
    spec = np.asarray(spec)
    sign = np.where(np.arange(spec.shape[0]) % 2 == 0, 1.0, -1.0)[:, None]
    phase = np.unwrap(np.angle(spec * sign), axis=0)
    lgd = np.zeros_like(phase)
    lgd[:-1] = phase[:-1] - phase[1:]
    return np.abs(lgd) / np.pi


def complex_flux(spec, fb, filt, temporal_filter=3):
    """madmom.features.onsets.complex_flux (Boeck & Widmer 2013, "Local group delay based vibrato and tremolo suppression for
    onset detection"), the fifth member of the sum at signal.py:58-67: the SuperFlux difference (positive difference against
    the 3-bin frequency maximum of the previous frame) of the filtered spectrogram ``filt`` [frames, bands], weighted per band
    by the MINIMUM local group delay over the FFT bins the band's filter covers (widened by one bin on each side), after a
    3-frame temporal maximum filter of the local group delay.  ``spec`` [bins, frames] complex STFT, ``fb`` [bands, bins]."""
    import scipy.ndimage

    lgd = local_group_delay(spec).T  # [frames, bins]
    if temporal_filter > 0:
        lgd = scipy.ndimage.maximum_filter(lgd, size=[temporal_filter, 1])
    filt = np.asarray(filt, dtype=np.float64)
    mask = np.zeros_like(filt)
    n_bins = lgd.shape[1]
    for b in range(mask.shape[1]):
        corner = np.nonzero(fb[b])[0]
        start, stop = max(corner[0] - 1, 0), min(corner[-1] + 2, n_bins)
        mask[:, b] = lgd[:, start:stop].min(axis=1)
    prev = np.concatenate([filt[:1], filt[:-1]], axis=0)
    padded = np.pad(prev, ((0, 0), (1, 1)), mode="edge")
    prev_max = np.maximum(np.maximum(padded[:, :-2], padded[:, 1:-1]), padded[:, 2:])
    diff = np.maximum(filt - prev_max, 0.0)
    diff[0] = 0.0
    return (diff * mask).sum(axis=1)


def madmom_like_onset_strength(y, sr, fmin=20.0, fmax=8000.0, n_fft=2048, hop=441):
    """Sum of :func:`madmom_like_onset_functions` and :func:`complex_flux` — all five members of signal.py:58-67 — on the
    log-filtered magnitude spectrogram (frame 2048, hop 441 = 50 frames per second at 22050 Hz, signal.py:54-57).  Frames are the
    centred, reflect-padded, periodic-Hann frames of :func:`stft_complex` (madmom zero-pads and uses a symmetric window: only the
    edge frames differ)."""
    spec = stft_complex(y, n_fft, hop)[: n_fft // 2]  # [bins, frames]
    fb = log_filterbank(sr, n_fft, 24, fmin, fmax)
    filt = (fb @ np.abs(spec)).T
    return sum(madmom_like_onset_functions(filt).values()) + complex_flux(spec, fb, filt)


def onsets(y, sr, n_frames, margin=8, fmin=20, fmax=8000, smooth=1, clip=100, power=1, smf=1.0, type="rosa"):
    """signal.py:31-73: percussive separation (:49), then the spectral-flux onset strength (type="rosa", :51) or the sum of
    madmom-style onset functions on a log-filtered spectrogram (type="mm", :53-67), then resample / smooth / clip / power."""
    if margin:
        y = hpss(y, margin)[1]
    env = onset_strength(y, sr, fmin=fmin, fmax=fmax) if type == "rosa" else madmom_like_onset_strength(y, sr, fmin, fmax)
    env = np.clip(resample(env, n_frames), env.min(), env.max())
    env = torch.from_numpy(env).float()
    env = gaussian_filter(env, smooth, causal=0, smf=smf)
    env = percentile_clip(env, clip)
    return env ** power


def cqt_frequencies(n_bins=252, fmin=32.70319566257483, bins_per_octave=36):
    return fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)


def cqt_lengths(sr, freqs, bins_per_octave=36, filter_scale=1.0):
    """Filter lengths of a constant-Q analysis: N_k = ceil(Q sr / f_k), Q = filter_scale / (2**(1/bpo) - 1)."""
    q = filter_scale / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    return np.ceil(q * sr / freqs).astype(np.int64)


def cqt_magnitude(y, sr, hop=512, n_bins=252, fmin=32.70319566257483, bins_per_octave=36):
    """|constant-Q transform| by its definition (Brown 1991): for every frame t (centred, reflect padded, hop 512) and
    bin k, correlate N_k samples with a periodic-Hann-windowed complex exponential at f_k, window normalised to unit L1
    and scaled by 1/sqrt(N_k) (librosa.cqt: norm=1, scale=True).  librosa evaluates the same transform recursively with
    down-sampling and sparsified FFT-domain kernels; this direct form is its definition.  **parity unpinned.**
    Returns [n_bins, 1 + len(y)//hop] float64."""
    y = np.asarray(y, dtype=np.float64)
    freqs = cqt_frequencies(n_bins, fmin, bins_per_octave)
    lengths = cqt_lengths(sr, freqs, bins_per_octave)
    n_frames = 1 + len(y) // hop
    pad = int(lengths.max() // 2 + 1)
    ypad = np.pad(y, pad, mode="reflect")  # (numpy reflects repeatedly when pad exceeds the signal length)
    out = np.zeros((n_bins, n_frames))
    for k in range(n_bins):
        n = int(lengths[k])
        m = np.arange(n)
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * m / n)
        kern = win / win.sum() * np.exp(-2j * np.pi * freqs[k] * (m - n // 2) / sr)
        start = pad - n // 2
        idx = start + hop * np.arange(n_frames)[:, None] + m[None, :]
        out[k] = np.abs(ypad[idx] @ kern) / np.sqrt(n)
    return out


def cq_to_chroma_matrix(n_bins=252, bins_per_octave=36, n_chroma=12):
    """Fold constant-Q bins to pitch classes: every semitone owns the bins_per_octave/n_chroma bins centred on it (the
    bin below C1's centre does not exist and is dropped); fmin = C1, so chroma row 0 is C."""
    merge = bins_per_octave // n_chroma
    w = np.zeros((n_chroma, n_bins))
    for b in range(n_bins):
        semitone = int(np.floor((b + merge // 2) / merge))
        w[semitone % n_chroma, b] = 1.0
    return w


def chroma_cqt(y, sr, hop=512, tuning=None):
    """Constant-Q chromagram, each frame normalised by its max (librosa.feature.chroma_cqt defaults: 7 octaves x 36 bins from
    C1 x 2^(tuning/36), ``tuning`` None = librosa.estimate_tuning(y, bins_per_octave=36))."""
    if tuning is None:
        tuning = estimate_tuning(y, sr, bins_per_octave=36)
    raw = cq_to_chroma_matrix() @ cqt_magnitude(y, sr, hop, fmin=32.70319566257483 * 2.0 ** (tuning / 36))
    peak = raw.max(axis=0, keepdims=True)
    return raw / np.where(peak > np.finfo(np.float64).tiny, peak, 1.0)


def cens_from_chroma(ch, win_len=41):
    """CENS post-processing (Mueller & Ewert 2011, the steps librosa.feature.chroma_cens applies after its chromagram):
    per-frame L1 normalisation -> quantisation with thresholds .05/.1/.2/.4 (0.25 each) -> smoothing along time with a
    sum-normalised Hann window of ``win_len`` taps (zero-padded 'same' convolution) -> per-frame L2 normalisation.
    ``ch`` [12, T] non-negative.  **parity unpinned** (librosa absent); the chromagram fed in is the STFT one, not a CQT."""
    ch = np.asarray(ch, dtype=np.float64)
    l1 = np.abs(ch).sum(axis=0, keepdims=True)
    c = ch / np.where(l1 > np.finfo(np.float32).tiny, l1, 1.0)
    q = np.zeros_like(c)
    for thr in (0.4, 0.2, 0.1, 0.05):
        q += 0.25 * (c > thr)
    n = win_len + 2
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / (n - 1))  # scipy get_window("hann", n, fftbins=False)
    win = win[1:-1]
    win = win / win.sum()
    half = win_len // 2
    padded = np.pad(q, ((0, 0), (half, half)))
    sm = np.stack([(padded[:, t: t + win_len] * win[::-1]).sum(axis=1) for t in range(ch.shape[1])], axis=1)
    l2 = np.sqrt((sm ** 2).sum(axis=0, keepdims=True))
    return sm / np.where(l2 > np.finfo(np.float32).tiny, l2, 1.0)


def nn_filter_median(ch, width=1):
    """Nearest-neighbour median filter of a [F, T] feature sequence (the role of librosa.decompose.nn_filter(S,
    aggregate=np.median, metric="cosine") in signal.py:131): for every frame, the k = 2*ceil(sqrt(T - 2*width + 1)) frames
    of highest cosine similarity (frames closer than ``width`` excluded; ties broken toward the lower index) are
    aggregated by the per-feature median.  **parity unpinned.**"""
    ch = np.asarray(ch, dtype=np.float64)
    f, t = ch.shape
    k = int(min(t - 1, 2 * np.ceil(np.sqrt(max(t - 2 * width + 1, 1)))))
    norms = np.sqrt((ch ** 2).sum(axis=0))
    inv = 1.0 / np.where(norms > np.finfo(np.float32).tiny, norms, 1.0)
    unit = ch * inv
    sim = unit.T @ unit  # [T, T]
    out = np.empty_like(ch)
    idx = np.arange(t)
    for i in range(t):
        s = sim[i].copy()
        s[np.abs(idx - i) < width] = -np.inf
        nb = np.argsort(-s, kind="stable")[:k]
        out[:, i] = np.median(ch[:, nb], axis=1)
    return out


def chroma(y, sr, n_frames, margin=16, notes=12, type="stft", nearest_neighbor=False, return_order=False):
    """signal.py:136-156: harmonic separation (:150) -> raw_chroma (:102-133; "stft" chromagram, optionally CENS
    post-processed and nearest-neighbour median filtered) -> resample -> note selection -> per-frame normalisation.
    ``return_order``: also return (pitch class of every output column, the medians the columns were ordered by) — the
    tests compare column for column and allow a swap only between columns whose medians are closer than their tolerance."""
    if margin:
        y = hpss(y, margin)[0]
    raw = chroma_stft(y, sr) if type == "stft" else chroma_cqt(y, sr)
    if type == "cens":
        raw = cens_from_chroma(raw)
    if nearest_neighbor:
        raw = np.minimum(raw, nn_filter_median(raw))
    ch = raw.T
    ch = resample(ch, n_frames)
    medians = np.median(ch, axis=0)
    keep = np.argsort(medians)[:notes]
    ch = ch[:, keep]
    out = torch.from_numpy(ch / ch.sum(1)[:, None]).float()
    return (out, keep, medians[keep]) if return_order else out


def perlin_noise(shape, res, theta, phi, tileable=(True, False, False)):
    """3-D Perlin noise in [-1,1]*... (latent.py:188-246) with the gradient angles supplied by the caller
    (``theta``/``phi`` [res0+1,res1+1,res2+1], the two np.random.rand draws at :209-210 times 2*pi)."""
    d = tuple(shape[i] // res[i] for i in range(3))
    axes = [np.arange(shape[i], dtype=np.float64) * (res[i] / shape[i]) for i in range(3)]
    frac = [a % 1 for a in axes]
    g = np.stack([np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)], axis=3)
    if tileable[0]:
        g[-1] = g[0]
    if tileable[1]:
        g[:, -1] = g[:, 0]
    if tileable[2]:
        g[:, :, -1] = g[:, :, 0]
    cell = [np.arange(shape[i]) // d[i] for i in range(3)]
    fx, fy, fz = np.meshgrid(frac[0], frac[1], frac[2], indexing="ij")
    cx, cy, cz = np.meshgrid(cell[0], cell[1], cell[2], indexing="ij")

    def corner(ox, oy, oz):
        gv = g[cx + ox, cy + oy, cz + oz]
        return (fx - ox) * gv[..., 0] + (fy - oy) * gv[..., 1] + (fz - oz) * gv[..., 2]

    def fade(t):
        return t * t * t * (t * (t * 6 - 15) + 10)

    tx, ty, tz = fade(fx), fade(fy), fade(fz)
    n00 = corner(0, 0, 0) * (1 - tx) + tx * corner(1, 0, 0)
    n10 = corner(0, 1, 0) * (1 - tx) + tx * corner(1, 1, 0)
    n01 = corner(0, 0, 1) * (1 - tx) + tx * corner(1, 0, 1)
    n11 = corner(0, 1, 1) * (1 - tx) + tx * corner(1, 1, 1)
    n0 = (1 - ty) * n00 + ty * n10
    n1 = (1 - ty) * n01 + ty * n11
    return ((1 - tz) * n0 + tz * n1) * 2 - 1


def affine_reflect_warp(x, inv_maps, pads, add_noise=None):
    """CenterCrop(h,w)(Affine(ReflectionPad2d(pads)(x) [+ noise])) evaluated per output pixel — restates the
    composition at /root/reference/audioreactive/bend.py:60-68,84,101 with kornia's documented conventions
    (bilinear, zeros outside the canvas, pixel-unit maps).  x [B,C,h,w] numpy, inv_maps [B,6], pads (l,r,t,b)."""
    x = np.asarray(x, dtype=np.float64)
    b, c, h, w = x.shape
    # ``pads`` may be a list of (l,r,t,b): stacked ReflectionPad2d modules (bend.py:60-64), each padding the canvas built so
    # far — pinned against real stacked torch.nn.ReflectionPad2d in tests/test_oracle_golden.py
    chain = [tuple(pads)] if isinstance(pads[0], (int, np.integer)) else [tuple(p) for p in pads]
    canvas = x
    for cl, cr, ct, cb in chain:
        canvas = np.pad(canvas, ((0, 0), (0, 0), (ct, cb), (cl, cr)), mode="reflect")
    pl, pr, pt, pb = (sum(p[i] for p in chain) for i in range(4))
    if add_noise is not None:
        canvas = canvas + np.asarray(add_noise, dtype=np.float64).reshape(1, 1, h + pt + pb, w + pl + pr)
    ch, cw = canvas.shape[-2:]
    oy, ox = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    cy, cx = oy + (ch - h) // 2, ox + (cw - w) // 2
    out = np.zeros_like(x)
    for i in range(b):
        a = np.asarray(inv_maps[i], dtype=np.float64)
        sx = a[0] * cx + a[1] * cy + a[2]
        sy = a[3] * cx + a[4] * cy + a[5]
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        fx, fy = sx - x0, sy - y0
        for dy in (0, 1):
            for dx in (0, 1):
                yy, xx = y0 + dy, x0 + dx
                ok = (yy >= 0) & (yy < ch) & (xx >= 0) & (xx < cw)
                wgt = np.where(dy, fy, 1 - fy) * np.where(dx, fx, 1 - fx) * ok
                out[i] += wgt[None] * canvas[i][:, np.clip(yy, 0, ch - 1), np.clip(xx, 0, cw - 1)]
    return out


def rms(y, sr, n_frames, fmin=20, fmax=8000, smooth=180, clip=50, power=6, smf=1.0):
    """signal.py:76-99: 12th-order Butterworth band-pass (scipy, installed), STFT magnitude -> librosa.feature.rms(S=...)
    = sqrt(2 * sum_k w_k |S_k|^2 / n_fft^2) with the DC and Nyquist bins halved, then the same envelope post-processing."""
    y_filt = scipy.signal.sosfilt(scipy.signal.butter(12, [fmin, fmax], "bp", fs=sr, output="sos"), np.asarray(y, dtype=np.float64))
    n_fft = 2048
    p = stft_power(y_filt, n_fft, 512)
    p[0] *= 0.5
    p[-1] *= 0.5
    env = np.sqrt(2.0 * p.sum(axis=0) / n_fft ** 2)
    env = np.clip(resample(env, n_frames), env.min(), env.max())
    env = torch.from_numpy(env).float()
    env = gaussian_filter(env, smooth, causal=0.05, smf=smf)
    env = percentile_clip(env, clip)
    return env ** power
