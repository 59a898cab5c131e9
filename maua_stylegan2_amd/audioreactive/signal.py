"""Audio-feature library with the reference's ``audioreactive.signal`` surface, computed on the MI355X.

Mirrors /root/reference/audioreactive/signal.py (names, arguments, return shapes):
  set_SMF :21-23 · onsets :31-73 · rms :76-99 · raw_chroma :102-133 · chroma :136-156 · normalize :243-254 ·
  percentile :257-268 · percentile_clip :271-292 · compress/expand :295-316 · gaussian_filter :319-368 ·
  load_audio :371-405.

What runs where: the STFT / inverse STFT (LDS radix-2 FFT), the median filters and soft masks of the harmonic /
percussive separation, the mel / log-band / chroma / rms filterbank projections, the constant-Q transform, the CENS and
nearest-neighbour chroma post-processing, the Fourier-method envelope resampling and the temporal Gaussian FIR are HIP
kernels (csrc/signal.hip) behind the C ABI; the remaining O(n_frames) envelope post-processing (clip, percentile,
power, the onset-function sum) is a handful of torch ops on the same device.  Inputs may be numpy arrays or tensors on
any device; results come back on the device the reference would have produced them on (envelopes: CPU tensors) unless
``device=`` says otherwise, so existing plugins keep working while the heavy tensors can stay in HBM.

Divergences from the reference, all stated in DESIGN.md §7 ("parity unpinned" rows): librosa / madmom are not
dependencies here.  ``onsets(type="mm")`` sums madmom's five onset functions (incl. the phase-based complex flux) on a
log-filtered spectrogram built here, ``type="rosa"`` is the mel spectral flux; ``chroma`` runs harmonic
separation -> tuning estimation -> constant-Q chromagram -> CENS -> nearest-neighbour median filter,
and madmom's "deep" / "clp" chroma models fall back to the constant-Q chromagram with a warning.
"""
import math
import os
import warnings
from pathlib import Path

import numpy as np
import scipy.signal
import torch as th

from .. import _lib

SMF = 1  # smoothing factor, set by generate() from the rendering fps (reference :18-23)


def set_SMF(smf):
    global SMF
    SMF = smf


def _dev():
    return th.device("cuda", th.cuda.current_device())


def _to_dev(x, dtype=th.float32):
    if isinstance(x, np.ndarray):
        x = th.from_numpy(np.ascontiguousarray(x))
    return x.to(device=_dev(), dtype=dtype).contiguous()


_DEVICE_CONSTANTS = {}  # (device, dtype, shape, content hash) -> device tensor, insertion-ordered (oldest dropped beyond 48 entries)


def _const_dev(arr, dtype=None, device=None):
    """A host-built CONSTANT table (window, filterbank, CQT frequencies ...) on the device, uploaded once per content: an upload from pageable
    memory blocks the host until everything queued on the stream before it has run, so re-uploading the same half-megabyte filterbank in every
    call made the front end wait for its own kernels three times per job (0.075 s of a 0.9 s warm generate(), round 6 profile).  The result is
    shared: callers must not write to it."""
    arr = np.ascontiguousarray(arr if dtype is None else np.asarray(arr).astype(dtype, copy=False))
    if device is not None and th.device(device).type != "cuda":  # (host-side callers, e.g. the CPU tests of complex_flux)
        return th.from_numpy(arr.copy())
    dev = _dev()
    key = (dev.index, arr.dtype.str, arr.shape, hash(arr.tobytes()))
    hit = _DEVICE_CONSTANTS.get(key)
    if hit is None:
        hit = th.from_numpy(arr.copy()).to(dev)
        _DEVICE_CONSTANTS[key] = hit
        while len(_DEVICE_CONSTANTS) > 48:
            _DEVICE_CONSTANTS.pop(next(iter(_DEVICE_CONSTANTS)))
    return hit


# ------------------------------------------------------------------------------------------------ filterbanks (host, once)
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) / (np.log(6.4) / 27.0)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))


def mel_filterbank(sr, n_fft=2048, n_mels=128, fmin=0.0, fmax=None):
    """Slaney mel scale, triangular, area-normalised — [n_mels, n_fft/2+1] float32."""
    fmax = sr / 2.0 if fmax is None else float(fmax)
    freqs = np.arange(1 + n_fft // 2) * (sr / n_fft)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    width = np.diff(edges)
    ramps = edges[:, None] - freqs[None, :]
    lower = -ramps[:-2] / width[:-1, None]
    upper = ramps[2:] / width[1:, None]
    fb = np.maximum(0.0, np.minimum(lower, upper))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]
    return fb.astype(np.float32)


def chroma_filterbank(sr, n_fft=2048, n_chroma=12, ctroct=5.0, octwidth=2.0, tuning=0.0):
    """STFT-bin -> pitch-class weights (C-based; ``tuning`` = deviation of A440 in fractions of a bin) — [12, n_fft/2+1]
    float32."""
    f = np.arange(1, n_fft) * (sr / n_fft)
    pos = n_chroma * np.log2(f / (27.5 * 2.0 ** (tuning / n_chroma)))
    pos = np.concatenate([[pos[0] - 1.5 * n_chroma], pos])
    bw = np.concatenate([np.maximum(np.diff(pos), 1.0), [1.0]])
    dist = pos[None, :] - np.arange(n_chroma)[:, None]
    half = round(n_chroma / 2)
    dist = np.remainder(dist + half + 10 * n_chroma, n_chroma) - half
    w = np.exp(-0.5 * (2.0 * dist / bw[None, :]) ** 2)
    w /= np.maximum(np.linalg.norm(w, axis=0, keepdims=True), np.finfo(float).tiny)
    w *= np.exp(-0.5 * ((pos / n_chroma - ctroct) / octwidth) ** 2)[None, :]
    w = np.roll(w, -3 * (n_chroma // 12), axis=0)
    return np.ascontiguousarray(w[:, : 1 + n_fft // 2]).astype(np.float32)


# ------------------------------------------------------------------------------------------------ device primitives
def stft_power(audio, n_fft=2048, hop=512):
    """|STFT|^2 on device: [n_fft/2+1, 1 + len/hop] float32 (centred, reflect-padded, periodic Hann)."""
    lib = _lib.load()
    y = _to_dev(audio)
    n = y.numel()
    n_frames = 1 + n // hop
    win = _hann(n_fft, y.device)
    p = th.empty((n_fft // 2 + 1, n_frames), dtype=th.float32, device=y.device)
    with th.cuda.device(y.device):
        _lib.check(lib.maua_stft_power_f32(y.data_ptr(), n, win.data_ptr(), n_fft, hop, p.data_ptr(), n_frames,
                                           _lib.stream_ptr(y.device)), "maua_stft_power_f32")
    return p


def project(fb, p, to_db=False, amin=1e-10):
    """fb [M,K] (numpy or tensor) @ p [K,N] on device, optional 10*log10(max(amin, .))."""
    lib = _lib.load()
    fbt = _const_dev(fb, np.float32) if isinstance(fb, np.ndarray) else _to_dev(fb)
    m, k = fbt.shape
    n = p.shape[1]
    out = th.empty((m, n), dtype=th.float32, device=p.device)
    with th.cuda.device(p.device):
        _lib.check(lib.maua_filterbank_f32(fbt.data_ptr(), p.data_ptr(), out.data_ptr(), m, k, n, int(to_db), float(amin),
                                           _lib.stream_ptr(p.device)), "maua_filterbank_f32")
    return out


def _hann(n_fft, device):
    return _const_dev((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32))


def hpss(audio, margin=1.0, kernel_size=31, power=2.0, n_fft=2048, hop=512):
    """Harmonic / percussive source separation on device (librosa.effects.hpss as the reference uses it through
    effects.percussive / effects.harmonic, signal.py:49,150): complex STFT -> |D| median-filtered along time
    (harmonic-enhanced) and along frequency (percussive-enhanced), soft masks with ``margin``, inverse STFT.
    Returns (y_harmonic, y_percussive) as device tensors of the input length."""
    lib = _lib.load()
    y = _to_dev(audio)
    n = y.numel()
    n_frames = 1 + n // hop
    n_bins = n_fft // 2 + 1
    dev = y.device
    win = _hann(n_fft, dev)
    re = th.empty((n_bins, n_frames), dtype=th.float32, device=dev)
    im = th.empty_like(re)
    outs = []
    with th.cuda.device(dev):
        st = _lib.stream_ptr(dev)
        _lib.check(lib.maua_stft_complex_f32(y.data_ptr(), n, win.data_ptr(), n_fft, hop, re.data_ptr(), im.data_ptr(),
                                             n_frames, st), "maua_stft_complex_f32")
        mag = th.sqrt(re * re + im * im)
        harm, perc = th.empty_like(mag), th.empty_like(mag)
        _lib.check(lib.maua_median_filter_f32(mag.data_ptr(), harm.data_ptr(), n_bins, n_frames, kernel_size, 1, st),
                   "maua_median_filter_f32")
        _lib.check(lib.maua_median_filter_f32(mag.data_ptr(), perc.data_ptr(), n_bins, n_frames, kernel_size, 0, st),
                   "maua_median_filter_f32")
        split = int(margin == 1)
        frames_ws = th.empty((n_frames, n_fft), dtype=th.float32, device=dev)
        for own, other in ((harm, perc), (perc, harm)):
            mre, mim = th.empty_like(re), th.empty_like(im)
            _lib.check(lib.maua_softmask_apply_f32(re.data_ptr(), im.data_ptr(), own.data_ptr(), other.data_ptr(),
                                                   float(margin), float(power), split, mre.data_ptr(), mim.data_ptr(),
                                                   re.numel(), st), "maua_softmask_apply_f32")
            out = th.empty(n, dtype=th.float32, device=dev)
            _lib.check(lib.maua_istft_f32(mre.data_ptr(), mim.data_ptr(), win.data_ptr(), n_fft, hop, n_frames,
                                          frames_ws.data_ptr(), out.data_ptr(), n, st), "maua_istft_f32")
            outs.append(out)
    return outs[0], outs[1]


def percussive(audio, margin=1.0):
    return hpss(audio, margin)[1]


def harmonic(audio, margin=1.0):
    return hpss(audio, margin)[0]


def resample(x, num):
    """Fourier-method resampling along dim 0 (what scipy.signal.resample does at reference :68,152) on device:
    maua_resample_f64, float64 in and out like scipy."""
    src = _to_dev(x, th.float64).contiguous()
    n = src.shape[0]
    num = int(num)
    out = th.empty((num,) + tuple(src.shape[1:]), dtype=th.float64, device=src.device)
    features = max(src.numel() // max(n, 1), 1)
    with th.cuda.device(src.device):
        _lib.check(_lib.load().maua_resample_f64(src.data_ptr(), n, features, out.data_ptr(), num,
                                                 _lib.stream_ptr(src.device)), "maua_resample_f64")
    return out


TEMPORAL_FIR_MAX_RADIUS = 8143  # MAUA_TEMPORAL_FIR_MAX_RADIUS (include/maua_hip.h)


def gaussian_filter(x, sigma, causal=None):
    """Circular Gaussian smoothing along time (dim 0), HIP FIR kernel; same tap construction as reference :335-343."""
    lib = _lib.load()
    src_device = x.device if isinstance(x, th.Tensor) else th.device("cpu")
    xd = _to_dev(x)
    dim = xd.dim()
    n_frames = xd.shape[0]
    lifted = xd
    while lifted.dim() < 3:
        lifted = lifted[:, None]
    radius = min(int(sigma * 4 * SMF), 3 * len(lifted))
    taps = th.arange(-radius, radius + 1, dtype=th.float32)
    taps = th.exp(-0.5 / sigma ** 2 * taps ** 2)
    if causal is not None:
        taps[radius + 1:] *= 0 if not isinstance(causal, float) else causal
    taps = (taps / taps.sum()).to(xd.device)
    if radius > n_frames:
        print(f"WARNING: Gaussian filter radius ({int(sigma * 4 * SMF)}) is larger than number of frames ({n_frames}).\n"
              f"\t Filter size has been lowered to ({radius}). You might want to consider lowering sigma ({sigma}).")
    feats = xd.numel() // max(n_frames, 1)
    y = th.empty_like(xd)
    if radius > TEMPORAL_FIR_MAX_RADIUS:  # (include/maua_hip.h: the taps live in LDS)
        raise RuntimeError(f"gaussian_filter: radius {radius} frames (sigma {sigma}) exceeds the device filter's {TEMPORAL_FIR_MAX_RADIUS} taps "
                           "each side — a Gaussian that wide is a mean over the clip; lower sigma")
    with th.cuda.device(xd.device):
        _lib.check(lib.maua_temporal_fir_f32(xd.data_ptr(), taps.data_ptr(), y.data_ptr(), n_frames, feats, radius,
                                             _lib.stream_ptr(xd.device)), "maua_temporal_fir_f32")
    if dim < 3:
        y = y.reshape(list(xd.shape) + [1] * (3 - dim)).squeeze()
    return y.to(src_device)


# ------------------------------------------------------------------------------------------------ envelope post-processing
def normalize(signal):
    """Rescale to [0, 1] IN PLACE and return the same object (reference :243-254 mutates its argument too)."""
    low = signal.min()
    span = signal.max() - low
    signal -= low
    signal /= span
    return signal


def percentile(signal, p):
    """Value at the p-th percentile by rank (no interpolation): the (1 + round(p% of n-1))-th smallest (reference :257-268)."""
    ordered = signal.reshape(-1).sort().values
    return ordered[round(0.01 * float(p) * (ordered.numel() - 1))].item()


def percentile_clip(signal, p):
    """Clamp to [0, p-th percentile of the strict local maxima], then scale the maximum to 1 (reference :271-292).  The two
    end points are never maxima (each is compared with itself there)."""
    is_peak = th.zeros(signal.shape[0], dtype=th.bool, device=signal.device)
    middle = signal[1:-1]
    is_peak[1:-1] = (middle > signal[2:]) & (middle > signal[:-2])
    clipped = signal.clamp(0, percentile(signal[is_peak], p))
    return clipped.div_(clipped.max())


def compress(signal, threshold, ratio, invert=False):
    """Multiply everything above (``invert``: below) ``threshold`` by ``ratio`` in place, then :func:`normalize`
    (reference :295-311)."""
    selected = (signal < threshold) if invert else (signal > threshold)
    signal *= th.where(selected, float(ratio), 1.0).to(signal.dtype)
    return normalize(signal)


expand = compress  # same operation; which one it is depends on threshold / ratio (reference :314-316)


# ------------------------------------------------------------------------------------------------ features
def onset_strength(audio, sr, fmin=0.0, fmax=None, n_fft=2048, hop=512, n_mels=128):
    """Spectral-flux onset envelope on device (mel power -> dB -> positive first difference -> mean over bands,
    shifted by lag + n_fft//(2*hop) frames)."""
    fmax = sr / 2.0 if fmax is None else fmax
    p = stft_power(audio, n_fft, hop)
    db = project(mel_filterbank(sr, n_fft, n_mels, fmin, fmax), p, to_db=True)
    db = th.maximum(db, db.max() - 80.0)
    flux = th.clamp(db[:, 1:] - db[:, :-1], min=0).mean(0)
    pad = 1 + n_fft // (2 * hop)
    env = th.cat([th.zeros(pad, device=flux.device), flux])
    return env[: db.shape[1]]


def log_filterbank(sr, n_fft=2048, num_bands=24, fmin=20.0, fmax=8000.0, fref=440.0):
    """Triangular filters on a logarithmic frequency grid, [n_filters, n_fft/2] float32 (host-built constant like
    :func:`mel_filterbank`): ``num_bands`` centres per octave around ``fref`` inside [fmin, fmax], snapped to FFT bins
    with duplicates dropped, each filter normalised to unit sum — the spectrogram filtering the reference gets from
    madmom's LogarithmicFilterbank (signal.py:57)."""
    n_bins = n_fft // 2
    k = np.arange(math.floor(math.log2(fmin / fref) * num_bands), math.ceil(math.log2(fmax / fref) * num_bands) + 1)
    f = fref * np.exp2(k / num_bands)
    f = f[(f >= fmin) & (f <= fmax)]
    c = np.unique(np.clip(np.rint(f * n_fft / sr).astype(np.int64), 0, n_bins - 1))
    if c.size < 3:
        raise ValueError(f"onsets: no log-spaced bands between {fmin} and {fmax} Hz at n_fft={n_fft}")
    cols = np.arange(n_bins)[None, :]
    lo, mid, hi = c[:-2, None], c[1:-1, None], c[2:, None]
    rise = (cols - lo) / (mid - lo)
    fall = (hi - cols) / (hi - mid)
    fb = np.where((cols >= lo) & (cols < mid), rise, np.where((cols >= mid) & (cols < hi), fall, 0.0))
    return (fb / fb.sum(axis=1, keepdims=True)).astype(np.float32)


def onset_functions_sum(filt):
    """Sum of four onset detection functions of a filtered magnitude spectrogram ``filt`` [bands, frames] (any device):
    squared and plain positive flux, flux against the 3-band maximum of the previous frame, and the mean log-ratio
    between consecutive frames — the phase-free members of the sum at signal.py:58-67 (the fifth is :func:`complex_flux`)."""
    prev = th.cat([filt[:, :1], filt[:, :-1]], dim=1)
    pos = (filt - prev).clamp(min=0)
    widened = th.nn.functional.max_pool1d(prev.t()[None], 3, 1, 1)[0].t()
    sup = (filt - widened).clamp(min=0)
    ratio = th.log1p(filt.double() / (prev.double() + float(np.finfo(np.float64).eps))).float()
    keep = th.ones(filt.shape[1], device=filt.device)
    keep[0] = 0.0
    return (pos * pos).sum(0) + pos.sum(0) + (sup.sum(0) + ratio.mean(0)) * keep


def stft_complex(audio, n_fft=2048, hop=512):
    """Complex STFT on device as (re, im), each [n_fft/2+1, 1 + len/hop] float32 (centred, reflect-padded, periodic Hann)."""
    lib = _lib.load()
    y = _to_dev(audio)
    n = y.numel()
    n_frames = 1 + n // hop
    win = _hann(n_fft, y.device)
    re = th.empty((n_fft // 2 + 1, n_frames), dtype=th.float32, device=y.device)
    im = th.empty_like(re)
    with th.cuda.device(y.device):
        _lib.check(lib.maua_stft_complex_f32(y.data_ptr(), n, win.data_ptr(), n_fft, hop, re.data_ptr(), im.data_ptr(), n_frames,
                                             _lib.stream_ptr(y.device)), "maua_stft_complex_f32")
    return re, im


def local_group_delay(re, im):
    """|local group delay| / pi of a complex STFT [bins, frames] measured against the frame centre (madmom's
    circular_shift=True, reference signal.py:55): (-1)^k on bin k, phase differenced along frequency with the jumps wrapped to
    (-pi, pi] — the difference of the unwrapped phase IS the wrapped difference of the phase — last bin 0."""
    sign = 1.0 - 2.0 * (th.arange(re.shape[0], device=re.device) % 2).float()
    phase = th.atan2(im * sign[:, None], re * sign[:, None]).double()
    jump = phase[1:] - phase[:-1]
    wrapped = th.remainder(jump + math.pi, 2 * math.pi) - math.pi
    wrapped = th.where((wrapped == -math.pi) & (jump > 0), th.full_like(wrapped, math.pi), wrapped)  # numpy.unwrap's tie rule
    lgd = th.zeros_like(phase)
    lgd[:-1] = -wrapped
    return (lgd.abs() / math.pi).float()


def complex_flux(re, im, fb, filt):
    """madmom's complex_flux (reference signal.py:63) on device tensors: the SuperFlux difference of the filtered
    spectrogram ``filt`` [bands, frames], weighted per band by the minimum — over the FFT bins the band's filter spans, widened
    by one bin on each side — of the 3-frame temporal maximum of the local group delay.  ``fb`` [bands, bins] numpy."""
    lgd = local_group_delay(re, im)
    lgd = th.nn.functional.max_pool1d(th.nn.functional.pad(lgd[None], (1, 1), mode="replicate"), 3, 1)[0]
    support = np.asarray(fb) != 0
    first = support.argmax(axis=1)
    last = support.shape[1] - 1 - support[:, ::-1].argmax(axis=1)
    n_bins = lgd.shape[0]
    width = int((last - first).max()) + 3
    rows = np.clip(first[:, None] - 1 + np.arange(width)[None, :], 0, n_bins - 1)  # [bands, width] bin indices
    keep = (np.arange(width)[None, :] <= (last - first + 2)[:, None])              # beyond a band's span: ignored (inf)
    idx = _const_dev(rows, device=lgd.device)
    gathered = lgd[idx.reshape(-1)].reshape(rows.shape[0], width, -1)
    gathered = th.where(_const_dev(keep, device=lgd.device)[:, :, None], gathered, th.full_like(gathered, float("inf")))
    mask = gathered.min(dim=1).values                                              # [bands, frames]
    prev = th.cat([filt[:, :1], filt[:, :-1]], dim=1)
    widened = th.nn.functional.max_pool1d(th.nn.functional.pad(prev.t()[None], (1, 1), mode="replicate"), 3, 1)[0].t()
    diff = (filt - widened).clamp(min=0)
    diff[:, 0] = 0.0
    return (diff * mask).sum(0)


def onset_strength_bands(audio, sr, fmin=20.0, fmax=8000.0, n_fft=2048, hop=441):
    """type="mm" onset envelope: complex STFT (frame 2048, hop 441 — 50 frames/s at 22050 Hz) -> magnitudes through the
    24-per-octave log filterbank -> the five onset functions of reference signal.py:58-67: :func:`onset_functions_sum` (spectral
    difference, spectral flux, SuperFlux, modified Kullback-Leibler) + :func:`complex_flux`."""
    re, im = stft_complex(audio, n_fft, hop)
    re, im = re[: n_fft // 2].contiguous(), im[: n_fft // 2].contiguous()
    fb = log_filterbank(sr, n_fft, 24, fmin, fmax)
    filt = project(fb, th.sqrt(re * re + im * im))
    return onset_functions_sum(filt) + complex_flux(re, im, fb, filt)


def onsets(audio, sr, n_frames, margin=8, fmin=20, fmax=8000, smooth=1, clip=100, power=1, type="mm", device=None):
    y_perc = percussive(audio, margin=margin) if margin else audio  # reference :49 (margin=0/None skips the separation)
    if type == "rosa":
        env = onset_strength(y_perc, sr, fmin=fmin, fmax=fmax)
    elif type == "mm":
        env = onset_strength_bands(y_perc, sr, fmin=fmin, fmax=fmax)
    else:
        raise ValueError(f"onsets: unknown type {type!r} (expected 'mm' or 'rosa')")
    onset = resample(env, n_frames).clamp(float(env.min()), float(env.max())).float()
    onset = gaussian_filter(onset, smooth, causal=0)
    onset = percentile_clip(onset, clip)
    onset = onset ** power
    return onset.to(device if device is not None else "cpu")


def rms(y, sr, n_frames, fmin=20, fmax=8000, smooth=180, clip=50, power=6, device=None):
    y_filt = scipy.signal.sosfilt(scipy.signal.butter(12, [fmin, fmax], "bp", fs=sr, output="sos"), np.asarray(y))
    n_fft = 2048
    p = stft_power(y_filt.astype(np.float32), n_fft, 512)
    wts = np.ones((1, n_fft // 2 + 1), np.float32)
    wts[0, 0] = wts[0, -1] = 0.5
    env = th.sqrt(project(wts * (2.0 / n_fft ** 2), p))[0]
    env = resample(env, n_frames).clamp(float(env.min()), float(env.max())).float()
    env = gaussian_filter(env, smooth, causal=0.05)
    env = percentile_clip(env, clip)
    env = env ** power
    return env.to(device if device is not None else "cpu")


def cens(ch, win_len=41):
    """CENS post-processing of a [n_bins, T] chromagram on the device (maua_chroma_cens_f32): L1 normalise, quantise,
    Hann-smooth over ``win_len`` frames, L2 normalise — the steps of librosa.feature.chroma_cens after its chromagram."""
    ch = _to_dev(ch).float().contiguous()
    out = th.empty_like(ch)
    with th.cuda.device(ch.device):
        _lib.check(_lib.load().maua_chroma_cens_f32(ch.data_ptr(), out.data_ptr(), ch.shape[0], ch.shape[1], win_len,
                                                    _lib.stream_ptr(ch.device)), "maua_chroma_cens_f32")
    return out


def nn_filter(ch, width=1):
    """Nearest-neighbour median filter of a [n_bins, T] sequence on the device (maua_nn_median_f32): per frame, the median
    over the k = 2 ceil(sqrt(T - 2 width + 1)) most cosine-similar frames (reference signal.py:131).  Tracks whose similarity
    rows no longer fit LDS (> ~16k frames, 6 min of audio) run the same kernel on a device workspace."""
    ch = _to_dev(ch).float().contiguous()
    t = ch.shape[1]
    k = int(min(t - 1, 2 * math.ceil(math.sqrt(max(t - 2 * width + 1, 1)))))
    if k < 1:
        return ch
    out = th.empty_like(ch)
    lib = _lib.load()
    n_ws = lib.maua_nn_median_ws_doubles(ch.shape[0], t, k)
    ws = th.empty(n_ws, dtype=th.float64, device=ch.device) if n_ws else None
    with th.cuda.device(ch.device):
        _lib.check(lib.maua_nn_median_f32(ch.data_ptr(), out.data_ptr(), ch.shape[0], t, k, width, _lib.ptr(ws),
                                          _lib.stream_ptr(ch.device)), "maua_nn_median_f32")
    return out


def piptrack(S, sr, n_fft=2048, fmin=150.0, fmax=4000.0, threshold=0.1):
    """Pitch candidates of a device spectrogram ``S`` [1 + n_fft/2, T] (librosa.piptrack's published algorithm): parabolic
    interpolation around every local maximum above ``threshold`` x the frame maximum inside [fmin, fmax) -> (pitches in Hz,
    interpolated magnitudes), zero elsewhere.  Elementwise device work next to the HIP STFT."""
    S = _to_dev(S).abs()
    k = S.shape[0]
    freqs = th.arange(k, device=S.device, dtype=th.float32) * (float(sr) / n_fft)
    avg = 0.5 * (S[2:] - S[:-2])
    curv = 2 * S[1:-1] - S[2:] - S[:-2]
    shift = avg / (curv + (curv.abs() < th.finfo(th.float32).tiny).float())
    avg = th.nn.functional.pad(avg, (0, 0, 1, 1))
    shift = th.nn.functional.pad(shift, (0, 0, 1, 1))
    gated = S * (S > threshold * S.max(dim=0, keepdim=True).values)
    before = th.cat([gated[:1], gated[:-1]], 0)  # edge padding: the first row is never strictly greater than itself
    after = th.cat([gated[1:], gated[-1:]], 0)
    peak = ((freqs >= max(fmin, 0.0)) & (freqs < min(fmax, sr / 2.0)))[:, None] & (gated > before) & (gated >= after)
    bins = th.arange(k, device=S.device, dtype=th.float32)[:, None]
    zero = th.zeros((), device=S.device)
    return th.where(peak, (bins + shift) * (float(sr) / n_fft), zero), th.where(peak, S + 0.5 * avg * shift, zero)


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    """Histogram peak of the deviation of ``frequencies`` (device tensor, Hz) from the bins of an A440 scale, in fractions of a
    bin in [-0.5, 0.5) (librosa.pitch_tuning)."""
    f = frequencies.reshape(-1).double()
    f = f[f > 0]
    if f.numel() == 0:
        return 0.0
    residual = th.remainder(bins_per_octave * th.log2(f / 27.5), 1.0)
    residual = th.where(residual >= 0.5, residual - 1.0, residual)
    edges = th.linspace(-0.5, 0.5, int(math.ceil(1.0 / resolution)) + 1, dtype=th.float64, device=f.device)
    which = (th.bucketize(residual, edges, right=True) - 1).clamp_(0, edges.numel() - 2)  # numpy.histogram's bin rule
    counts = th.bincount(which, minlength=edges.numel() - 1)
    return float(edges[int(counts.argmax())])


def estimate_tuning(audio=None, sr=22050, S=None, n_fft=2048, resolution=0.01, bins_per_octave=12):
    """Tuning deviation of a track in fractions of a bin (librosa.estimate_tuning, which librosa's chroma_cqt / chroma_cens /
    chroma_stft run when ``tuning`` is None — reference signal.py:115-119 leaves it None): piptrack on |STFT| (hop n_fft/4) or on
    a given spectrogram, the candidates at least as strong as the median one, pitch_tuning."""
    if S is None:
        S = stft_power(audio, n_fft, n_fft // 4).sqrt()
    pitch, mag = piptrack(S, sr, n_fft)
    sel = pitch > 0
    if not bool(sel.any()):
        return 0.0
    strengths = mag[sel].sort().values
    n = strengths.numel()
    median = 0.5 * (strengths[(n - 1) // 2] + strengths[n // 2])
    return pitch_tuning(pitch[sel & (mag >= median)], resolution, bins_per_octave)


CQT_FMIN = 32.70319566257483  # C1, librosa's default


def cqt_magnitude(audio, sr, hop=512, n_bins=252, fmin=CQT_FMIN, bins_per_octave=36):
    """|constant-Q transform| on the device, [n_bins, 1 + len/hop] float32 (maua_cqt_mag_f32): 7 octaves x 36 bins from
    C1 as librosa.feature.chroma_cqt asks of librosa.cqt, evaluated directly from the definition."""
    y = _to_dev(audio).float().contiguous()
    freqs = (fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave))
    q = 1.0 / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    lengths = np.ceil(q * sr / freqs).astype(np.int32)
    n_frames = 1 + y.numel() // hop
    out = th.empty((n_bins, n_frames), dtype=th.float32, device=y.device)
    f_dev = _const_dev(freqs.astype(np.float32))
    l_dev = _const_dev(lengths)
    with th.cuda.device(y.device):
        _lib.check(_lib.load().maua_cqt_mag_f32(y.data_ptr(), y.numel(), f_dev.data_ptr(), l_dev.data_ptr(), n_bins, hop,
                                                float(sr), out.data_ptr(), n_frames, _lib.stream_ptr(y.device)),
                   "maua_cqt_mag_f32")
    return out


def cq_to_chroma_matrix(n_bins=252, bins_per_octave=36, n_chroma=12):
    """[12, n_bins] fold of constant-Q bins onto pitch classes (every semitone owns the 3 bins centred on it; row 0 = C)."""
    merge = bins_per_octave // n_chroma
    w = np.zeros((n_chroma, n_bins), dtype=np.float32)
    semitone = (np.arange(n_bins) + merge // 2) // merge
    w[semitone % n_chroma, np.arange(n_bins)] = 1.0
    return w


def raw_chroma(audio, sr, type="cens", nearest_neighbor=True):
    """[12, n_frames] numpy chromagram (reference :102-133).  ``type``: "stft" = STFT filterbank chromagram; "cqt" =
    constant-Q chromagram (direct CQT on the device); "cens" (the default) = the constant-Q chromagram with the CENS
    post-processing; the madmom "deep" / "clp" models are not built (constant-Q chromagram + a warning).
    ``nearest_neighbor`` adds the median filter over cosine-nearest frames (:131)."""
    if type not in ("stft", "cqt", "cens"):
        warnings.warn(f"chroma type {type!r}: madmom's chroma models are not on this path; using the constant-Q chromagram",
                      stacklevel=2)
    if type == "stft":  # librosa estimates the tuning from the power spectrogram here, from |STFT| of the signal below
        power = stft_power(audio)
        raw = project(chroma_filterbank(sr, tuning=estimate_tuning(S=power, sr=sr, bins_per_octave=12)), power)
    else:
        tuning = estimate_tuning(audio, sr, bins_per_octave=36)
        raw = project(cq_to_chroma_matrix(), cqt_magnitude(audio, sr, fmin=CQT_FMIN * 2.0 ** (tuning / 36)))
    peak = raw.max(dim=0, keepdim=True).values
    ch = raw / th.where(peak > 0, peak, th.ones_like(peak))
    if type == "cens":
        ch = cens(ch)
    if nearest_neighbor:
        ch = th.minimum(ch, nn_filter(ch))
    return ch.cpu().numpy()


def chroma(audio, sr, n_frames, margin=16, type="cens", notes=12, device=None):
    y_harm = harmonic(audio, margin=margin) if margin else audio  # reference :150
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ch = th.from_numpy(raw_chroma(y_harm, sr, type=type)).to(_dev()).t()
    ch = resample(ch, n_frames)
    keep = th.argsort(th.quantile(ch, 0.5, dim=0))[:notes]  # np.median semantics (mean of the two middle values)
    ch = ch[:, keep]
    ch = (ch / ch.sum(1)[:, None]).float()
    return ch.to(device if device is not None else "cpu")


def laplacian_segmentation(signal, sr, k=5, plot=False):
    raise NotImplementedError("laplacian_segmentation is outside the hot path (SURVEY.md §2 row 7: only kelp.py uses it)")


# ------------------------------------------------------------------------------------------------ audio loading
def _read_audio_file(audio_file, target_sr=22050):
    """WAV (PCM / float) via scipy; anything else needs ffmpeg on PATH.  Returns mono float32 at ``target_sr``."""
    import scipy.io.wavfile

    path = str(audio_file)
    if not path.lower().endswith(".wav"):
        import shutil
        import subprocess

        if shutil.which("ffmpeg") is None:
            raise RuntimeError(f"cannot decode {path}: only .wav is readable without an ffmpeg binary on PATH")
        raw = subprocess.run(["ffmpeg", "-v", "error", "-i", path, "-f", "f32le", "-ac", "1", "-ar", str(target_sr), "-"],
                             check=True, capture_output=True).stdout
        return np.frombuffer(raw, dtype=np.float32).copy(), target_sr
    sr, data = scipy.io.wavfile.read(path)
    if data.dtype.kind == "i":
        data = data.astype(np.float32) / float(np.iinfo(data.dtype).max + 1)
    elif data.dtype.kind == "u":
        data = (data.astype(np.float32) - 128.0) / 128.0
    data = data.astype(np.float32)
    if data.ndim == 2:
        data = data.mean(axis=1)
    if sr != target_sr:
        g = math.gcd(int(sr), int(target_sr))
        data = scipy.signal.resample_poly(data, target_sr // g, sr // g).astype(np.float32)
    return data, target_sr


def load_audio(audio_file, offset=0, duration=-1, cache=True):
    """Reference :371-405: (audio float32 mono @22050 Hz, sr, duration); cached under workspace/<stem>*.npy."""
    full, sr = None, 22050
    cache_file = None
    audio, sr = _read_audio_file(audio_file, 22050)
    audio_dur = len(audio) / sr
    if duration == -1 or audio_dur < duration:
        duration = audio_dur
        if offset != 0:
            duration -= offset
    if cache:
        cache_file = (f"workspace/{Path(audio_file).stem}" + ("" if duration == -1 else f"_length{duration}")
                      + ("" if offset == 0 else f"_start{offset}") + ".npy")
        if os.path.exists(cache_file):
            return np.load(cache_file), sr, duration
    start = int(round(offset * sr))
    audio = audio[start: start + int(round(duration * sr))]
    if cache_file is not None:
        os.makedirs("workspace", exist_ok=True)
        np.save(cache_file, audio)
    return audio, sr, duration
