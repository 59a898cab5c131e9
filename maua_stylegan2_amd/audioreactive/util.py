"""Inspection helpers of the ``ar.*`` namespace (role of /root/reference/audioreactive/util.py:11-99): array statistics
and quick plots of envelopes, spectrograms and chromagrams.  They are diagnostics, not part of the synthesis path: data
is pulled to the host, figures are drawn with matplotlib (imported lazily — without it the plot helpers warn and return, ``info``
needs none; ``show`` falls back to saving a PNG under ``workspace/`` when no display is available), spectra come from this
package's own device kernels instead of librosa.
"""
import os

import numpy as np

__all__ = ["info", "plot_signals", "plot_spectra", "plot_audio", "plot_chroma_comparison"]

PITCH_CLASSES = ["C", "C#", "D", "D#", "E", "F", "F#", "G", "G#", "A", "A#", "B"]


def _host(array):
    """numpy view of a tensor / array-like, wherever it lives."""
    if hasattr(array, "detach"):
        array = array.detach().cpu().numpy()
    return np.asarray(array)


def _summary(array):
    a = _host(array)
    return list(a.shape), f"{a.min():.2f}", f"{a.mean():.2f}", f"{a.max():.2f}"


def info(arr):
    """Print shape / min / mean / max of one array or of every array in a list."""
    if isinstance(arr, (list, tuple)):
        print([_summary(a) for a in arr])
    else:
        print(*_summary(arr))


def _pyplot():
    try:
        import matplotlib

        if not os.environ.get("DISPLAY") and matplotlib.get_backend().lower() not in ("agg", "pdf", "svg", "ps"):
            matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except ImportError:  # an unmodified plugin that plots (examples/kelp.py:33) must keep running on a box without matplotlib
        import warnings

        warnings.warn("ar.plot_*: matplotlib is not installed; the plot is skipped")
        return None
    return plt


def _finish(plt, name):
    plt.tight_layout()
    if plt.get_backend().lower() == "agg":
        os.makedirs("workspace", exist_ok=True)
        path = os.path.join("workspace", f"{name}.png")
        plt.savefig(path)
        plt.close()
        print(f"saved {path}")
        return path
    plt.show()
    return None


def plot_signals(signals):
    """One stacked line plot per 1-D signal (envelopes such as onsets / rms)."""
    plt = _pyplot()
    if plt is None:
        return None
    signals = list(signals)
    plt.figure(figsize=(16, 4 * len(signals)))
    for row, sig in enumerate(signals, start=1):
        plt.subplot(len(signals), 1, row)
        plt.plot(_host(sig).squeeze())
    return _finish(plt, "signals")


def plot_spectra(spectra, chroma=False):
    """One image per spectrogram ([bins, frames]; [frames, 12] chromagrams are transposed), pitch-class ticks for chroma."""
    plt = _pyplot()
    if plt is None:
        return None
    spectra = list(spectra)
    fig, axes = plt.subplots(len(spectra), 1, figsize=(16, 4 * len(spectra)), squeeze=False)
    for ax, spec in zip(axes[:, 0], spectra):
        spec = _host(spec)
        if spec.ndim == 2 and spec.shape[1] == 12 and spec.shape[0] != 12:
            spec = spec.T
        ax.imshow(spec, origin="lower", aspect="auto", interpolation="nearest")
        ax.set_xlabel("frame")
        if chroma and spec.shape[0] == 12:
            ax.set_yticks(range(12))
            ax.set_yticklabels(PITCH_CLASSES)
    return _finish(plt, "spectra")


def plot_audio(audio, sr):
    """Mel power spectrogram in dB (128 bands, referenced to its maximum) of an audio signal."""
    from . import signal as sig

    plt = _pyplot()
    if plt is None:
        return None
    mel_db = sig.project(sig.mel_filterbank(sr), sig.stft_power(audio), to_db=True).cpu().numpy()
    plt.figure(figsize=(16, 9))
    plt.imshow(mel_db - mel_db.max(), origin="lower", aspect="auto", interpolation="nearest")
    plt.colorbar(format="%+2.f dB")
    plt.xlabel("frame")
    plt.ylabel("mel band")
    return _finish(plt, "audio")


def plot_chroma_comparison(audio, sr):
    """Side-by-side chromagrams of the strategies this path implements (cens, cqt, stft)."""
    from . import signal as sig

    plt = _pyplot()
    if plt is None:
        return None
    kinds = ["cens", "cqt", "stft"]
    fig, axes = plt.subplots(1, len(kinds), figsize=(16, 5), squeeze=False)
    for ax, kind in zip(axes[0], kinds):
        ax.imshow(sig.raw_chroma(audio, sr, type=kind), origin="lower", aspect="auto", interpolation="nearest")
        ax.set_title(kind)
        ax.set_yticks(range(12))
        ax.set_yticklabels(PITCH_CLASSES)
        ax.label_outer()
    return _finish(plt, "chroma_comparison")
