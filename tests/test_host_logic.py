"""CPU: host-side logic of the drop-in surface (no GPU calls)."""
import argparse
import inspect
import os

import numpy as np
import pytest
import torch

from oracle import signal_oracle


def test_generate_signature_matches_reference_contract(built_lib):
    """generate(...) keyword contract and defaults (reference generate_audiovisual.py:59-91)."""
    from maua_stylegan2_amd.generate_audiovisual import generate

    sig = inspect.signature(generate)
    expected = dict(initialize=None, get_latents=None, get_noise=None, get_bends=None, get_rewrites=None,
                    get_truncation=None, output_dir="./output", audioreactive_file="audioreactive/examples/default.py",
                    offset=0, duration=-1, latent_file=None, shuffle_latents=False, G_res=1024, out_size=1024, fps=30,
                    latent_count=12, batch=8, dataparallel=False, truncation=1.0, stylegan1=False, noconst=False,
                    latent_dim=512, n_mlp=8, channel_multiplier=2, randomize_noise=False, ffmpeg_preset="slow",
                    base_res_factor=1, output_file=None, args=None)
    names = list(sig.parameters)
    assert names[:2] == ["ckpt", "audio_file"]
    for k, v in expected.items():
        assert sig.parameters[k].default == v, k
    assert names[2:] == list(expected)


def test_render_signature(built_lib):
    from maua_stylegan2_amd.render import render

    names = list(inspect.signature(render).parameters)
    assert names == ["generator", "latents", "noise", "offset", "duration", "batch_size", "out_size", "output_file",
                     "audio_file", "truncation", "bends", "rewrites", "randomize_noise", "ffmpeg_preset"]


def test_noise_range_matches_reference_golden(built_lib, golden):
    from maua_stylegan2_amd.generate_audiovisual import get_noise_range

    g = golden("audioreactive_torch.npz")
    for row in g["noise_range"]:
        out_size, g_res, lo, hi = (int(v) for v in row[:4])
        rmin, rmax, fn = get_noise_range(out_size, g_res, False)
        assert (rmin, rmax) == (lo, hi)
        assert [2 ** fn(s) for s in range(rmin, rmax)] == [int(v) for v in row[4:] if v]


def test_plugin_loader_and_override(tmp_path, built_lib):
    from maua_stylegan2_amd.generate_audiovisual import load_plugin

    f = tmp_path / "plug.py"
    f.write_text("OVERRIDE = dict(fps=24, batch=4)\n\ndef initialize(args):\n    args.hello = 1\n    return args\n\n"
                 "def get_noise(height, width, scale, num_scales, args):\n    return None\n")
    funcs, override = load_plugin(str(f))
    assert override == {"fps": 24, "batch": 4}
    assert funcs["initialize"](argparse.Namespace()).hello == 1
    assert funcs["get_latents"] is None and funcs["get_bends"] is None and callable(funcs["get_noise"])


def test_reference_module_aliases(built_lib):
    import sys

    import maua_stylegan2_amd.generate_audiovisual  # noqa: F401

    import audioreactive as ar
    from models.stylegan2 import Generator  # noqa: F401
    from op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401

    for name in ["onsets", "chroma", "rms", "gaussian_filter", "percentile_clip", "normalize", "compress", "expand",
                 "load_audio", "set_SMF", "chroma_weight_latents", "slerp", "slerp_loops", "spline_loops", "wrapping_slice",
                 "generate_latents", "save_latents", "load_latents", "perlin_noise", "NetworkBend", "AddNoise", "Print",
                 "Translate", "Zoom", "Rotate"]:
        assert hasattr(ar, name), name
    assert "render" in sys.modules


def test_state_dict_keys_equal_seeded_layout(built_lib):
    """The mirror Generator exposes exactly the reference checkpoint keys (pinned by load_state_dict(strict) in
    tests/golden/make_golden.py against the real reference class)."""
    from maua_stylegan2_amd import seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    for size in (32, 256):
        g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        want = seeding.generator_tensor_shapes(size)
        got = {k: tuple(v.shape) for k, v in g.state_dict().items()}
        assert got == {k: tuple(v) for k, v in want.items()}
        g.load_state_dict(seeding.seeded_state_dict(size), strict=True)
        assert (g.n_latent, g.num_layers) == (2 * int(np.log2(size)) - 2, 2 * (int(np.log2(size)) - 2) + 1)


def test_state_dict_layout_equals_reference_fixture(built_lib):
    """tests/golden/state_dict_layout.json: keys (in order) and shapes of the reference Generator's state dict at 256^2
    (135 tensors) and 1024^2 (171 tensors) — what th.load(ckpt)["g_ema"] holds (models/stylegan2.py:458-459)."""
    import json

    from maua_stylegan2_amd import seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    layout = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_layout.json")))
    for size, want in layout.items():
        g = Generator(int(size), 512, 8, channel_multiplier=2, constant_input=True)
        got = [[k, list(v.shape)] for k, v in g.state_dict().items()]
        assert got == want, size
        assert {k: list(v) for k, v in seeding.generator_tensor_shapes(int(size)).items()} == {k: v for k, v in want}, size


def test_filterbanks_match_oracle(built_lib):
    from maua_stylegan2_amd.audioreactive import signal as sig

    for fmin, fmax in [(0.0, None), (20, 8000), (0, 150), (500, 8000)]:
        np.testing.assert_allclose(sig.mel_filterbank(22050, fmin=fmin, fmax=fmax),
                                   signal_oracle.mel_filterbank(22050, fmin=fmin, fmax=fmax), atol=1e-6)
    np.testing.assert_allclose(sig.chroma_filterbank(22050), signal_oracle.chroma_filterbank(22050), atol=1e-6)


def test_band_onset_functions_match_oracle(built_lib):
    """type="mm" onsets: the log-spaced filterbank and the onset-function sum (both host/torch code around the STFT and
    projection kernels) against the oracle's loop restatement."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    for kw in [dict(), dict(fmax=150), dict(fmin=500), dict(fmin=20, fmax=60), dict(num_bands=12, fmin=100.0, fmax=4000.0)]:
        want = signal_oracle.log_filterbank(22050, **kw)
        got = sig.log_filterbank(22050, **kw)
        assert got.shape == want.shape and got.dtype == np.float32
        np.testing.assert_allclose(got, want, atol=1e-7)
        np.testing.assert_allclose(got.sum(1), 1.0, atol=1e-6)
    with pytest.raises(ValueError):
        sig.log_filterbank(22050, fmin=20, fmax=25)
    rng = np.random.default_rng(3)
    filt = (np.abs(rng.standard_normal((40, 97))) * rng.random((40, 1)) * 30).astype(np.float32)
    filt[:, 50:53] = 0.0  # silent frames: the log-ratio term must stay finite
    parts = signal_oracle.madmom_like_onset_functions(filt.T.astype(np.float64))
    got = sig.onset_functions_sum(torch.from_numpy(filt)).numpy()
    assert np.isfinite(got).all() and got[0] == 0.0
    np.testing.assert_allclose(got, sum(parts.values()), rtol=1e-5, atol=1e-4)
    # a step in every band is an onset for every function, a steady spectrogram for none but the log-ratio term (log 2)
    steady = np.ones((5, 8))
    steady[:, 4:] = 3.0
    parts = signal_oracle.madmom_like_onset_functions(steady.T)
    assert all(np.argmax(v) == 4 for v in parts.values())
    np.testing.assert_allclose(parts["spectral_flux"], [0, 0, 0, 0, 10, 0, 0, 0])
    np.testing.assert_allclose(parts["modified_kullback_leibler"][1:4], np.log(2.0))


def test_envelope_postprocessing_matches_golden(built_lib, golden):
    from maua_stylegan2_amd.audioreactive import signal as sig

    g = golden("audioreactive_torch.npz")
    for name in g["pc.cases"]:
        y = sig.percentile_clip(torch.from_numpy(g[f"pc.{name}.x"]).clone(), int(g[f"pc.{name}.p"]))
        np.testing.assert_allclose(y.numpy(), g[f"pc.{name}.y"], atol=1e-6)
    np.testing.assert_allclose(sig.normalize(torch.from_numpy(g["normalize.x"]).clone()).numpy(), g["normalize.y"], atol=1e-6)
    np.testing.assert_allclose(sig.compress(torch.from_numpy(g["compress.x"]).clone(), 0.5, 0.25).numpy(), g["compress.y"], atol=1e-6)


def test_latent_helpers_match_golden(built_lib, golden):
    from maua_stylegan2_amd.audioreactive import latent

    g = golden("audioreactive_torch.npz")
    y = latent.chroma_weight_latents(torch.from_numpy(g["cwl.chroma"]), torch.from_numpy(g["cwl.latents"]))
    np.testing.assert_allclose(y.numpy(), g["cwl.y"], atol=1e-5)
    assert (latent.wrapping_slice(torch.arange(10), 7, 6).numpy() == g["wrapping_slice_10_7_6"]).all()
    np.testing.assert_allclose(latent.spline_loops(g["spline.sel"], 37, 2).numpy(), g["spline.y"], atol=1e-9)


def test_wav_loading_and_cache(tmp_path, built_lib, monkeypatch):
    import scipy.io.wavfile

    from maua_stylegan2_amd import seeding
    from maua_stylegan2_amd.audioreactive.signal import load_audio

    y = seeding.synthetic_audio(2.0, sr=44100)
    path = tmp_path / "clip.wav"
    scipy.io.wavfile.write(str(path), 44100, (y * 32767).astype(np.int16))
    monkeypatch.chdir(tmp_path)
    audio, sr, dur = load_audio(str(path), 0, -1)
    assert sr == 22050 and abs(dur - 2.0) < 1e-3 and audio.dtype == np.float32 and abs(len(audio) - 44100) <= 1
    assert os.path.exists(tmp_path / "workspace")
    audio2, _, _ = load_audio(str(path), 0, -1)
    assert np.array_equal(audio, audio2)
    a3, _, d3 = load_audio(str(path), 0.5, 1.0)
    assert abs(len(a3) - 22050) <= 1 and d3 == 1.0


def test_frame_sink_raw_file(tmp_path, built_lib, monkeypatch):
    from maua_stylegan2_amd import render

    monkeypatch.setattr(render.shutil, "which", lambda name: None)
    sink = render.FrameSink(str(tmp_path / "out.mp4"), 512, 512, 30)
    frame = np.arange(512 * 512 * 3, dtype=np.uint8).reshape(512, 512, 3)
    sink.write(frame), sink.write(frame[::-1].copy())
    sink.close()
    raw = np.fromfile(tmp_path / "out.mp4.rgb24", dtype=np.uint8)
    assert raw.size == 2 * 512 * 512 * 3 and np.array_equal(raw[: frame.size].reshape(frame.shape), frame)
    with pytest.raises(AssertionError, match="does not match"):
        render.FrameSink(None, 512, 512, 30).write(np.zeros((256, 256, 3), np.uint8))
    with pytest.raises(Exception, match="output sizes"):
        render._output_dims(777)


def test_bend_inverse_maps_are_inverses(built_lib):
    """Rotate / Zoom inverse maps composed with the forward (kornia-convention) matrices give identity."""
    from maua_stylegan2_amd.audioreactive import bend

    ang = torch.tensor([30.0, -75.0])
    cw, ch = 40, 28
    m = bend._inverse_maps_rotate(ang, cw, ch).numpy()
    cx, cy = (cw - 1) / 2, (ch - 1) / 2
    for i, a in enumerate(np.deg2rad(ang.numpy())):
        alpha, beta = np.cos(a), np.sin(a)
        fwd = np.array([[alpha, beta, (1 - alpha) * cx - beta * cy], [-beta, alpha, beta * cx + (1 - alpha) * cy], [0, 0, 1]])
        inv = np.vstack([m[i].reshape(2, 3), [0, 0, 1]])
        np.testing.assert_allclose(inv @ fwd, np.eye(3), atol=1e-5)
    s = torch.tensor([[2.0, 0.5]])
    mz = bend._inverse_maps_scale(s, cw, ch).numpy()[0].reshape(2, 3)
    fwd = np.array([[2.0, 0, (1 - 2.0) * cx], [0, 0.5, (1 - 0.5) * cy], [0, 0, 1]])
    np.testing.assert_allclose(np.vstack([mz, [0, 0, 1]]) @ fwd, np.eye(3), atol=1e-6)


def test_chroma_chain_oracle_properties():
    """The unpinned half of the oracle (no librosa here) is at least held to closed-form facts: a pure tone peaks in the
    constant-Q bin / pitch class its frequency dictates, CENS frames are unit-L2 (or zero), the nearest-neighbour median of a
    sequence made of two repeated frames returns those frames."""
    from oracle import signal_oracle as so

    sr = 22050
    for freq, pitch_class in ((220.0, 9), (261.6255653005986, 0), (329.6275569128699, 4)):
        tone = np.sin(2 * np.pi * freq * np.arange(sr // 2) / sr)
        c = so.cqt_magnitude(tone, sr)
        assert int(c[:, 10].argmax()) == int(round(36 * np.log2(freq / 32.70319566257483)))
        assert int(so.chroma_cqt(tone, sr)[:, 10].argmax()) == pitch_class
    rng = np.random.default_rng(0)
    ch = np.abs(rng.standard_normal((12, 90)))
    ch[:, 30:33] = 0
    cens = so.cens_from_chroma(ch)
    norms = np.sqrt((cens ** 2).sum(0))
    assert np.all((np.abs(norms - 1) < 1e-12) | (norms == 0))
    a, b = np.abs(rng.standard_normal(12)), np.abs(rng.standard_normal(12))
    seq = np.stack([a if (t // 5) % 2 == 0 else b for t in range(60)], axis=1)
    np.testing.assert_allclose(so.nn_filter_median(seq), seq, atol=1e-12)


def test_ar_namespace_covers_the_shipped_example_plugins(tmp_path, monkeypatch):
    """Every ``ar.<name>`` the reference's example plugins use (default / temper / kelp / tauceti: onsets, chroma, rms,
    gaussian_filter, spline_loops, wrapping_slice, perlin_noise, AddNoise, plot_signals, ...) exists in the drop-in package.
    The inspection helpers of audioreactive/util.py (diagnostics, not on the hot path) exist with matplotlib
    imported lazily: ``ar.info`` prints array statistics as the reference's does (util.py:11-21), ``from audioreactive.util import
    ...`` works in a plugin, and a plot call (examples/kelp.py:33) draws to workspace/ without a display — or warns and returns
    when matplotlib is missing."""
    import maua_stylegan2_amd.audioreactive as ar

    used = ["gaussian_filter", "onsets", "chroma", "chroma_weight_latents", "wrapping_slice", "spline_loops", "rms",
            "plot_signals", "perlin_noise", "normalize", "load_latents", "laplacian_segmentation", "expand", "AddNoise",
            "info", "plot_spectra", "plot_audio", "plot_chroma_comparison", "raw_chroma", "percentile_clip", "compress",
            "slerp_loops", "generate_latents", "save_latents", "NetworkBend", "Translate", "Zoom", "Rotate", "set_SMF"]
    assert [n for n in used if not hasattr(ar, n)] == []
    monkeypatch.chdir(tmp_path)
    monkeypatch.delenv("DISPLAY", raising=False)
    from maua_stylegan2_amd.audioreactive.util import info, plot_signals  # the import form plugins use

    assert plot_signals is ar.plot_signals
    import contextlib
    import io

    out = io.StringIO()
    with contextlib.redirect_stdout(out):
        info(np.arange(6.0).reshape(2, 3))
        info([torch.ones(3), np.zeros((2, 2))])
    assert out.getvalue().splitlines() == ["[2, 3] 0.00 2.50 5.00", "[([3], '1.00', '1.00', '1.00'), ([2, 2], '0.00', '0.00', '0.00')]"]
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        with pytest.warns(UserWarning, match="matplotlib"):
            assert ar.plot_signals([np.sin(np.linspace(0, 6, 100))]) is None
    else:
        with contextlib.redirect_stdout(io.StringIO()):
            path = ar.plot_signals([np.sin(np.linspace(0, 6, 100)), torch.linspace(0, 1, 50)])
        assert path is not None and (tmp_path / path).exists()


def test_generator_constructor_bookkeeping_matches_reference(built_lib, golden):
    """n_latent, num_layers and the noise-buffer shapes after the output_size / base_res_factor resize
    (reference models/stylegan2.py:395-470) for square, 1920 (2:1), 1080 (1:2) and scaled configurations."""
    from maua_stylegan2_amd.models.stylegan2 import Generator

    for row in golden("generator_meta.npz")["rows"]:
        size, output_size, factor100, n_latent, num_layers = [int(v) for v in row[:5]]
        g = Generator(size, 512, 2, channel_multiplier=2, constant_input=True, output_size=output_size,
                      base_res_factor=factor100 / 100)
        assert (g.n_latent, g.num_layers) == (n_latent, num_layers)
        want = [tuple(int(v) for v in row[5 + 2 * i: 7 + 2 * i]) for i in range(num_layers)]
        got = [tuple(getattr(g.noises, f"noise_{i}").shape[-2:]) for i in range(num_layers)]
        assert got == want, (size, output_size, factor100)


def test_interface_signatures_match_reference_fixture(built_lib):
    """tests/golden/signatures.json records, with inspect from the imported reference, the parameter names and defaults of
    54 callables on the path (generate, render, the ops, every generator module's __init__ / forward, the audioreactive
    signal and latent functions, the bend constructors).  The drop-in's mirrors must lead with exactly those parameters and defaults; the only
    additions allowed are trailing keyword extensions (``device=`` on the envelope functions, ``gradients=`` on
    perlin_noise)."""
    import json

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import op, render
    from maua_stylegan2_amd.audioreactive import bend, latent, signal
    from maua_stylegan2_amd.models import stylegan2

    table = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "signatures.json")))
    assert len(table) >= 54
    roots = {"generate_audiovisual": gav, "render": render, "op": op, "models.stylegan2": stylegan2,
             "audioreactive.signal": signal, "audioreactive.latent": latent, "audioreactive.bend": bend}
    extensions = {"audioreactive.signal.onsets": ["device"], "audioreactive.signal.rms": ["device"],
                  "audioreactive.signal.chroma": ["device"], "audioreactive.latent.perlin_noise": ["gradients"]}
    namespace = table.pop("__namespace__")
    import maua_stylegan2_amd.audioreactive as ar

    assert [n for n in namespace["used_by_example_plugins"] + namespace["exported_callables"] if not hasattr(ar, n)] == []
    for qualified, ref_params in table.items():
        root = max((r for r in roots if qualified.startswith(r + ".")), key=len)
        obj = roots[root]
        for part in qualified[len(root) + 1:].split("."):
            obj = getattr(obj, part)
        mine = [(n, p.default) for n, p in inspect.signature(obj).parameters.items() if n != "self"]
        names = [n for n, _ in mine]
        ref_names = [p[0] for p in ref_params]
        assert names[: len(ref_names)] == ref_names, qualified
        assert names[len(ref_names):] == extensions.get(qualified, []), qualified
        for (name, default), (_, ref_default) in zip(mine, ref_params):
            if ref_default == "<object>":
                continue
            have = None if default is inspect.Parameter.empty else repr(default)
            assert have == ref_default, (qualified, name, have, ref_default)


def test_latent_helpers_match_reference_golden(golden):
    """slerp (scalar and vectorised fractions, parallel vectors) and wrapping_slice (incl. n == 1 and the reference's
    single-wrap behaviour for over-long requests) against values of the reference functions (latent_utils.npz)."""
    from maua_stylegan2_amd.audioreactive import latent

    g = golden("latent_utils.npz")
    a, b, vals = g["slerp.a"], g["slerp.b"], g["slerp.vals"]
    for i, v in enumerate(vals):
        np.testing.assert_allclose(latent.slerp(float(v), a, b), g["slerp.y"][i], atol=1e-12)
    np.testing.assert_allclose(latent.slerp(vals, a, b), g["slerp.y"], atol=1e-12)
    np.testing.assert_allclose(latent.slerp(vals, a, 2.0 * a), g["slerp.parallel"], atol=1e-12)
    for n, start, length in g["wrap.cases"]:
        want = g[f"wrap.{n}_{start}_{length}"]
        got = latent.wrapping_slice(torch.arange(int(n)), int(start), int(length), return_indices=True)
        assert got.dtype == torch.int64 and got.tolist() == want.tolist(), (n, start, length)
        assert latent.wrapping_slice(torch.arange(int(n)) * 3, int(start), int(length)).tolist() == (want * 3).tolist()
    assert any(int(start) == int(n) for n, start, _ in g["wrap.cases"])  # the start == n edge is in the fixture
    with pytest.raises(RuntimeError):  # the reference's arange(start, n) raises for a start beyond the end
        latent.wrapping_slice(torch.arange(10), 12, 3)


def test_complex_flux_and_local_group_delay():
    """The fifth madmom onset function of the reference's default onset envelope (signal.py:63).  (a) The oracle's local group
    delay restates madmom's definition: an impulse d samples after the frame centre has phase -2 pi k d / N against the centre,
    i.e. |lgd| / pi = 2 d / N in every bin, 0 for d = 0.  (b) The device-side torch implementation (run here on CPU tensors)
    equals the oracle on a noisy two-tone signal: same wrapped phase differences, band masks, SuperFlux difference."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    n_fft, hop = 256, 64
    for d in (0, 3, 17):
        y = np.zeros(4 * n_fft)
        centre = 5 * hop
        y[centre + d] = 1.0
        spec = signal_oracle.stft_complex(y, n_fft, hop)[: n_fft // 2]
        lgd = signal_oracle.local_group_delay(spec)[:-1, 5]  # frame 5 is centred on sample 5 * hop
        np.testing.assert_allclose(lgd, 2.0 * d / n_fft, atol=1e-9)
    rng = np.random.default_rng(3)
    sr, n_fft, hop = 22050, 2048, 441
    t = np.arange(2 * sr) / sr
    y = np.sin(2 * np.pi * 440 * t) * (t > 0.7) + 0.5 * np.sin(2 * np.pi * (900 + 40 * np.sin(2 * np.pi * 6 * t)) * t) + 0.05 * rng.standard_normal(t.size)
    spec = signal_oracle.stft_complex(y, n_fft, hop)[: n_fft // 2]
    fb = signal_oracle.log_filterbank(sr, n_fft, 24, 20.0, 8000.0)
    filt = fb @ np.abs(spec)
    want = signal_oracle.complex_flux(spec, fb, filt.T)
    re, im = torch.from_numpy(spec.real.astype(np.float32)), torch.from_numpy(spec.imag.astype(np.float32))
    got = sig.complex_flux(re, im, fb.astype(np.float32), torch.from_numpy(filt.astype(np.float32))).numpy()
    assert abs(int(np.argmax(want[2:])) + 2 - round(0.7 * sr / hop)) <= 2  # the tone onset at 0.7 s is the envelope's peak
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * want.max())
    np.testing.assert_allclose(sig.local_group_delay(re, im).numpy(), signal_oracle.local_group_delay(spec), atol=2e-4)


def test_tuning_estimation_known_answers():
    """librosa.estimate_tuning restated (oracle/signal_oracle.py): tones detuned by a known fraction of a semitone come back as
    that fraction, in 12-bin and in 36-bin units (the unit chroma_cqt uses), an empty pitch set gives 0.  (Tones above 800 Hz:
    the parabolic peak interpolation is good to ~0.3 Hz, which is 2-3 cents at 220 Hz.)"""
    sr = 22050
    t = np.arange(int(2.0 * sr)) / sr
    for cents in (0.0, 20.0, -30.0, 45.0):
        y = sum(np.sin(2 * np.pi * f0 * 2.0 ** (cents / 1200.0) * t) / (i + 1)
                for i, f0 in enumerate((880.0, 1108.73052390749, 1318.51022765149, 1760.0, 2637.02045530296)))
        got12 = signal_oracle.estimate_tuning(y, sr, bins_per_octave=12)
        assert abs(got12 - cents / 100.0) <= 0.02, (cents, got12)
        want36 = np.mod(3 * cents / 100.0 + 0.5, 1.0) - 0.5  # the same deviation against third-of-a-semitone bins
        got36 = signal_oracle.estimate_tuning(y, sr, bins_per_octave=36)
        assert abs(got36 - want36) <= 0.06, (cents, got36, want36)
    assert signal_oracle.pitch_tuning(np.array([440.0 * 2.0 ** (0.25 / 12), 880.0 * 2.0 ** (0.25 / 12)])) == pytest.approx(0.25, abs=0.011)
    assert signal_oracle.pitch_tuning(np.zeros(4)) == 0.0
    assert signal_oracle.estimate_tuning(np.zeros(4096), sr) == 0.0
    # piptrack: one pure tone -> one candidate per frame, at the tone's frequency, nothing outside [fmin, fmax)
    tone = np.sin(2 * np.pi * 1000.0 * t)
    S = np.sqrt(signal_oracle.stft_power(tone, 2048, 512))
    pitch, mag = signal_oracle.piptrack(S, sr)
    mid = pitch[:, 10]
    assert (mid > 0).sum() == 1 and abs(mid.max() - 1000.0) < 1.0 and mag[:, 10].max() > 0
    assert not (signal_oracle.piptrack(np.sqrt(signal_oracle.stft_power(np.sin(2 * np.pi * 100.0 * t), 2048, 512)), sr)[0] > 0).any()
    # the tuning moves the chroma filterbank's and the constant-Q transform's bin centres
    fb0, fb1 = signal_oracle.chroma_filterbank(sr, tuning=0.0), signal_oracle.chroma_filterbank(sr, tuning=0.3)
    assert fb0.shape == fb1.shape and not np.allclose(fb0, fb1)


def pil_bilinear_upscale_restated(frame, x0, y0, cw, ch, ow, oh):
    """The arithmetic of csrc/runtime.hip crop_resize_u8_kernel in numpy: Pillow's 8-bit bilinear resampling for an up-scale
    (2-tap triangle at the rational source position, 22-bit fixed-point coefficients, horizontal pass rounded to uint8, then
    vertical)."""
    bits = 22

    def taps(n_in, n_out):
        o = np.arange(n_out, dtype=np.int64)
        num, den = (2 * o + 1) * n_in - n_out, 2 * n_out
        i0 = np.floor_divide(num, den)
        f = (num - i0 * den).astype(np.float64) / den
        k1 = np.floor(0.5 + f * (1 << bits)).astype(np.int64)
        k0 = np.floor(0.5 + (1 - f) * (1 << bits)).astype(np.int64)
        return np.clip(i0, 0, n_in - 1), np.clip(i0 + 1, 0, n_in - 1), k0, k1

    f = frame[y0:y0 + ch, x0:x0 + cw].astype(np.int64)
    xa, xb, kx0, kx1 = taps(cw, ow)
    ya, yb, ky0, ky1 = taps(ch, oh)
    h = np.clip((kx0[None, :, None] * f[:, xa] + kx1[None, :, None] * f[:, xb] + (1 << (bits - 1))) >> bits, 0, 255)
    v = np.clip((ky0[:, None, None] * h[ya] + ky1[:, None, None] * h[yb] + (1 << (bits - 1))) >> bits, 0, 255)
    return v.astype(np.uint8)


@pytest.mark.parametrize("shape,box", [((1024, 2048, 3), (112, 0, 1824, 1024, 1920, 1080)),   # --out_size 1920 (render.py:97-100)
                                       ((2048, 1024, 3), (0, 112, 1024, 1824, 1080, 1920)),   # --out_size 1080 (:101-104)
                                       ((64, 128, 3), (7, 0, 114, 64, 120, 68))])
def test_device_resize_formula_is_pils_bilinear(shape, box):
    """The device-side wide-output delivery restates Pillow's Image.resize(BILINEAR) for the reference's crop + resize: the numpy
    twin of the kernel's arithmetic equals PIL bit for bit (the kernel itself is compared with PIL in tests/test_render_gpu.py)."""
    import PIL.Image

    x0, y0, cw, ch, ow, oh = box
    frame = np.random.default_rng(sum(shape)).integers(0, 256, shape, dtype=np.uint8)
    want = np.array(PIL.Image.fromarray(np.ascontiguousarray(frame[y0:y0 + ch, x0:x0 + cw])).resize((ow, oh), PIL.Image.BILINEAR))
    assert np.array_equal(pil_bilinear_upscale_restated(frame, x0, y0, cw, ch, ow, oh), want)


def test_parked_heap_is_refcounted_and_respects_a_foreign_freeze():
    """render_shard / bench.time_region park the heap (gc.freeze) for their frame loop.  The freeze is process-global: the last loop out
    thaws it, not the first (two renders in two threads), and a heap the embedding application froze itself stays frozen."""
    import gc

    from maua_stylegan2_amd.render import parked_heap

    assert gc.get_freeze_count() == 0
    with parked_heap():
        assert gc.get_freeze_count() > 0
        with parked_heap():
            pass
        assert gc.get_freeze_count() > 0  # the inner loop's exit must not thaw the heap under the outer one
    assert gc.get_freeze_count() == 0
    gc.freeze()  # the application's own freeze (e.g. a pre-fork server)
    try:
        n = gc.get_freeze_count()
        with parked_heap():
            assert gc.get_freeze_count() >= n
        assert gc.get_freeze_count() > 0  # left as it was found
    finally:
        gc.unfreeze()
    try:
        with parked_heap():
            raise RuntimeError("loop died")
    except RuntimeError:
        pass
    assert gc.get_freeze_count() == 0
