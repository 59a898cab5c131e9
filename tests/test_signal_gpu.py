"""GPU parity: audio-feature / temporal / noise / bend kernels vs golden vectors (pinned stages) and the oracle."""
import numpy as np
import pytest
import torch

from maua_stylegan2_amd import seeding
from oracle import signal_oracle

from chroma_check import check_chroma

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def test_gaussian_filter_golden(gpu, golden):
    from maua_stylegan2_amd.audioreactive import signal as sig

    g = golden("audioreactive_torch.npz")
    for name in g["gf.cases"]:
        sigma, causal, smf, kind = g[f"gf.{name}.cfg"]
        c = None if kind == 0 else (float(causal) if kind == 2 else int(causal))
        sig.set_SMF(float(smf))
        x = torch.from_numpy(g[f"gf.{name}.x"])
        y = sig.gaussian_filter(x, float(sigma) if sigma != int(sigma) else int(sigma), causal=c)
        assert y.device == x.device and y.shape == torch.Size(g[f"gf.{name}.y"].shape)
        np.testing.assert_allclose(y.numpy(), g[f"gf.{name}.y"], atol=1e-5, err_msg=str(name))
        y_dev = sig.gaussian_filter(x.to(gpu), float(sigma) if sigma != int(sigma) else int(sigma), causal=c)
        assert y_dev.is_cuda
    sig.set_SMF(1)


def test_gaussian_filter_noise_sized_vs_oracle(gpu):
    """Noise-shaped input [T,1,h,w] with the sigma of the default plugin's slow noise (radius 512 > T -> wrap branch)."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    x = torch.from_numpy(seeding.seeded_array(0, "gf", (300, 1, 16, 16)))
    for sigma, causal in [(5, None), (128, None), (2, 0.2)]:
        want = signal_oracle.gaussian_filter(x, sigma, causal=causal)
        got = sig.gaussian_filter(x.to(gpu), sigma, causal=causal).cpu()
        np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5)
    # circular shift equivariance at full noise size (property, no oracle needed)
    big = torch.randn(256, 1, 64, 64, device=gpu)
    a = sig.gaussian_filter(torch.roll(big, 17, 0), 5)
    b = torch.roll(sig.gaussian_filter(big, 5), 17, 0)
    assert float((a - b).abs().max()) < 1e-5


@pytest.mark.parametrize("shape,sigma,causal", [
    ((37, 5), 0.3, None), ((37, 5), 1, None), ((64, 300), 2.6, None), ((33, 1000), 7, 0.2), ((50, 3, 7), 40, None),   # radius 1 .. 150 > 3 T
    ((31, 257), 3.9, None), ((32, 256), 4, 0), ((65, 513), 4.1, None), ((900, 18, 8), 20, None), ((7, 4), 5, None),     # 16-row walk residues, T < 32
    ((1, 9), 2, None), ((129, 1), 11, 0.5),
])
def test_gaussian_filter_shapes_vs_oracle(gpu, shape, sigma, causal):
    """The temporal FIR (maua_temporal_fir_f32; reference audioreactive/signal.py:335-343) across its blocking: 32 output times per thread,
    source rows walked 16 at a time, 256 features per workgroup — frame counts either side of 32, feature counts either side of 256, radii of
    every residue, the wrap (radius > T) and zero-beyond-one-wrap (radius = 3 T) branches, causal taps."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    x = torch.from_numpy(seeding.seeded_array(sum(shape), "gfs", shape))
    want = signal_oracle.gaussian_filter(x, sigma, causal=causal)
    got = sig.gaussian_filter(x.to(gpu), sigma, causal=causal).cpu()
    assert got.shape == want.shape
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-5)


def test_gaussian_filter_confines_a_non_finite_sample_to_its_support(gpu):
    """reference audioreactive/signal.py:357-359 (circular pad + conv1d): an inf / NaN sample reaches the outputs within `radius` frames of
    it and no other.  The device kernel walks source rows against zero-PADDED taps (0 * inf = NaN up to 45 frames outside the support):
    threads that meet a non-finite sample recompute tap by tap.  Also: the documented radius limit raises instead of a bare error code."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    T, F, sigma = 400, 300, 3.0
    radius = int(sigma * 4)
    x = torch.from_numpy(seeding.seeded_array(5, "gfinf", (T, F)))
    x[200, 7] = float("inf")
    x[57, 280] = float("nan")
    got = sig.gaussian_filter(x.to(gpu), sigma).cpu()
    want = signal_oracle.gaussian_filter(x, sigma)
    bad = ~torch.isfinite(got)
    assert torch.equal(bad, ~torch.isfinite(want))
    rows7 = torch.nonzero(bad[:, 7]).flatten()
    assert rows7.min() == 200 - radius and rows7.max() == 200 + radius and bad[:, 7].sum() == 2 * radius + 1
    assert bad[:, 280].sum() == 2 * radius + 1 and bad.sum() == 2 * (2 * radius + 1)
    np.testing.assert_allclose(got[~bad].numpy(), want[~bad].numpy(), atol=2e-5)
    with pytest.raises(RuntimeError, match="exceeds the device filter"):
        sig.gaussian_filter(torch.zeros(4000, 2, device=gpu), 2500.0)


def test_stft_mel_chroma_vs_oracle(gpu):
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    y = seeding.synthetic_audio(3.0, sr)
    want = signal_oracle.stft_power(y)
    got = sig.stft_power(y).cpu().numpy()
    assert got.shape == want.shape
    scale = want.max()
    np.testing.assert_allclose(got / scale, want / scale, atol=2e-6)  # fp32 radix-2 FFT vs float64 numpy
    mel_w = signal_oracle.mel_filterbank(sr, fmin=20, fmax=8000) @ want
    mel_g = sig.project(sig.mel_filterbank(sr, fmin=20, fmax=8000), sig.stft_power(y)).cpu().numpy()
    np.testing.assert_allclose(mel_g, mel_w, rtol=2e-4, atol=1e-6 * mel_w.max())
    env_w = signal_oracle.onset_strength(y, sr, fmin=20, fmax=8000)
    env_g = sig.onset_strength(y, sr, fmin=20, fmax=8000).cpu().numpy()
    np.testing.assert_allclose(env_g, env_w, atol=2e-3)
    ch_w = signal_oracle.chroma_stft(y, sr)
    ch_g = sig.raw_chroma(y, sr, type="stft", nearest_neighbor=False)
    np.testing.assert_allclose(ch_g, ch_w, atol=2e-4)


def product_chroma_order(sig, y, sr, n_frames, kind, delivered, margin=16):
    """The pitch class of every column ``sig.chroma`` delivered and the medians it ordered them by, recomputed from the product's own
    stages (signal.py:150-154: harmonic -> raw_chroma -> resample -> argsort(median)); also asserts that ``delivered`` IS that pipeline."""
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        raw = torch.from_numpy(sig.raw_chroma(sig.harmonic(y, margin=margin), sr, type=kind)).to("cuda").t()
    ch = sig.resample(raw, n_frames)
    med = torch.quantile(ch, 0.5, dim=0)
    order = torch.argsort(med)
    rebuilt = ch[:, order]
    rebuilt = (rebuilt / rebuilt.sum(1)[:, None]).float().cpu().numpy()
    np.testing.assert_allclose(delivered, rebuilt, atol=1e-6, err_msg="chroma() is not harmonic -> raw_chroma -> resample -> note selection")
    return order.cpu().numpy(), med[order].cpu().numpy()


def test_onsets_and_chroma_features_vs_oracle(gpu):
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    y = seeding.synthetic_audio(6.0, sr)
    n_frames = 180
    for kind in ("rosa", "mm"):  # "mm" (band-filtered onset-function sum) is the reference's default
        for kw in [dict(fmax=150, smooth=5, clip=97, power=2), dict(fmin=500, smooth=5, clip=99, power=2)]:
            want = signal_oracle.onsets(y, sr, n_frames, type=kind, **kw).numpy()
            got = sig.onsets(y, sr, n_frames, type=kind, **kw)
            assert got.device.type == "cpu" and got.shape == (n_frames,)
            np.testing.assert_allclose(got.numpy(), want, atol=5e-3, err_msg=f"{kind} {kw}")
    np.testing.assert_array_equal(sig.onsets(y, sr, n_frames, fmax=150).numpy(), sig.onsets(y, sr, n_frames, fmax=150, type="mm").numpy())
    with pytest.raises(ValueError):
        sig.onsets(y, sr, n_frames, type="flux")
    env_w = signal_oracle.madmom_like_onset_strength(y, sr, 20.0, 8000.0)
    env_g = sig.onset_strength_bands(y, sr).cpu().numpy()
    np.testing.assert_allclose(env_g, env_w, rtol=2e-4, atol=2e-4 * env_w.max())
    for kind in ("stft", "cqt", "cens"):  # "cens" is the reference's default chroma type; all go through the nn median filter
        want, want_order, want_med = signal_oracle.chroma(y, sr, n_frames, type=kind, nearest_neighbor=True, return_order=True)
        got = sig.chroma(y, sr, n_frames, type=kind).numpy()
        assert np.allclose(got.sum(1), 1.0, atol=1e-5)
        # columns are ordered by their median (reference signal.py:153-154): compared PITCH CLASS FOR PITCH CLASS (tests/chroma_check.py;
        # round 4 re-sorted both sides by column mean, under which any permutation of the pitch classes passed)
        got_order, got_med = product_chroma_order(sig, y, sr, n_frames, kind, got)
        swapped = check_chroma(got, got_order, got_med, want.numpy(), want_order, want_med)
        print(f"[chroma {kind}] delivered order {got_order.tolist()}, oracle {want_order.tolist()}, tie swaps {swapped}")


def test_resample_on_device_matches_scipy(gpu):
    """maua_resample_f64 (Dirichlet-kernel sum in fp64) against scipy.signal.resample: shortening / lengthening, odd and
    even retained-bin counts (the shared Nyquist bin), equal lengths, 1-D and wide inputs."""
    from maua_stylegan2_amd.audioreactive.signal import resample

    r = np.random.default_rng(1)
    cases = [(431, 300, 12), (300, 431, 3), (128, 64, 3), (64, 128, 3), (87, 87, 2), (100, 51, 1), (51, 100, 1), (1293, 900, 1),
             (1292, 900, 20), (7, 2, 1), (2, 7, 1), (1, 3, 1), (3, 1, 1)]
    for n, num, feat in cases:
        x = r.standard_normal((n, feat))
        got = resample(torch.from_numpy(x).to(gpu), num)
        assert got.dtype == torch.float64 and tuple(got.shape) == (num, feat)
        np.testing.assert_allclose(got.cpu().numpy(), signal_oracle.resample(x, num), atol=1e-9, err_msg=f"{n}->{num}")
    x = r.standard_normal(517).astype(np.float32)  # 1-D float32 envelope, as onsets() passes it
    np.testing.assert_allclose(resample(torch.from_numpy(x).to(gpu), 360).cpu().numpy(), signal_oracle.resample(x, 360), atol=1e-9)


def test_perlin_noise_vs_oracle(gpu):
    from maua_stylegan2_amd.audioreactive import latent

    rng = np.random.default_rng(5)
    for shape, res in [((32, 16, 16), (8, 4, 4)), ((24, 8, 8), (1, 1, 1)), ((30, 12, 20), (3, 2, 5))]:
        theta = 2 * np.pi * rng.random((res[0] + 1, res[1] + 1, res[2] + 1))
        phi = 2 * np.pi * rng.random((res[0] + 1, res[1] + 1, res[2] + 1))
        want = signal_oracle.perlin_noise(shape, res, theta.copy(), phi.copy())
        g = np.stack((np.sin(phi) * np.cos(theta), np.sin(phi) * np.sin(theta), np.cos(phi)), axis=3)
        g[-1] = g[0]  # tileable along time
        got = latent.perlin_noise(shape, res, gradients=g).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-5)
        assert abs(got[0] - got[-1]).max() < 1.0  # continuous loop along time
    out = latent.perlin_noise((16, 8, 8), (4, 2, 2))
    assert out.is_cuda and out.shape == (16, 8, 8) and float(out.abs().max()) < 3


def test_perlin_noise_vs_reference_golden(gpu, golden):
    """The HIP Perlin kernel against outputs of the REFERENCE's perlin_noise (tests/golden/perlin.npz): with numpy's
    global generator seeded as the fixture was, the drop-in draws the same gradient angles and must reproduce the
    reference's tensor (fp32 kernel vs the reference's fp64 arithmetic)."""
    from maua_stylegan2_amd.audioreactive import latent

    g = golden("perlin.npz")
    for name in g["cases"]:
        cfg = [int(v) for v in g[f"{name}.cfg"]]
        shape, res, tileable, seed = tuple(cfg[:3]), tuple(cfg[3:6]), tuple(bool(v) for v in cfg[6:9]), cfg[9]
        np.random.seed(seed)
        got = latent.perlin_noise(shape, res, tileable)
        assert got.is_cuda and tuple(got.shape) == shape
        np.testing.assert_allclose(got.cpu().numpy(), g[f"{name}.y"], atol=2e-5, err_msg=str(name))


def test_bends_vs_oracle(gpu):
    from maua_stylegan2_amd.audioreactive import bend

    rng = np.random.default_rng(9)
    b, c, h, w = 3, 4, 8, 12
    x = rng.standard_normal((b, c, h, w)).astype(np.float32)
    xd = torch.from_numpy(x).to(gpu)
    # Translate: scroll by fractional pixel amounts, with the reference's 5w-wide noise canvas
    noise = (0.1 * rng.standard_normal((1, 1, h, 5 * w))).astype(np.float32)
    t = torch.tensor([[0.0, 0.0], [3.5, 0.0], [-7.25, 0.0]])
    mod = bend.Translate(t, h, w, torch.from_numpy(noise))
    pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]  # three STACKED pads (reference bend.py:60-64)
    want = signal_oracle.affine_reflect_warp(x, bend._inverse_maps_translate(t).numpy(), pads, noise)
    np.testing.assert_allclose(mod(xd).cpu().numpy(), want, atol=1e-5)
    # the point of the stacked pads (examples/tauceti.py:142-144): scrolling by a whole width w shows the same features
    # as scrolling by 0, so the saw-tooth modulation loops seamlessly
    zero = torch.zeros(1, 1, h, 5 * w)
    scroll = lambda dx: bend.Translate(torch.tensor([[dx, 0.0]] * b), h, w, zero)(xd).cpu().numpy()  # noqa: E731
    np.testing.assert_allclose(scroll(float(w)), scroll(0.0), atol=1e-6)
    assert np.abs(scroll(w / 2.0) - scroll(0.0)).max() > 0.1
    # zero translation: the asymmetric reference padding (2.5 w left, 1.5 w right) + centre crop shows source columns
    # [-w/2, w/2) -> the reflected left half followed by the left half
    got0 = mod(xd).cpu().numpy()[0] - noise[0, 0][:, 2 * w: 3 * w][None]
    np.testing.assert_allclose(got0[:, :, w // 2:], x[0][:, :, : w // 2], atol=1e-6)
    np.testing.assert_allclose(got0[:, :, : w // 2], x[0][:, :, w // 2: 0: -1], atol=1e-6)
    # Zoom / Rotate about the canvas centre
    z = torch.tensor([1.0, 1.7, 0.6])
    pad = max(h, w) - 1
    want = signal_oracle.affine_reflect_warp(x, bend._inverse_maps_scale(z, w + 2 * pad, h + 2 * pad).numpy(), (pad,) * 4)
    got = bend.Zoom(z, h, w)(xd).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-5)
    np.testing.assert_allclose(got[0], x[0], atol=1e-6)  # zoom 1 = identity
    a = torch.tensor([0.0, 33.0, -120.0])
    pad = int(max(h, w) * (1 - np.sqrt(2) / 2))
    want = signal_oracle.affine_reflect_warp(x, bend._inverse_maps_rotate(a, w + 2 * pad, h + 2 * pad).numpy(), (pad,) * 4)
    got = bend.Rotate(a, h, w)(xd).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-5)
    np.testing.assert_allclose(got[0], x[0], atol=1e-6)


def test_rms_envelope_vs_oracle(gpu):
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    y = seeding.synthetic_audio(6.0, sr)
    want = signal_oracle.rms(y, sr, 180, smooth=10, clip=50, power=2).numpy()
    got = sig.rms(y, sr, 180, smooth=10, clip=50, power=2)
    assert got.shape == (180,) and got.device.type == "cpu"
    np.testing.assert_allclose(got.numpy(), want, atol=5e-3)


def test_generate_latents_maps_through_the_mapping_network(gpu):
    """generate_latents = mapping network (8 x EqualLinear + fused leaky ReLU on the HIP op) applied to z, repeated to
    n_latent — the evident intent of the reference; checked against the oracle's mapping_network."""
    from maua_stylegan2_amd.models.stylegan2 import Generator
    from oracle import stylegan2_oracle as so

    sd = seeding.seeded_state_dict(32, seed=3)
    g = Generator(32, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(sd)
    g = g.to(gpu)
    z = torch.from_numpy(seeding.seeded_array(4, "z", (5, 512)))
    got = g(z.to(gpu), map_latents=True).cpu()
    want = so.mapping_network(sd, z)
    assert got.shape == (5, g.n_latent, 512)
    np.testing.assert_allclose(got[:, 0].numpy(), want.numpy(), atol=2e-4, rtol=1e-3)
    assert torch.equal(got[:, 0], got[:, -1])


def test_hpss_pieces_vs_oracle(gpu):
    """Complex STFT / inverse STFT round trip, 31-tap median filters (both axes, reflect boundary) vs scipy.ndimage, and the
    full harmonic / percussive separation vs the oracle restatement of librosa's hpss."""
    import scipy.ndimage

    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.audioreactive import signal as sig

    lib = _lib.load()
    r = np.random.default_rng(2)
    x = np.abs(r.standard_normal((70, 45))).astype(np.float32)
    xd = torch.from_numpy(x).to(gpu)
    for axis, size in [(0, 31), (1, 31), (0, 5), (1, 9)]:
        out = torch.empty_like(xd)
        _lib.check(lib.maua_median_filter_f32(xd.data_ptr(), out.data_ptr(), 70, 45, size, axis, _lib.stream_ptr()), "median")
        shape = (size, 1) if axis == 0 else (1, size)
        want = scipy.ndimage.median_filter(x, size=shape, mode="reflect")
        assert np.array_equal(out.cpu().numpy(), want), (axis, size)

    sr = 22050
    y = seeding.synthetic_audio(3.0, sr)
    harm, perc = sig.hpss(y, margin=1.0)
    # margin 1 with split masks: the two components add back up to the signal (masks sum to one, istft(stft) = identity)
    recon = (harm + perc).cpu().numpy()
    assert np.abs(recon - y).max() < 1e-4
    for margin in (1.0, 8.0):
        wh, wp = signal_oracle.hpss(y, margin)
        gh, gp = sig.hpss(y, margin)
        scale = np.abs(y).max()
        assert np.abs(gh.cpu().numpy() - wh).max() < 2e-4 * scale + 1e-5
        assert np.abs(gp.cpu().numpy() - wp).max() < 2e-4 * scale + 1e-5
    # the kick lives in the percussive part, the chord in the harmonic part
    gh, gp = sig.hpss(y, 4.0)
    assert float(gp.abs().max()) > 0.05 and float(gh.std()) > 0.02


def test_cens_and_nn_filter_vs_oracle(gpu):
    """Chroma post-processing kernels (signal.py:115,131 roles): CENS quantise/smooth/normalise and the nearest-neighbour
    median filter, on random chromagrams (ragged lengths, zero frames) and on the chromagram of the synthetic track."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    rng = np.random.default_rng(4)
    cases = [np.abs(rng.standard_normal((12, t))).astype(np.float32) for t in (3, 40, 41, 257, 700)]
    cases[1][:, 5:9] = 0.0  # silent frames: L1 / L2 / cosine norms of zero
    cases.append(sig.raw_chroma(seeding.synthetic_audio(8.0), 22050, type="stft", nearest_neighbor=False).astype(np.float32))
    for ch in cases:
        want = signal_oracle.cens_from_chroma(ch)
        got = sig.cens(torch.from_numpy(ch)).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=2e-6)
        if ch.shape[1] > 2:
            want = signal_oracle.nn_filter_median(ch)
            got = sig.nn_filter(torch.from_numpy(ch)).cpu().numpy()
            # the neighbour order comes from fp64 cosine similarities on both sides; allow a vanishing number of
            # near-tie swaps (they move a median by one order statistic)
            close = np.isclose(got, want, atol=1e-6)
            assert close.mean() > 0.999, (ch.shape, close.mean())


def test_nn_filter_long_track_path_equals_lds_path(gpu):
    """Tracks beyond ~16k chroma frames keep the similarity rows in a device workspace and stride the frames over a fixed
    grid (round 1 returned them unfiltered).  The two paths of the same kernel must agree bit for bit: a caller-supplied
    workspace selects the workspace path at any size (include/maua_hip.h), here on a clip with more frames than workspace
    rows (1024) so that workgroups really walk several frames."""
    import math

    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.audioreactive import signal as sig

    rng = np.random.default_rng(14)
    t = 1500
    ch = torch.from_numpy(np.abs(rng.standard_normal((12, t))).astype(np.float32)).cuda()
    lds_path = sig.nn_filter(ch).cpu().numpy()
    lib = _lib.load()
    k = int(min(t - 1, 2 * math.ceil(math.sqrt(t - 1))))
    assert lib.maua_nn_median_ws_doubles(12, t, k) == 0  # fits LDS: nn_filter took the LDS path
    ws = torch.empty(1024 * t, dtype=torch.float64, device="cuda")
    out = torch.empty_like(ch)
    _lib.check(lib.maua_nn_median_f32(ch.data_ptr(), out.data_ptr(), 12, t, k, 1, ws.data_ptr(), _lib.stream_ptr(ch.device)),
               "maua_nn_median_f32")
    np.testing.assert_array_equal(out.cpu().numpy(), lds_path)
    # and a genuinely long sequence runs (20k frames: 160 KB of similarities per frame) and stays a median of its inputs
    long = np.abs(rng.standard_normal((12, 20000))).astype(np.float32)
    assert lib.maua_nn_median_ws_doubles(12, 20000, 284) == 1024 * 20000
    out = sig.nn_filter(torch.from_numpy(long)).cpu().numpy()
    assert out.shape == long.shape and np.isfinite(out).all() and out.min() >= long.min() and out.max() <= long.max()
    assert 0.3 < float(np.median(out)) < 1.0  # medians of |N(0,1)| samples sit near 0.67


def test_constant_q_transform_vs_oracle(gpu):
    """Direct constant-Q magnitude (252 bins from C1, hop 512) and the chroma fold against the oracle, on the synthetic
    track, on a pure tone (peak bin / pitch class known in closed form) and on a clip shorter than the longest filter."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    tone = np.sin(2 * np.pi * 220.0 * np.arange(sr) / sr).astype(np.float32)
    c = sig.cqt_magnitude(tone, sr).cpu().numpy()
    assert c.shape == (252, 1 + sr // 512) and int(c[:, 20].argmax()) == 99  # 36 * log2(220 / C1) = 99 exactly
    assert int(sig.raw_chroma(tone, sr, type="cqt", nearest_neighbor=False)[:, 20].argmax()) == 9  # pitch class A
    for y in (seeding.synthetic_audio(1.5, sr), seeding.synthetic_audio(0.4, sr)):
        want = signal_oracle.cqt_magnitude(y, sr)
        got = sig.cqt_magnitude(y, sr).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=1e-6 + 1e-5 * want.max())
        np.testing.assert_allclose(sig.raw_chroma(y, sr, type="cqt", nearest_neighbor=False), signal_oracle.chroma_cqt(y, sr),
                                   atol=2e-4)


def test_slerp_loops_matches_reference_golden(gpu, golden):
    """slerp_loops against the reference function (latent_utils.npz; run there with the float32 cast its own
    gaussian_filter needs): key spacing, Gaussian smoothing on the HIP FIR kernel, tiling and the ragged tail."""
    from maua_stylegan2_amd.audioreactive import latent, signal as sig

    g = golden("latent_utils.npz")
    sig.set_SMF(1)
    sel = g["slerp_loops.sel"]
    for tag in "abc":
        n_frames, n_loops, smoothing, loop = (int(v) for v in g[f"slerp_loops.{tag}.cfg"])
        y = latent.slerp_loops(sel, n_frames, n_loops, smoothing, bool(loop))
        assert list(y.shape) == g[f"slerp_loops.{tag}.shape"].tolist() and y.dtype == torch.float32
        np.testing.assert_allclose(y.cpu().numpy()[:, ::6, :], g[f"slerp_loops.{tag}.y"], atol=1e-5, err_msg=tag)


@pytest.mark.gpu
def test_tuning_estimation_vs_oracle(gpu):
    """Device piptrack / estimate_tuning (what librosa's chroma functions run when tuning is None) against the oracle's
    restatement: same candidates on a seeded spectrogram, same tuning on detuned tone stacks and on the synthetic track, and the
    estimate reaches the constant-Q transform (a detuned track keeps its pitch classes)."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    y = seeding.synthetic_audio(4.0, sr)
    S = np.sqrt(signal_oracle.stft_power(y, 2048, 512)).astype(np.float32)
    want_p, want_m = signal_oracle.piptrack(S, sr)
    got_p, got_m = (t.cpu().numpy() for t in sig.piptrack(S, sr))
    assert ((got_p > 0) == (want_p > 0)).mean() > 0.9999  # a bin sitting exactly on the 0.1 x max gate may flip in fp32
    both = (got_p > 0) & (want_p > 0)
    np.testing.assert_allclose(got_p[both], want_p[both], rtol=1e-4)
    np.testing.assert_allclose(got_m[both], want_m[both], rtol=1e-4)
    for bpo in (12, 36):
        assert abs(sig.estimate_tuning(y, sr, bins_per_octave=bpo) - signal_oracle.estimate_tuning(y, sr, bins_per_octave=bpo)) <= 0.0101
    t = np.arange(int(2.0 * sr)) / sr
    tones = (880.0, 1108.73052390749, 1318.51022765149, 1760.0, 2637.02045530296)
    stacks = {}
    for cents in (0.0, 20.0, -30.0):
        stacks[cents] = sum(np.sin(2 * np.pi * f0 * 2.0 ** (cents / 1200.0) * t) / (i + 1) for i, f0 in enumerate(tones)).astype(np.float32)
        got = sig.estimate_tuning(stacks[cents], sr)
        assert abs(got - cents / 100.0) <= 0.02 and abs(got - signal_oracle.estimate_tuning(stacks[cents], sr)) <= 0.0101
        power = sig.stft_power(stacks[cents])
        assert abs(sig.estimate_tuning(S=power, sr=sr) - signal_oracle.estimate_tuning(S=signal_oracle.stft_power(stacks[cents]), sr=sr)) <= 0.0101
    assert sig.estimate_tuning(np.zeros(8192, np.float32), sr) == 0.0
    # A-major-ish stack: the pitch classes that carry energy are the same with and without detuning (the transform follows it)
    base = sig.raw_chroma(stacks[0.0], sr, type="cqt", nearest_neighbor=False)
    sharp = sig.raw_chroma(stacks[20.0], sr, type="cqt", nearest_neighbor=False)
    assert set(np.argsort(base[:, 40])[-3:]) == set(np.argsort(sharp[:, 40])[-3:]) == {9, 1, 4}  # A, C#, E


# ---- round 5: known-answer tests on the HIP path — what can be KNOWN without librosa / madmom / kornia -------------


def test_pure_tones_land_in_their_own_pitch_class(gpu):
    """A tone (plus its octave) at every one of the 12 equal-tempered pitches C4 ... B4 must dominate ITS chroma bin — C = 0, as
    librosa's chroma functions number them (reference audioreactive/signal.py:102-133) — for all three chromagram types of the device path
    (STFT filterbank, constant-Q, CENS).  A permutation or rotation of the pitch classes anywhere in the HIP chain fails this."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr = 22050
    t = np.arange(int(2.0 * sr)) / sr
    for kind in ("stft", "cqt", "cens"):
        hits = []
        for k in range(12):
            f0 = 261.6255653 * 2.0 ** (k / 12.0)
            y = (0.6 * np.sin(2 * np.pi * f0 * t) + 0.3 * np.sin(2 * np.pi * 2 * f0 * t)).astype(np.float32)
            raw = sig.raw_chroma(y, sr, type=kind, nearest_neighbor=False)
            assert raw.shape[0] == 12
            profile = raw[:, 10:-10].mean(1)
            hits.append(int(np.argmax(profile)))
            assert profile[k] > 2.0 * np.delete(profile, k).max(), (kind, k, profile)
        assert hits == list(range(12)), (kind, hits)


def test_click_train_gives_one_onset_peak_per_beat(gpu):
    """120 BPM click train (a 30 ms noise burst every 0.5 s), 8 s at 30 fps: the onset envelope of the device path — HPSS percussive
    component, band onset functions incl. complex flux ("mm", the reference's default) or mel spectral flux ("rosa"), resampled to the
    frame rate (signal.py:31-73) — has exactly one peak above a quarter of its maximum per beat, within one frame of the beat."""
    from maua_stylegan2_amd.audioreactive import signal as sig

    sr, secs, fps = 22050, 8.0, 30
    n = int(secs * fps)
    rng = np.random.default_rng(1)
    y = np.zeros(int(secs * sr), np.float32)
    beats = np.arange(0.5, secs - 0.25, 0.5)
    for b in beats:
        i, length = int(b * sr), int(0.03 * sr)
        y[i: i + length] += (rng.standard_normal(length) * np.exp(-np.arange(length) / (0.005 * sr))).astype(np.float32)
    y += 0.001 * rng.standard_normal(len(y)).astype(np.float32)
    want = [int(round(b * fps)) for b in beats]
    for kind in ("mm", "rosa"):
        env = sig.onsets(y, sr, n, type=kind).numpy()
        peaks = [i for i in range(1, n - 1) if env[i] > env[i - 1] and env[i] >= env[i + 1] and env[i] > 0.25 * env.max()]
        assert len(peaks) == len(want) and all(abs(p - q) <= 1 for p, q in zip(peaks, want)), (kind, peaks, want)


def test_translate_known_answers_on_the_device(gpu):
    """audioreactive/bend.py:52-70 (three stacked reflection pads -> translate -> centre crop), known answers that need no kornia: the
    feature map itself sits half a width right of the crop window, so a translation by -w/2 is the IDENTITY (bit for bit: integer
    sample positions), and a scroll by exactly one width shows the same features as no scroll at all (the seamless loop the reference's
    docstring promises); both for the eager module and for the captured per-frame form (run_static through a frame source row)."""
    from maua_stylegan2_amd.audioreactive import bend

    h, w = 16, 16
    x = torch.from_numpy(seeding.seeded_array(31, "bend_x", (3, 8, h, w))).to(gpu)
    zero = torch.zeros(1, 1, h, 5 * w, device=gpu)

    def run(tx):
        mod = torch.tensor([[tx, 0.0]] * x.shape[0], device=gpu)
        return bend.Translate(mod, h, w, zero)(x)

    assert torch.equal(run(-w / 2), x), "Translate by -w/2 must return the feature map itself"
    np.testing.assert_allclose(run(float(w)).cpu().numpy(), run(0.0).cpu().numpy(), atol=1e-6)
    assert float((run(0.0) - x).abs().max()) > 0.5 and float((run(w / 4) - run(0.0)).abs().max()) > 0.5  # (the transform is not a no-op)
    # Zoom by 1 and Rotate by 0 degrees are identities as well (centre-preserving maps on the padded canvas)
    np.testing.assert_allclose(bend.Zoom(torch.ones(x.shape[0], device=gpu), h, w)(x).cpu().numpy(), x.cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(bend.Rotate(torch.zeros(x.shape[0], device=gpu), h, w)(x).cpu().numpy(), x.cpu().numpy(), atol=1e-6)
    # Rotate by 180 degrees about the centre = flip of both axes
    np.testing.assert_allclose(bend.Rotate(torch.full((x.shape[0],), 180.0, device=gpu), h, w)(x).cpu().numpy(),
                               x.flip(2, 3).cpu().numpy(), atol=1e-4)
