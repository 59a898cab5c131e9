"""CPU: the oracle/ restatement reproduces every golden vector captured from the imported reference
(tests/golden/make_golden.py).  This is what pins the oracle on boxes where /root/reference does not exist."""
import numpy as np
import pytest
import torch

from maua_stylegan2_amd import seeding
from oracle import ops_oracle, signal_oracle, stylegan2_oracle as so

torch.set_grad_enabled(False)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def test_upfirdn2d_cases(golden):
    g = golden("ops_upfirdn2d.npz")
    for name in g["cases"]:
        up, down, p0, p1 = (int(v) for v in g[f"{name}.cfg"])
        y = ops_oracle.upfirdn2d(t(g[f"{name}.x"]), t(g[f"{name}.k"]), up=up, down=down, pad=(p0, p1))
        np.testing.assert_allclose(y.numpy(), g[f"{name}.y"], atol=1e-5, err_msg=str(name))
        loops = ops_oracle.upfirdn2d_loops(g[f"{name}.x"], g[f"{name}.k"], up, down, (p0, p1))
        np.testing.assert_allclose(loops, g[f"{name}.y"], atol=1e-4, err_msg=str(name))


def test_fused_leaky_relu_cases(golden):
    g = golden("ops_fused_leaky_relu.npz")
    for name in g["cases"]:
        y = ops_oracle.fused_leaky_relu(t(g[f"{name}.x"]), t(g[f"{name}.b"]))
        np.testing.assert_allclose(y.numpy(), g[f"{name}.y"], atol=1e-6)
        k = ops_oracle.fused_bias_act_kernel_semantics(g[f"{name}.x"], g[f"{name}.b"], None, 3, 0, 0.2, 2 ** 0.5)
        np.testing.assert_allclose(k, g[f"{name}.y"], atol=1e-6)


def test_modulated_conv_cases(golden):
    g = golden("layers.npz")
    blur = t(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0))
    for name in g["modconv.cases"]:
        cin, cout, k, up, demod = (int(v) for v in g[f"modconv.{name}.cfg"])
        p = lambda key: t(g[f"modconv.{name}.{key}"])  # noqa: E731
        y = so.modulated_conv2d(p("x"), p("s"), p("w"), p("mw"), p("mb"), demodulate=bool(demod), upsample=bool(up),
                                blur_kernel=blur)
        np.testing.assert_allclose(y.numpy(), g[f"modconv.{name}.y"], atol=2e-5, err_msg=str(name))


@pytest.mark.parametrize("name,up", [("styled_plain", False), ("styled_up", True)])
def test_styled_conv(golden, name, up):
    g = golden("layers.npz")
    sd = {k[len(name) + 4:]: t(g[k]) for k in g.files if k.startswith(name + ".sd.")}
    y = so.styled_conv(sd, "L", t(g[f"{name}.x"]), t(g[f"{name}.s"]), t(g[f"{name}.noise"]), up)
    np.testing.assert_allclose(y.numpy(), g[f"{name}.y"], atol=2e-5)


@pytest.mark.parametrize("name", ["torgb_noskip", "torgb_skip"])
def test_to_rgb(golden, name):
    g = golden("layers.npz")
    sd = {k[len(name) + 4:]: t(g[k]) for k in g.files if k.startswith(name + ".sd.")}
    skip = t(g[f"{name}.skip"]) if f"{name}.skip" in g.files else None
    y = so.to_rgb(sd, "L", t(g[f"{name}.x"]), t(g[f"{name}.s"]), skip)
    np.testing.assert_allclose(y.numpy(), g[f"{name}.y"], atol=2e-5)


@pytest.mark.parametrize("size", [8, 16, 64])
def test_generator_small(golden, size):
    g = golden(f"gen_{size}.npz")
    batch, stride = int(g["batch"]), int(g["stride"])
    s_sd, s_lat, s_noise, s_tl = (int(v) for v in g["seeds"])
    sd = seeding.seeded_state_dict(size, seed=s_sd)
    n_latent = int(np.log2(size)) * 2 - 2
    lat = seeding.seeded_latents(batch, n_latent, seed=s_lat)
    noise = seeding.seeded_noise(batch, size, seed=s_noise)
    tl = t(seeding.seeded_array(s_tl, "truncation_latent", (1, 512)))
    img, acts = so.generator_forward(sd, lat, noise, truncation=torch.full((batch,), float(g["truncation"])),
                                     truncation_latent=tl, return_activations=True)
    np.testing.assert_allclose(img.numpy()[:, :, ::stride, ::stride], g["image"], atol=2e-4)
    np.testing.assert_allclose([a.abs().mean().item() for a in acts], g["act_mean_abs"], rtol=1e-4)
    img2 = so.generator_forward(sd, lat, None)
    np.testing.assert_allclose(img2.numpy()[:, :, ::stride, ::stride], g["image_buffer_noise"], atol=2e-4)


def test_state_dict_layout_1024():
    shapes = seeding.generator_tensor_shapes(1024)
    assert len(shapes) == 171  # 171 tensors in a 1024^2 g_ema checkpoint
    n_params = sum(int(np.prod(s)) for k, s in shapes.items() if not (k.startswith("noises.") or k.endswith("kernel")))
    assert n_params == 30370060
    assert seeding.noise_sizes(1024) == [4] + [s for r in range(3, 11) for s in (2 ** r, 2 ** r)]


def test_gaussian_filter_cases(golden):
    g = golden("audioreactive_torch.npz")
    for name in g["gf.cases"]:
        sigma, causal, smf, kind = g[f"gf.{name}.cfg"]
        c = None if kind == 0 else (float(causal) if kind == 2 else int(causal))
        y = signal_oracle.gaussian_filter(t(g[f"gf.{name}.x"]), float(sigma) if sigma != int(sigma) else int(sigma), causal=c,
                                          smf=float(smf))
        np.testing.assert_allclose(y.numpy(), g[f"gf.{name}.y"], atol=1e-5, err_msg=str(name))


def test_percentile_clip_normalize_compress(golden):
    g = golden("audioreactive_torch.npz")
    for name in g["pc.cases"]:
        y = signal_oracle.percentile_clip(t(g[f"pc.{name}.x"]).clone(), int(g[f"pc.{name}.p"]))
        np.testing.assert_allclose(y.numpy(), g[f"pc.{name}.y"], atol=1e-6)
    np.testing.assert_allclose(signal_oracle.normalize(t(g["normalize.x"])).numpy(), g["normalize.y"], atol=1e-6)
    np.testing.assert_allclose(signal_oracle.compress(t(g["compress.x"]), 0.5, 0.25).numpy(), g["compress.y"], atol=1e-6)


def test_chroma_weight_latents_and_noise_range(golden):
    g = golden("audioreactive_torch.npz")
    y = signal_oracle.chroma_weight_latents(t(g["cwl.chroma"]), t(g["cwl.latents"]))
    np.testing.assert_allclose(y.numpy(), g["cwl.y"], atol=1e-5)
    for row in g["noise_range"]:
        out_size, g_res = int(row[0]), int(row[1])
        sides = [int(v) for v in row[4:] if v]
        assert signal_oracle.noise_side_lengths(out_size, g_res) == sides


def test_postprocess_uint8(golden):
    g = golden("postprocess.npz")
    assert (so.frames_to_uint8(t(g["x"])) == g["y"]).all()


def test_oracle_audio_features_are_sane():
    """Unpinned stages: structural checks only (documented 'parity unpinned')."""
    sr = 22050
    y = seeding.synthetic_audio(4.0, sr)
    p = signal_oracle.stft_power(y)
    assert p.shape == (1025, 1 + len(y) // 512)
    fb = signal_oracle.mel_filterbank(sr, fmin=20, fmax=8000)
    assert fb.shape == (128, 1025) and (fb >= 0).all() and fb.sum() > 0
    env = signal_oracle.onsets(y, sr, 120, fmax=150, smooth=5, clip=97, power=2)
    assert env.shape == (120,) and float(env.max()) <= 1.0 + 1e-6 and float(env.min()) >= 0.0
    # 120 BPM kick -> onset envelope periodic with 15 frames at 30 fps
    e = env.numpy() - env.numpy().mean()
    ac = np.correlate(e, e, "full")[len(e) - 1:]
    assert 13 <= int(np.argmax(ac[8:40])) + 8 <= 17
    ch = signal_oracle.chroma(y, sr, 120)
    assert ch.shape == (120, 12) and np.allclose(ch.sum(1).numpy(), 1.0, atol=1e-5)
    # a pure A4 sine lands in pitch class A (index 9 with C-based chroma)
    tone = np.sin(2 * np.pi * 440.0 * np.arange(sr) / sr).astype(np.float32)
    assert int(np.argmax(signal_oracle.chroma_stft(tone, sr).mean(axis=1))) == 9


def test_c_restatement_matches_golden():
    """oracle/c/ops_ref.c (scalar loops, independent of the conv2d-based oracle) against the reference's outputs."""
    import os
    import subprocess

    from c_oracle_check import check_c_oracle
    from conftest import REPO

    subprocess.run(["make", "-C", os.path.join(REPO, "oracle")], check=True, capture_output=True)
    check_c_oracle(os.path.join(REPO, "oracle", "c", "libops_ref.so"))


def test_c_restatement_under_asan_and_ubsan():
    """SURVEY.md 5 row 2 (sanitizers), VERDICT r4 item 6: the same golden check with oracle/c/ops_ref.c built with
    -fsanitize=address,undefined -fno-sanitize-recover=all (`make -C oracle san`), in a subprocess that pre-loads gcc's ASAN runtime: any
    out-of-bounds access, signed overflow, misaligned or invalid shift in the scalar restatement aborts the run."""
    import os
    import subprocess
    import sys

    from conftest import REPO

    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "san"], check=True, capture_output=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan):
        pytest.skip("gcc has no shared ASAN runtime here")
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "c_oracle_check.py"), os.path.join(REPO, "oracle", "c", "libops_ref_san.so")],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "c oracle ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


# ---------------------------------------------------------------------------------------------------------------------
# Independent cross-checks of the "parity unpinned" audio restatements against third-party implementations that ARE
# installed here (scipy, scikit-learn).  librosa itself is absent; these pin the building blocks it is made of.
def test_oracle_stft_matches_scipy_stft():
    """oracle.stft_complex (centred, reflect-padded, periodic Hann — librosa.stft's defaults) against scipy.signal.stft
    on the same padded signal (scipy scales by 1 / sum(window))."""
    import scipy.signal

    sr = 22050
    y = seeding.synthetic_audio(1.5, sr).astype(np.float64)
    for n_fft, hop in [(2048, 512), (2048, 441), (1024, 256)]:
        want_frames = 1 + len(y) // hop
        ypad = np.pad(y, n_fft // 2, mode="reflect")
        _, _, z = scipy.signal.stft(ypad, window="hann", nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=None,
                                    padded=False)
        z = z * signal_oracle.hann_periodic(n_fft).sum()
        got = signal_oracle.stft_complex(y, n_fft, hop)
        assert got.shape == (n_fft // 2 + 1, want_frames) and z.shape[1] >= want_frames
        np.testing.assert_allclose(got, z[:, :want_frames], atol=1e-9)
        np.testing.assert_allclose(signal_oracle.stft_power(y, n_fft, hop), np.abs(z[:, :want_frames]) ** 2, atol=1e-8)


def test_oracle_istft_inverts_stft():
    y = seeding.synthetic_audio(1.0).astype(np.float64)
    back = signal_oracle.istft(signal_oracle.stft_complex(y), len(y))
    np.testing.assert_allclose(back, y, atol=1e-10)


def test_oracle_nn_filter_matches_sklearn_neighbours():
    """The neighbour selection of nn_filter_median against scikit-learn's cosine NearestNeighbors (the engine librosa's
    recurrence matrix is built on): k + 2 width candidates, the |i - j| < width band removed, the k nearest kept."""
    from sklearn.neighbors import NearestNeighbors

    rng = np.random.default_rng(5)
    for f, t, width in [(12, 60, 1), (12, 200, 1), (7, 90, 3)]:
        ch = rng.random((f, t)) + 0.05
        k = int(min(t - 1, 2 * np.ceil(np.sqrt(t - 2 * width + 1))))
        knn = NearestNeighbors(n_neighbors=min(t, k + 2 * width), metric="cosine", algorithm="brute").fit(ch.T)
        _, idx = knn.kneighbors(ch.T)
        want = np.empty_like(ch)
        for i in range(t):
            nb = [j for j in idx[i] if abs(j - i) >= width][:k]
            assert len(nb) == k
            want[:, i] = np.median(ch[:, nb], axis=1)
        np.testing.assert_allclose(signal_oracle.nn_filter_median(ch, width), want, atol=1e-12)


def test_oracle_band_onset_envelope_peaks_on_the_beats():
    """Property of the type="mm" onset envelope (50 frames per second): impulses every half second put its largest values
    on frames 25, 50, 75, ... and nothing comparable in between."""
    sr = 22050
    y = np.zeros(4 * sr)
    rng = np.random.default_rng(0)
    for beat in range(1, 8):
        start = beat * sr // 2
        y[start:start + 400] += rng.standard_normal(400) * np.exp(-np.arange(400) / 80.0)
    env = signal_oracle.madmom_like_onset_strength(y, sr)
    assert env.shape == (1 + len(y) // 441,)
    # frame t is centred on sample 441 t with a 2048-sample window: each burst lights up frames 25 beat - 1 and 25 beat
    beats = [25 * b for b in range(1, 8)]
    for t in beats:
        assert int(np.argmax(env[t - 6:t + 7])) + t - 6 in (t - 1, t), (t, env[t - 6:t + 7])
    off_beat = np.delete(env, np.concatenate([np.arange(t - 3, t + 4) for t in beats]))
    assert off_beat.max() < 0.01 * min(env[t - 1:t + 1].max() for t in beats)


def test_oracle_affine_warp_matches_torch_grid_sample():
    """affine_reflect_warp (the bend transform of audioreactive/bend.py:60-102: ReflectionPad2d -> kornia affine warp ->
    CenterCrop) against torch's own ReflectionPad2d + grid_sample(bilinear, zeros, align_corners=True) — the engine
    kornia's warp_affine calls — for rotations, anisotropic scales and translations that leave the canvas."""
    rng = np.random.default_rng(11)
    b, c, h, w = 3, 2, 9, 13
    x = rng.standard_normal((b, c, h, w))
    for pads in [(4, 4, 4, 4), (3, 5, 2, 6), (0, 0, 0, 0)]:
        pl, pr, pt, pb = pads
        ch_, cw_ = h + pt + pb, w + pl + pr
        maps = []
        for i in range(b):
            ang, sx, sy = rng.uniform(-1, 1), rng.uniform(0.6, 1.6), rng.uniform(0.6, 1.6)
            cx0, cy0 = (cw_ - 1) / 2, (ch_ - 1) / 2
            a = np.array([[np.cos(ang) * sx, -np.sin(ang) * sy], [np.sin(ang) * sx, np.cos(ang) * sy]])
            t = np.array([cx0, cy0]) - a @ np.array([cx0, cy0]) + rng.uniform(-4, 4, 2)
            maps.append([a[0, 0], a[0, 1], t[0], a[1, 0], a[1, 1], t[1]])
        maps = np.array(maps)
        noise = rng.standard_normal((ch_, cw_)) if pads[0] == 3 else None
        got = signal_oracle.affine_reflect_warp(x, maps, pads, noise)

        canvas = torch.nn.ReflectionPad2d(pads)(torch.from_numpy(x)) if any(pads) else torch.from_numpy(x)
        if noise is not None:
            canvas = canvas + torch.from_numpy(noise)[None, None]
        oy, ox = np.meshgrid(np.arange(h) + (ch_ - h) // 2, np.arange(w) + (cw_ - w) // 2, indexing="ij")
        grid = np.empty((b, h, w, 2))
        for i in range(b):
            m = maps[i]
            grid[i, ..., 0] = 2 * (m[0] * ox + m[1] * oy + m[2]) / (cw_ - 1) - 1
            grid[i, ..., 1] = 2 * (m[3] * ox + m[4] * oy + m[5]) / (ch_ - 1) - 1
        want = torch.nn.functional.grid_sample(canvas, torch.from_numpy(grid), mode="bilinear", padding_mode="zeros",
                                               align_corners=True).numpy()
        np.testing.assert_allclose(got, want, atol=1e-10)


def test_oracle_stacked_reflection_pads_match_torch():
    """Translate (audioreactive/bend.py:60-64) stacks THREE ReflectionPad2d modules; each mirrors the canvas built so far, so
    the result is not one reflection fold of the source.  The oracle's pad chain and the host's index tables
    (bend.reflection_chain_index, what the HIP warp kernel reads) against really stacked torch.nn.ReflectionPad2d."""
    from maua_stylegan2_amd.audioreactive.bend import reflection_chain_index

    rng = np.random.default_rng(12)
    for h, w in [(6, 8), (5, 9), (4, 16)]:
        x = rng.standard_normal((2, 3, h, w))
        chain = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]
        canvas = torch.from_numpy(x)
        for p in chain:
            canvas = torch.nn.ReflectionPad2d(p)(canvas)
        cw = canvas.shape[-1]
        assert cw == w + 2 * int(w / 2) + 3 * w
        cols = reflection_chain_index(w, [(p[0], p[1]) for p in chain])
        np.testing.assert_array_equal(canvas.numpy(), x[..., cols])
        # identity map + centre crop through the oracle == the centre columns of the stacked canvas; a shift of w columns
        # lands on the same features (the seamless scroll of examples/tauceti.py:142-144) when w is even
        ident = np.tile(np.array([1.0, 0, 0, 0, 1.0, 0]), (2, 1))
        got = signal_oracle.affine_reflect_warp(x, ident, chain)
        lo = (cw - w) // 2
        np.testing.assert_allclose(got, canvas.numpy()[..., lo: lo + w], atol=1e-12)
        shifted = ident.copy()
        shifted[:, 2] = -w
        if w % 2 == 0:
            np.testing.assert_allclose(signal_oracle.affine_reflect_warp(x, shifted, chain), got, atol=1e-12)
        # a single fold of the total padding is NOT the same canvas (what round 1 implemented)
        single = np.pad(x, ((0, 0), (0, 0), (0, 0), (int(w / 2) + 2 * w, int(w / 2) + w)), mode="reflect")
        assert np.abs(single - canvas.numpy()).max() > 0.1
    # mixed vertical / horizontal chain
    x = rng.standard_normal((1, 2, 7, 6))
    chain = [(2, 1, 3, 0), (4, 5, 2, 6)]
    canvas = torch.nn.ReflectionPad2d(chain[1])(torch.nn.ReflectionPad2d(chain[0])(torch.from_numpy(x))).numpy()
    rows = reflection_chain_index(7, [(p[2], p[3]) for p in chain])
    cols = reflection_chain_index(6, [(p[0], p[1]) for p in chain])
    np.testing.assert_array_equal(canvas, x[..., rows, :][..., cols])


def test_perlin_oracle_matches_reference_golden(golden):
    """perlin.npz holds outputs of the reference's own perlin_noise (tests/golden/make_golden.py runs it on the CPU by
    neutralising its hard-coded .cuda() calls) together with the gradient angles it drew."""
    g = golden("perlin.npz")
    for name in g["cases"]:
        cfg = g[f"{name}.cfg"]
        shape, res, tileable = tuple(cfg[:3]), tuple(cfg[3:6]), tuple(bool(v) for v in cfg[6:9])
        got = signal_oracle.perlin_noise(shape, res, g[f"{name}.theta"].copy(), g[f"{name}.phi"].copy(), tileable)
        np.testing.assert_allclose(got, g[f"{name}.y"], atol=1e-12, err_msg=str(name))


def test_oracle_frames_match_the_reference_render_loop(golden):
    """render_512.npz holds frames the reference's own render() produced on the CPU (tests/golden/make_golden.py captures
    them from its ffmpeg pipe): 5 frames of a seeded 512^2 generator, batch 2, (a) checkpoint noise buffers and float
    truncation 1.0, (b) per-frame noise up to 64 px and a per-frame truncation tensor.  The oracle's generator + uint8
    post-process reproduce the stored pixel subsample to within one grey level."""
    g = golden("render_512.npz")
    size, n, _, s_w, s_l, s_n = [int(v) for v in g["cfg"]]
    sd = seeding.seeded_state_dict(size, seed=s_w)
    n_latent = 2 * int(np.log2(size)) - 2
    lat = seeding.seeded_latents(n, n_latent, seed=s_l)
    per_frame = seeding.seeded_noise(n, size, seed=s_n)
    tl = torch.from_numpy(seeding.seeded_array(5, "truncation_latent", (1, 512)))
    for tag, noise, trunc in (("a", [None] * len(per_frame), None),
                              ("b", [nz if nz.shape[-1] <= 64 else None for nz in per_frame], torch.from_numpy(g["b.truncation"]))):
        frames = np.asarray(so.frames_to_uint8(so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl)))
        assert frames.shape == (n, size, size, 3)
        diff = np.abs(frames[:, 3::8, 5::8, :].astype(np.int16) - g[f"{tag}.sub"].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, tag
        assert np.abs(frames.reshape(n, -1).sum(1).astype(np.int64) - g[f"{tag}.sums"]).max() < 1e-4 * size * size * 3, tag


def test_oracle_z_inputs_match_reference_golden(golden):
    """mapping.npz: the reference's mapping network and its one- / two-z forwards (style mixing at inject_index 3) of a
    seeded 32^2 generator (models/stylegan2.py:388-393,511-526)."""
    fx = golden("mapping.npz")
    sd = seeding.seeded_state_dict(32, seed=3)
    z = torch.from_numpy(seeding.seeded_array(4, "z", (5, 512)))
    np.testing.assert_allclose(so.mapping_network(sd, z).numpy(), fx["w"], atol=1e-6)
    z1 = torch.from_numpy(seeding.seeded_array(4, "z1", (2, 512)))
    z2 = torch.from_numpy(seeding.seeded_array(4, "z2", (2, 512)))
    noise = seeding.seeded_noise(2, 32, seed=9)
    for tag, zs, idx in (("one", [z1], None), ("mix", [z1, z2], int(fx["inject_index"]))):
        lat = so.latents_from_z(sd, zs, 8, idx)
        np.testing.assert_allclose(lat.numpy(), fx[f"{tag}.latents"], atol=1e-6)
        np.testing.assert_allclose(so.generator_forward(sd, lat, noise).numpy(), fx[f"{tag}.image"], atol=1e-5)


class _FlipX(torch.nn.Module):
    def forward(self, x):
        return x.flip(-1)


def test_oracle_bend_placement_matches_reference_golden(golden):
    """bends.npz: the reference generator with transform_dict_list entries at layer ids 0 (ReplicationPad2d widening the
    constant: the 2:1 output route), 1, 3 and 7 — pins where the oracle applies bends and its wide-noise handling."""
    fx = golden("bends.npz")
    s_w, s_l, s_n = [int(v) for v in fx["seeds"]]
    size, batch = 32, 2
    sd = seeding.seeded_state_dict(size, seed=s_w)
    lat = seeding.seeded_latents(batch, 8, seed=s_l)
    noise = [torch.from_numpy(seeding.seeded_array(s_n, f"wn{i}", (batch, 1, r, 2 * r))) for i, r in enumerate(seeding.noise_sizes(size))]
    bends = {0: torch.nn.ReplicationPad2d((2, 2, 0, 0)), 1: _FlipX(), 3: lambda t: t * 0.5, 7: lambda t: t * -1.25}
    assert sorted(bends) == list(fx["layers"])
    got = so.generator_forward(sd, lat, noise, bends=bends)
    assert tuple(got.shape) == (batch, 3, size, 2 * size)
    np.testing.assert_allclose(got.numpy(), fx["image"], atol=1e-5)
    moved = so.generator_forward(sd, lat, noise, bends={0: bends[0], 2: bends[1], 3: bends[3], 7: bends[7]})
    assert float((moved - got).abs().max()) > 1e-2  # the layer id matters


def test_oracle_generator_variants_match_reference_golden(golden):
    """LatentInput (--noconst) and min_rgb_size generators: the oracle against images of the reference classes."""
    from maua_stylegan2_amd import seeding
    from oracle import stylegan2_oracle as so

    fx = golden("generator_variants.npz")
    s_w, s_lat, s_noise, s_tl = (int(v) for v in fx["seeds"])
    lat = seeding.seeded_latents(2, 8, seed=s_lat)
    noise = seeding.seeded_noise(2, 32, seed=s_noise)
    trunc = torch.tensor([0.8, 1.0])
    tl = torch.from_numpy(seeding.seeded_array(s_tl, "truncation_latent", (1, 512)))
    sd = seeding.seeded_state_dict(32, seed=s_w, constant_input=False)
    got = so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl)
    np.testing.assert_allclose(got.numpy(), fx["noconst.image"], atol=1e-4)
    sd = seeding.seeded_state_dict(32, seed=s_w)
    got = so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl, min_rgb_size=16)
    np.testing.assert_allclose(got.numpy(), fx["min_rgb16.image"], atol=1e-4)
    assert np.abs(so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl).numpy()
                  - fx["min_rgb16.image"]).max() > 1e-2  # the skipped low-resolution ToRGBs matter


def test_plugin_stand_ins_are_deterministic():
    """tests/golden/plugin_stubs.py regenerates the inputs the reference was run on (make_golden.py --only-plugin): the
    envelopes and the shape-keyed randn must not depend on call order or on the process."""
    import sys

    from conftest import GOLDEN

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    import plugin_stubs as stubs

    lo, hi, ch = stubs.envelopes(600)
    lo2, hi2, ch2 = stubs.envelopes(600)
    assert np.array_equal(lo, lo2) and np.array_equal(ch, ch2) and 0 <= lo.min() and hi.max() <= 1
    np.testing.assert_allclose(ch.sum(1), 1.0, atol=1e-6)
    r1, r2 = stubs.SeededRandn(7), stubs.SeededRandn(7)
    a = r1(4, 1, 8, 8)
    r2(3, 3)  # an unrelated draw in between changes nothing
    assert torch.equal(a, r2((4, 1, 8, 8))) and not torch.equal(a, r1(4, 1, 8, 8))
    assert lo.std() > 0.05 and hi.std() > 0.05  # peaky envelopes, not constants


def _sg1_state_dict(fx):
    """The seeded narrow G_synthesis weights of tests/golden/stylegan1.npz, regenerated from (key, shape) + the seeding rules of
    make_golden.stylegan1_fixture."""
    from maua_stylegan2_amd import seeding

    seed = int(fx["synth.seeds"][0])
    sd = {}
    for key, shape in zip(fx["synth.keys"], fx["synth.shapes"]):
        key, shape = str(key), tuple(int(v) for v in str(shape).split(";"))
        if key.endswith("intermediate.kernel"):
            k = np.outer([1.0, 2.0, 1.0], [1.0, 2.0, 1.0]).astype(np.float32) / 16.0
            sd[key] = torch.from_numpy(k[None, None])
        elif key.endswith("noise.weight"):
            sd[key] = torch.from_numpy(seeding.seeded_array(seed, key, shape, std=0.3))
        elif key.endswith(".bias"):
            sd[key] = torch.from_numpy(seeding.seeded_array(seed, key, shape, std=0.2))
        else:
            sd[key] = torch.from_numpy(seeding.seeded_array(seed, key, shape))
    return sd


def test_stylegan1_oracle_matches_reference_golden(golden):
    """oracle/stylegan1_oracle.py against outputs of the reference's G_synthesis / G_mapping classes (stylegan1.npz), and the
    identity the device path is built on: the reference's fused conv_transpose2d upscale (>= 128 px, models/stylegan1.py:83-93)
    is nearest-neighbour upscaling followed by the SAME 3x3 kernel flipped."""
    from maua_stylegan2_amd import seeding
    from oracle import stylegan1_oracle as s1o

    fx = golden("stylegan1.npz")
    sd = _sg1_state_dict(fx)
    s_w, s_l, s_n = (int(v) for v in fx["synth.seeds"])
    n_blocks = 7
    dl = torch.from_numpy(seeding.seeded_array(s_l, "dlatents", (2, 2 * n_blocks, 512)))
    noise = [torch.from_numpy(seeding.seeded_array(s_n, f"noise_{i}", (2 if i % 2 else 1, 1, 4 * 2 ** i, 4 * 2 ** i))) for i in range(n_blocks)]
    np.testing.assert_allclose(s1o.synthesis(sd, dl, noise, prefix="").numpy(), fx["synth.image"], atol=2e-4)
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.standard_normal((2, 5, 64, 72)).astype(np.float32))
    w = torch.from_numpy(rng.standard_normal((7, 5, 3, 3)).astype(np.float32))
    fused = s1o.conv_layer(x, w, None, upscale=True)  # 64 * 2 >= 128: the conv_transpose2d branch
    w_mul = np.sqrt(2) * (5 * 9) ** -0.5
    plain = torch.nn.functional.conv2d(s1o.upscale2d(x), w.flip(-1, -2) * w_mul, padding=1)
    np.testing.assert_allclose(fused.numpy(), plain.numpy(), atol=2e-5)


def test_stylegan1_mirror_has_the_reference_module_tree(built_lib):
    """Key list and shapes of the mirror's G_synthesis (narrow variant of the fixture) and the constructor bookkeeping of G_style
    for a 128-px checkpoint and 1920-wide output (stylegan1_meta.json, read from the reference's own G_style): constant widened
    32 -> 36 columns, one noise buffer per block doubling in size, state-dict keys."""
    import json
    import os

    from conftest import GOLDEN
    from maua_stylegan2_amd.models import stylegan1 as sg1

    fx = np.load(os.path.join(GOLDEN, "stylegan1.npz"))
    gs = sg1.G_synthesis(resolution=256, fmap_base=512, fmap_max=64)
    got = {k: ";".join(str(d) for d in v.shape) for k, v in gs.state_dict().items()}
    assert list(got) == [str(k) for k in fx["synth.keys"]] and list(got.values()) == [str(v) for v in fx["synth.shapes"]]
    gs.load_state_dict(_sg1_state_dict(fx), strict=True)
    meta = json.load(open(os.path.join(GOLDEN, "stylegan1_meta.json")))
    import tempfile

    tmp = tempfile.mkdtemp(prefix="maua_sg1_")
    small = sg1.G_style.__new__(sg1.G_style)
    torch.nn.Sequential.__init__(small)
    small.g_mapping = sg1.G_mapping()
    small.g_synthesis = sg1.G_synthesis(resolution=128)
    torch.save(small.state_dict(), os.path.join(tmp, "sg1_128.pt"))
    g = sg1.G_style(output_size=1920, checkpoint=os.path.join(tmp, "sg1_128.pt"))
    assert list(getattr(g.g_synthesis.blocks, "4x4").const.shape) == meta["const"]
    assert list(g.g_synthesis.blocks.keys()) == meta["blocks"]
    assert [list(getattr(g, f"noise_{i}").shape) for i in range(len(meta["noise"]))] == meta["noise"]
    assert list(g.truncation_latent.shape) == meta["truncation_latent"] and list(g.state_dict().keys()) == meta["keys"]


def test_plugin_oracle_matches_reference_plugin(golden):
    """oracle/plugin_oracle.py (the CPU restatement of the reference's default plugin, used as the checker of the whole-workload GPU
    test) against what the reference's own callbacks produced on the stand-in features (default_plugin.npz)."""
    import sys

    from conftest import GOLDEN

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    import plugin_stubs as stubs

    from maua_stylegan2_amd import seeding
    from oracle import plugin_oracle

    fx = golden("default_plugin.npz")
    n = int(fx["n_frames"])
    lo, hi, ch = (torch.from_numpy(a) for a in stubs.envelopes(n))
    selection = torch.from_numpy(seeding.seeded_array(42, "selection", (12, 16, 512)))
    lat = plugin_oracle.get_latents(selection, ch, lo, hi)
    got = stubs.summary(lat)
    for key in ("stats", "frame_mean", "sub"):
        np.testing.assert_allclose(got[key], fx[f"latents.{key}"], atol=1e-5, err_msg=key)
    randn = stubs.SeededRandn(41)
    for h, w in [tuple(int(v) for v in hw) for hw in fx["noise_sizes"]][:4] + [(512, 512)]:  # (the larger maps: GPU test only)
        nz = plugin_oracle.get_noise(h, w, n, lo, hi, randn)
        if f"noise_{h}x{w}.none" in fx.files:
            assert nz is None
            continue
        got = stubs.summary(nz)
        for key in ("stats", "frame_mean", "sub"):
            np.testing.assert_allclose(got[key], fx[f"noise_{h}x{w}.{key}"], atol=2e-5, err_msg=f"{h}x{w} {key}")
