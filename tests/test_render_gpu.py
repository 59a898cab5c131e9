"""GPU: the render loop / generate() surface end to end on small generators (raw rgb24 sink, no ffmpeg needed)."""
import os

import numpy as np
import pytest
import torch

from maua_stylegan2_amd import seeding

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)


def build(size, dev, seed=0):
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(size, seed=seed), strict=True)
    return g.to(dev).eval()


def build_unsaturated(size, dev, seed, latents, noise, target_std=0.35):
    """Seeded generator whose frames can SHOW an error: the plain seeded checkpoint (ToRGB weights N(0,1)) gives images with a
    standard deviation of 1..3, i.e. 30..76 % of the uint8 frame sits on the clamp of render.py:40-43 where any error passes.  The
    image is linear in the ToRGB weights and biases (not demodulated, models/stylegan2.py:352-365), so the gain that brings the
    image's standard deviation to ``target_std`` follows from ONE forward at gain 1 (the product forward is used only to pick the
    test's input scale; the comparison itself is against the oracle on the resulting checkpoint).  Returns (state dict, generator)."""
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = build(size, dev, seed)
    noise2 = [None if nz is None else nz[:2].to(dev) for nz in noise]
    img, _ = g(styles=latents[:2].to(dev), noise=noise2, truncation=1.0, randomize_noise=False, input_is_latent=True)
    gain = float(target_std / float(img.std()))
    del g
    sd = seeding.seeded_state_dict(size, seed=seed, rgb_gain=gain)
    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(sd, strict=True)
    return sd, g.to(dev).eval()


def lane_of(g, batch, k, n_lanes):
    """The cached graph lane that rendered batch number ``k`` (render.synthesize replays its lanes round-robin)."""
    return g._graph_lanes[(batch, k % n_lanes, bool(g.tap_float_image))]


def test_synthesize_graph_equals_eager_and_oracle(gpu):
    """10 frames, batch 4 (2 graph batches + eager tail of 2): uint8 frames equal the eager path bit for bit and the
    oracle's frames to within one grey level."""
    from maua_stylegan2_amd import render
    from oracle import stylegan2_oracle as so

    size, n = 32, 10
    sd = seeding.seeded_state_dict(size, seed=4)
    g = build(size, gpu, 4)
    lat = seeding.seeded_latents(n, g.n_latent, seed=6)
    noise = seeding.seeded_noise(n, size, seed=7)
    noise[-1] = None  # checkpoint buffer for the last scale, like get_noise -> None
    trunc = torch.linspace(0.5, 1.0, n)
    g.truncation_latent = torch.from_numpy(seeding.seeded_array(5, "tl", (1, 512))).to(gpu)

    def run(use_graph):
        frames = np.zeros((n, size, size, 3), np.uint8)
        for first, u8 in render.synthesize(g, lat, noise, 4, truncation=trunc, use_graph=use_graph):
            frames[first: first + u8.shape[0]] = u8.cpu().numpy()
        return frames

    graphed, eager = run(True), run(False)
    assert np.array_equal(graphed, eager)
    want = so.frames_to_uint8(so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=g.truncation_latent.cpu()))
    diff = np.abs(graphed.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3


def test_float_truncation_reaches_the_captured_graph(gpu):
    """A float truncation != 1 (CLI ``--truncation 0.7``) must truncate the hipGraph batches exactly like the eager tail
    batch and like the oracle (reference models/stylegan2.py:537-543 lerps every batch): round 1 captured the graph without
    the lerp, so full batches came out untruncated and the ragged tail truncated."""
    from maua_stylegan2_amd import render
    from oracle import stylegan2_oracle as so

    size, n = 32, 10
    sd = seeding.seeded_state_dict(size, seed=4)
    g = build(size, gpu, 4)
    lat = seeding.seeded_latents(n, g.n_latent, seed=6)
    noise = seeding.seeded_noise(n, size, seed=7)
    tl = torch.from_numpy(seeding.seeded_array(5, "tl", (1, 512)))
    g.truncation_latent = tl.to(gpu)

    def run(truncation, use_graph):
        frames = np.zeros((n, size, size, 3), np.uint8)
        for first, u8 in render.synthesize(g, lat, noise, 4, truncation=truncation, use_graph=use_graph):
            frames[first: first + u8.shape[0]] = u8.cpu().numpy()
        return frames

    graphed, eager = run(0.7, True), run(0.7, False)
    assert np.array_equal(graphed, eager)
    want = so.frames_to_uint8(so.generator_forward(sd, lat, noise, truncation=torch.full((n,), 0.7), truncation_latent=tl))
    diff = np.abs(graphed.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3
    untruncated = run(torch.ones(n), True)
    assert np.abs(graphed.astype(np.int16) - untruncated.astype(np.int16)).mean() > 1.0  # 0.7 is really applied


def test_render_matches_reference_render_loop(gpu, tmp_path, monkeypatch, golden):
    """render() against frames produced by the REFERENCE's own render loop (tests/golden/render_512.npz, captured from its
    ffmpeg pipe on the CPU): seeded 512^2 generator, 5 frames, batch 2 — (a) checkpoint noise buffers, float truncation;
    (b) per-frame noise up to 64 px, per-frame truncation tensor.  Compared on the stored pixel subsample, <= 1 grey level."""
    from maua_stylegan2_amd import render

    monkeypatch.setattr(render.shutil, "which", lambda name: None)  # raw rgb24 sink
    fx = golden("render_512.npz")
    size, n, batch, s_w, s_l, s_n = [int(v) for v in fx["cfg"]]
    g = build(size, gpu, s_w)
    lat = seeding.seeded_latents(n, g.n_latent, seed=s_l)
    per_frame = seeding.seeded_noise(n, size, seed=s_n)
    g.truncation_latent = torch.from_numpy(seeding.seeded_array(5, "truncation_latent", (1, 512))).to(gpu)
    scenarios = (("a", [None] * g.num_layers, 1.0),
                 ("b", [nz if nz.shape[-1] <= 64 else None for nz in per_frame], torch.from_numpy(fx["b.truncation"])))
    for tag, noise, truncation in scenarios:
        out = str(tmp_path / f"clip_{tag}.mp4")
        written = render.render(generator=g, latents=lat.clone(), noise=list(noise), offset=0, duration=n / 30,
                                batch_size=batch, out_size=size, output_file=out, truncation=truncation)
        assert written == n
        raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n, size, size, 3)
        diff = np.abs(raw[:, 3::8, 5::8, :].astype(np.int16) - fx[f"{tag}.sub"].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (tag, int(diff.max()), float((diff > 0).mean()))


def test_render_pipes_rawvideo_to_an_ffmpeg_process(gpu, tmp_path, monkeypatch):
    """The encoder branch of the frame sink (reference render.py:58-113: rawvideo rgb24 on the stdin of an ``ffmpeg`` child
    process, libx264 / yuv420p / preset, audio mux arguments).  No ffmpeg binary exists in this image, so a stand-in
    executable named ``ffmpeg`` is put on PATH: it records its argument vector and copies stdin to the output path.  Checked:
    the arguments the reference passes, every frame's bytes in order, the child is waited for."""
    import json
    import stat

    from maua_stylegan2_amd import render

    bindir = tmp_path / "bin"
    bindir.mkdir()
    fake = bindir / "ffmpeg"
    fake.write_text("#!/usr/bin/env python3\nimport json, sys\nargs = sys.argv[1:]\nout = args[-1]\n"
                    "json.dump(args, open(out + '.args.json', 'w'))\n"
                    "data = sys.stdin.buffer.read()\nopen(out, 'wb').write(data)\n")
    fake.chmod(fake.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", f"{bindir}:{os.environ['PATH']}")
    g = build(512, gpu, 1)
    n = 5
    lat = seeding.seeded_latents(n, g.n_latent, seed=2)
    out = str(tmp_path / "clip.mp4")
    written = render.render(generator=g, latents=lat, noise=[None] * g.num_layers, offset=1.5, duration=n / 24, batch_size=2,
                            out_size=512, output_file=out, audio_file="song.wav", ffmpeg_preset="veryfast")
    assert written == n
    args = json.load(open(out + ".args.json"))
    joined = " ".join(args)
    for expect in ("-f rawvideo", "-pix_fmt rgb24", "-s 512x512", "-i pipe:", f"-framerate {n / (n / 24)}", "-ss 1.5", f"-t {n / 24}",
                   "-i song.wav", "-vcodec libx264", "-pix_fmt yuv420p", "-preset veryfast", "-b:a 320K", "-ac 2"):
        assert expect in joined, (expect, joined)
    raw = np.fromfile(out, dtype=np.uint8).reshape(n, 512, 512, 3)
    for i in (0, n - 1):
        img, _ = g(styles=lat[i: i + 1].to(gpu), noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
        diff = np.abs(raw[i].astype(np.int16) - render.frames_to_uint8(img).cpu().numpy()[0].astype(np.int16))
        assert diff.max() <= 1


def test_render_writes_ordered_frames(gpu, tmp_path, monkeypatch):
    from maua_stylegan2_amd import render

    monkeypatch.setattr(render.shutil, "which", lambda name: None)  # force the raw sink even if ffmpeg exists
    g = build(512, gpu, 1)  # smallest size render() accepts (reference render.py:47-56)
    n = 5
    lat = seeding.seeded_latents(n, g.n_latent, seed=2)
    noise = [None] * g.num_layers
    out = str(tmp_path / "clip.mp4")
    written = render.render(generator=g, latents=lat, noise=noise, offset=0, duration=n / 30, batch_size=2, out_size=512,
                            output_file=out)
    assert written == n
    raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n, 512, 512, 3)
    for i in range(n):
        img, _ = g(styles=lat[i: i + 1].to(gpu), noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
        want = render.frames_to_uint8(img).cpu().numpy()[0]
        # batch-2 graph vs batch-1 eager: split-K depth depends on the batch, so sums may differ in the last ulp
        diff = np.abs(raw[i].astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, i
    with pytest.raises(Exception, match="output sizes"):
        render.render(g, lat, noise, 0, 1.0, 2, 300, None)


def test_generate_end_to_end_default_plugin(gpu, tmp_path, monkeypatch):
    """generate() with the default plugin semantics (onsets + chroma latents + reactive noise) on a 4 s synthetic WAV,
    random-init 512^2... kept small: G_res 512 is the smallest the reference's render() accepts."""
    import scipy.io.wavfile

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive.examples import default as plugin

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(render.shutil, "which", lambda name: None)
    sr = 22050
    y = seeding.synthetic_audio(2.0, sr)
    scipy.io.wavfile.write("track.wav", sr, (y * 32767).astype(np.int16))
    np.save("lat.npy", seeding.seeded_latents(12, 16, seed=3).numpy())
    out = gav.generate(ckpt=None, audio_file="track.wav", initialize=plugin.initialize, get_latents=plugin.get_latents,
                       get_noise=plugin.get_noise, latent_file="lat.npy", G_res=512, out_size=512, fps=12, batch=4,
                       output_file=str(tmp_path / "o.mp4"))
    raw = np.fromfile(out + ".rgb24", dtype=np.uint8)
    assert raw.size == 24 * 512 * 512 * 3
    frames = raw.reshape(24, 512, 512, 3)
    assert frames.std() > 5 and not np.array_equal(frames[0], frames[12])
    assert os.path.exists("workspace/last-latents.npy")


@pytest.mark.parametrize("size,all2d", [(1024, False), (512, True), (64, False)])
def test_frame_epilogue_fused_into_last_torgb(gpu, size, all2d):
    """capture_graph(frames_u8=True): the uint8 NHWC frames written by the last layer's fused ToRGB epilogue (1024^2: F(4,3)
    kernel; 512^2 with the 2-D Winograd kernel forced; 64^2: 512-channel last layer, not fusable -> maua_frames_to_u8 behind
    it) equal render.py:40-43 applied to the eager fp32 image, bit for bit; the fp32 image is not produced in the fused case."""
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    keep = ModulatedConv2d.winograd2d_min_cout
    try:
        if all2d:
            ModulatedConv2d.winograd2d_min_cout = 32
        g = build(size, gpu, 3)
        b = 2
        lat = seeding.seeded_latents(b, g.n_latent, seed=8).to(gpu)
        img, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
        want = render.frames_to_uint8(img)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            lane = g.capture_graph(b, frames_u8=True)
            lane.bind(lat, [None] * g.num_layers)
            lane.u8.fill_(7)
            lane.replay(0)
            stream.synchronize()
        assert lane.u8.shape == (b, size, size, 3) and lane.u8.dtype == torch.uint8
        assert torch.equal(lane.u8, want)
        assert (lane.image is None) == (size >= 512)  # fused: no fp32 image; fallback: image + conversion kernel
    finally:
        ModulatedConv2d.winograd2d_min_cout = keep


def _stubs():
    import sys

    from conftest import GOLDEN

    if GOLDEN not in sys.path:
        sys.path.insert(0, GOLDEN)
    import plugin_stubs

    return plugin_stubs


def _assert_summary(got, fx, prefix, atol):
    for key in ("stats", "frame_mean", "sub"):
        np.testing.assert_allclose(got[key], fx[f"{prefix}.{key}"], atol=atol, rtol=0, err_msg=f"{prefix}.{key}")


def test_default_plugin_matches_reference_plugin(gpu, golden, monkeypatch):
    """The default plugin's callbacks against the REFERENCE's audioreactive/examples/default.py:6-45 run on the same
    stand-in features and the same seeded ``randn`` draws (tests/golden/default_plugin.npz, written by make_golden.py from
    the imported reference): 600 frames, latents [600,16,512] and reactive noise for 4..128 px (incl. a 2:1 map),
    ``None`` above 256 px.  Everything between the feature calls and the returned tensors is pinned: which onset bands are
    asked for, chroma weighting, both Gaussian filters (sigma 128 on the normal, non-shortened path), the onset
    cross-fades, the order of the two random fields and the std normalisation."""
    import argparse

    import maua_stylegan2_amd.audioreactive as ar
    from maua_stylegan2_amd.audioreactive.examples import default as plugin

    stubs = _stubs()
    fx = golden("default_plugin.npz")
    n_frames = int(fx["n_frames"])
    feats = stubs.Features(n_frames)
    monkeypatch.setattr(ar, "onsets", feats.onsets)
    monkeypatch.setattr(ar, "chroma", feats.chroma)
    monkeypatch.setattr(torch, "randn", stubs.SeededRandn(41))
    ar.set_SMF(1)
    args = plugin.initialize(argparse.Namespace(audio=np.zeros(8, np.float32), sr=22050, n_frames=n_frames, fps=30))
    np.testing.assert_array_equal(np.array([list(c[1:]) for c in feats.calls], dtype=np.float64), fx["onset_calls"])
    selection = torch.from_numpy(seeding.seeded_array(42, "selection", (12, 16, 512)))
    lat = plugin.get_latents(selection, args)
    assert tuple(lat.shape) == (n_frames, 16, 512)
    _assert_summary(stubs.summary(lat), fx, "latents", 2e-5)
    sizes = [tuple(int(v) for v in hw) for hw in fx["noise_sizes"]]
    for h, w in sizes:
        nz = plugin.get_noise(h, w, 0, len(sizes), args)
        if f"noise_{h}x{w}.none" in fx.files:
            assert nz is None
            continue
        assert tuple(nz.shape) == (n_frames, 1, h, w)
        _assert_summary(stubs.summary(nz), fx, f"noise_{h}x{w}", 2e-5)


@pytest.mark.parametrize("tag,truncation", [("a", 1.0), ("b", 0.7)])
def test_generate_matches_reference_generate(gpu, golden, tmp_path, monkeypatch, tag, truncation):
    """``generate()`` end to end against the REFERENCE's generate_audiovisual.generate (:59-231) + default plugin + render
    loop run on the CPU with the same stand-ins (tests/golden/generate_e2e.npz): seeded 512^2 checkpoint file, latent file,
    10 frames at batch 4 (two hipGraph batches + a ragged eager tail), (a) truncation 1.0, (b) float truncation 0.7 with the
    lazily drawn ``mean_latent(2**14)`` truncation latent.  Latents within 2e-5, noise statistics within 2e-5, delivered
    frames within one grey level on the stored pixel subsample."""
    import maua_stylegan2_amd.audioreactive as ar
    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive.examples import default as plugin

    stubs = _stubs()
    fx = golden("generate_e2e.npz")
    size, n, batch, fps, s_w, s_sel = [int(v) for v in fx["cfg"]]
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(render.shutil, "which", lambda name: None)  # raw rgb24 sink
    torch.save({"g_ema": seeding.seeded_state_dict(size, seed=s_w)}, "seeded512.pt")
    np.save("selection.npy", seeding.seeded_array(s_sel, "selection", (12, 16, 512)))
    feats = stubs.Features(n, fps)
    monkeypatch.setattr(ar, "onsets", feats.onsets)
    monkeypatch.setattr(ar, "chroma", feats.chroma)
    monkeypatch.setattr(ar, "load_audio", feats.load_audio)
    monkeypatch.setattr(torch, "randn", stubs.SeededRandn(44))
    seen = {}

    def get_latents(selection, args):
        seen["latents"] = plugin.get_latents(selection, args)
        return seen["latents"]

    def get_noise(height, width, scale, num_scales, args):
        nz = plugin.get_noise(height, width, scale, num_scales, args)
        seen.setdefault("noise", []).append(nz)
        return nz

    out = gav.generate(ckpt="seeded512.pt", audio_file="clip.wav", initialize=plugin.initialize, get_latents=get_latents,
                       get_noise=get_noise, latent_file="selection.npy", G_res=size, out_size=size, fps=fps, batch=batch,
                       truncation=truncation, output_file=str(tmp_path / "o.mp4"))
    _assert_summary(stubs.summary(seen["latents"]), fx, f"{tag}.latents", 2e-5)
    assert [nz is None for nz in seen["noise"]] == [bool(v) for v in fx[f"{tag}.noise_is_none"]]
    for i, nz in enumerate(seen["noise"]):
        if nz is not None:
            np.testing.assert_allclose(stubs.summary(nz)["stats"], fx[f"{tag}.noise_{i}.stats"], atol=2e-5, rtol=0)
    np.testing.assert_array_equal(np.load("workspace/last-latents.npy"), np.load("selection.npy"))
    frames = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n, size, size, 3)
    diff = np.abs(frames[:, 3::8, 5::8, :].astype(np.int16) - fx[f"{tag}.sub"].astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (tag, int(diff.max()), float((diff > 0).mean()))
    sums = frames.reshape(n, -1).sum(1).astype(np.int64)
    assert np.abs(sums - fx[f"{tag}.sums"]).max() < 2e-3 * size * size * 3  # whole frames, not just the subsample


def test_whole_workload_wav_to_frames_vs_oracle(gpu, tmp_path, monkeypatch):
    """BASELINE config 2 / 3 as a TEST: a seeded WAV goes through ``generate()`` with the default plugin and the real HIP
    audio kernels (HPSS, band onsets incl. complex flux, constant-Q / CENS chroma, temporal FIR), a seeded 512^2 checkpoint
    (the smallest size render() accepts), 75 frames at 25 fps in batches of 8 (hipGraph batches + eager tail), raw rgb24 sink.
    Checked, stage by stage, against the oracle chain on the CPU:
      * the onset / chroma envelopes the product computed vs signal_oracle on the same WAV (5e-3; chroma pitch class for pitch
        class, a column swap only where the oracle's medians tie: tests/chroma_check.py);
      * latents and noise maps vs oracle/plugin_oracle.py fed with the PRODUCT's envelopes and the same seeded randn draws
        (1e-4 / statistics 1e-4) — isolates everything between the feature calls and the generator;
      * delivered frames vs the oracle generator on the product's latents / noise for three frames (<= 1 grey level)."""
    import scipy.io.wavfile

    import maua_stylegan2_amd.audioreactive as ar
    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive.examples import default as plugin
    from oracle import plugin_oracle, signal_oracle
    from oracle import stylegan2_oracle as so

    stubs = _stubs()
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(render.shutil, "which", lambda name: None)
    sr, seconds, fps, size = 22050, 3.0, 25, 512
    y = seeding.synthetic_audio(seconds, sr)
    scipy.io.wavfile.write("track.wav", sr, (y * 32767).astype(np.int16))
    y = (y * 32767).astype(np.int16).astype(np.float32) / 32768.0  # what load_audio decodes
    sd = seeding.seeded_state_dict(size, seed=43)
    torch.save({"g_ema": sd}, "seeded512.pt")
    selection = seeding.seeded_array(42, "selection", (12, 16, 512))
    np.save("selection.npy", selection)
    monkeypatch.setattr(torch, "randn", stubs.SeededRandn(45))
    seen = {"noise": []}
    real_onsets, real_chroma = ar.onsets, ar.chroma

    def onsets(*a, **k):
        out = real_onsets(*a, **k)
        seen.setdefault("onsets", []).append((k, out.clone()))
        return out

    def chroma(*a, **k):
        seen["chroma"] = real_chroma(*a, **k)
        return seen["chroma"].clone()

    def get_latents(selection, args):
        seen["latents"] = plugin.get_latents(selection, args)
        return seen["latents"]

    def get_noise(height, width, scale, num_scales, args):
        nz = plugin.get_noise(height, width, scale, num_scales, args)
        seen["noise"].append(None if nz is None else nz.detach().cpu())
        return nz

    monkeypatch.setattr(ar, "onsets", onsets)
    monkeypatch.setattr(ar, "chroma", chroma)
    out = gav.generate(ckpt="seeded512.pt", audio_file="track.wav", initialize=plugin.initialize, get_latents=get_latents,
                       get_noise=get_noise, latent_file="selection.npy", G_res=size, out_size=size, fps=fps, batch=8,
                       output_file=str(tmp_path / "o.mp4"))
    n = int(round(seconds * fps))
    frames = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n, size, size, 3)
    # ---- stage 1: features vs the oracle on the same audio
    assert len(seen["onsets"]) == 2
    env = {}
    for kw, got in seen["onsets"]:
        want = signal_oracle.onsets(y, sr, n, type="mm", smf=fps / 30, **kw).numpy()
        np.testing.assert_allclose(got.numpy(), want, atol=5e-3, err_msg=str(kw))
        env["lo" if "fmax" in kw else "hi"] = got
    want, want_order, want_med = signal_oracle.chroma(y, sr, n, type="cens", nearest_neighbor=True, return_order=True)
    got = seen["chroma"].numpy()
    # pitch class for pitch class (tests/chroma_check.py): the delivered column order is recomputed from the product's own stages
    from chroma_check import check_chroma
    from test_signal_gpu import product_chroma_order

    from maua_stylegan2_amd.audioreactive import signal as sig

    got_order, got_med = product_chroma_order(sig, y, sr, n, "cens", got)
    check_chroma(got, got_order, got_med, want.numpy(), want_order, want_med)
    # ---- stage 2: plugin arithmetic vs the oracle plugin on the product's envelopes and the same random draws
    lat_want = plugin_oracle.get_latents(torch.from_numpy(selection), seen["chroma"], env["lo"], env["hi"], smf=fps / 30)
    np.testing.assert_allclose(seen["latents"].cpu().numpy(), lat_want.numpy(), atol=1e-4)
    randn = stubs.SeededRandn(45)
    sizes = seeding.noise_sizes(size)
    assert len(seen["noise"]) == len(sizes)
    for side, nz in zip(sizes, seen["noise"]):
        nz_want = plugin_oracle.get_noise(side, side, n, env["lo"], env["hi"], randn, smf=fps / 30)
        if nz_want is None:
            assert nz is None
            continue
        np.testing.assert_allclose(stubs.summary(nz)["stats"], stubs.summary(nz_want)["stats"], atol=1e-4, err_msg=str(side))
        np.testing.assert_allclose(stubs.summary(nz)["sub"], stubs.summary(nz_want)["sub"], atol=2e-4, err_msg=str(side))
    # ---- stage 3: frames vs the oracle generator on the product's latents / noise (first graph batch, a middle one, eager tail)
    for i in (0, 37, n - 1):
        lat_i = seen["latents"][i: i + 1].cpu()
        noise_i = [None if nz is None else nz[i: i + 1] for nz in seen["noise"]]
        want = so.frames_to_uint8(so.generator_forward(sd, lat_i, noise_i))[0]
        diff = np.abs(frames[i].astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


def test_generator_with_bends_and_wide_output_vs_oracle(gpu):
    """Config-5 style network bending through the whole generator: a layer-0 bend that widens the constant (the
    reference's route to 2:1 output, tauceti.py:97-100) plus a per-frame modulated Translate at layer 4 and a Zoom at
    layer 5; the oracle applies the same transforms (oracle warp = restated kornia composition) at the same layer ids."""
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive import bend
    from oracle import signal_oracle
    from oracle import stylegan2_oracle as so

    size, n = 64, 4
    sd = seeding.seeded_state_dict(size, seed=8)
    g = build(size, gpu, 8)
    lat = seeding.seeded_latents(n, g.n_latent, seed=9)
    sizes = seeding.noise_sizes(size)
    noise = [torch.from_numpy(seeding.seeded_array(10, f"wn{i}", (n, 1, r, 2 * r))) for i, r in enumerate(sizes)]
    h, w = 16, 32  # layer-4 feature size once the constant is 4 x 8
    shift = torch.tensor([[0.0, 0.0], [2.5, 0.0], [-4.0, 0.0], [7.75, 0.0]])
    zoom = torch.tensor([1.0, 1.3, 0.8, 1.1])
    bnoise = torch.from_numpy(seeding.seeded_array(11, "bend_noise", (1, 1, h, 5 * w))) * 0.05
    bends = [
        {"layer": 0, "transform": torch.nn.ReplicationPad2d((2, 2, 0, 0))},
        {"layer": 4, "modulation": shift, "transform": lambda b: bend.Translate(b, h, w, bnoise)},
        {"layer": 5, "modulation": zoom, "transform": lambda b: bend.Zoom(b, h, w)},
    ]
    frames = np.zeros((n, size, 2 * size, 3), np.uint8)
    for first, u8 in render.synthesize(g, lat, noise, 2, bends=bends):
        frames[first: first + u8.shape[0]] = u8.cpu().numpy()

    def o_translate(t):
        pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]  # the reference's three stacked pads
        m = bend._inverse_maps_translate(shift).numpy()
        return torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, pads, bnoise.numpy())).float()

    def o_zoom(t):
        pad = max(h, w) - 1
        m = bend._inverse_maps_scale(zoom, w + 2 * pad, h + 2 * pad).numpy()
        return torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, (pad,) * 4)).float()

    want = so.generator_forward(sd, lat, noise, bends={0: torch.nn.ReplicationPad2d((2, 2, 0, 0)), 4: o_translate, 5: o_zoom})
    assert want.shape == (n, 3, size, 2 * size)
    want_u8 = so.frames_to_uint8(want)
    diff = np.abs(frames.astype(np.int16) - want_u8.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 5e-3


def test_rewrites_vs_oracle(gpu):
    """Model rewriting (get_rewrites): a per-batch modulated scaling of one conv weight and of one ToRGB bias; the oracle
    runs each batch with the correspondingly rewritten state dict."""
    from maua_stylegan2_amd import render
    from oracle import stylegan2_oracle as so

    size, n, bs = 32, 4, 2
    sd = seeding.seeded_state_dict(size, seed=12)
    g = build(size, gpu, 12)
    lat = seeding.seeded_latents(n, g.n_latent, seed=13)
    noise = seeding.seeded_noise(n, size, seed=14)
    mod = torch.tensor([1.0, 1.5, 0.25, 2.0])

    class Scale(torch.nn.Module):
        def __init__(self, m):
            super().__init__()
            self.m = float(m.mean())

        def forward(self, w):
            return w * self.m

    rewrites = {"convs.2.conv.weight": [lambda m: Scale(m), mod], "to_rgbs.1.bias": [lambda m: Scale(m), mod]}
    frames = np.zeros((n, size, size, 3), np.uint8)
    for first, u8 in render.synthesize(g, lat, noise, bs, rewrites=rewrites):
        frames[first: first + u8.shape[0]] = u8.cpu().numpy()
    for b0 in range(0, n, bs):
        sd2 = dict(sd)
        k = float(mod[b0: b0 + bs].mean())
        sd2["convs.2.conv.weight"] = sd["convs.2.conv.weight"] * k
        sd2["to_rgbs.1.bias"] = sd["to_rgbs.1.bias"] * k
        want = so.frames_to_uint8(so.generator_forward(sd2, lat[b0: b0 + bs], [x[b0: b0 + bs] for x in noise]))
        diff = np.abs(frames[b0: b0 + bs].astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 5e-3
    # the generator is restored afterwards
    assert torch.equal(g.convs[2].conv.weight.cpu(), sd["convs.2.conv.weight"])


def test_concurrent_graph_lanes_give_the_same_frames(gpu):
    """synthesize() replays ``lanes`` hipGraphs round-robin on their own streams; the frames must not depend on it
    (same kernels, same batch size -> bit-identical), including the eager tail batch."""
    from maua_stylegan2_amd import render, seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = Generator(64, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(64, seed=5), strict=True)
    g = g.to(gpu).eval()
    n = 14  # 4 full batches of 3 + a tail of 2
    lat = seeding.seeded_latents(n, g.n_latent, seed=6)
    noise = [torch.from_numpy(seeding.seeded_array(7, f"n{i}", (n, 1, r, r))) if r <= 16 else None
             for i, r in enumerate(seeding.noise_sizes(64))]
    out = {}
    for lanes in (1, 2, 3):
        frames = []
        for first, u8 in render.synthesize(g, lat, noise, 3, lanes=lanes):
            assert first == len(frames)
            frames.extend(u8.cpu().numpy())
        out[lanes] = np.stack(frames)
        assert out[lanes].shape == (n, 64, 64, 3)
    assert np.array_equal(out[1], out[2]) and np.array_equal(out[1], out[3])


def test_graph_replay_never_writes_outside_its_static_buffers(gpu):
    """Regression: every address a captured graph writes must stay owned by the generator.  (A split-K workspace that was
    re-allocated per layer under one name left the graph writing into freed memory — found by a soak test.)  Device
    memory handed out by the allocator after the capture is filled with a pattern and must survive the replays."""
    from maua_stylegan2_amd import seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = Generator(256, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(256, seed=5), strict=True)
    g = g.to(gpu).eval()
    lat = seeding.seeded_latents(16, g.n_latent, seed=6).to(gpu)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        lane = g.capture_graph(8)
        lane.bind(lat, [None] * g.num_layers)
        stream.synchronize()
        sentinels = [torch.full((size,), 7, dtype=torch.uint8, device=gpu)
                     for size in [1 << 12, 1 << 16, 1 << 20, 3 << 19, 1 << 22, 1 << 24, 1 << 26] * 6]
        for k in range(2):
            lane.replay(8 * k)
        stream.synchronize()
        assert all(bool((s == 7).all()) for s in sentinels)


@pytest.mark.parametrize("lanes", [1, 3])
def test_synthesize_graph_lanes_match_eager_256_batch8(gpu, lanes):
    """The configuration in which the freed-workspace bug showed (256^2, 8 frames per batch, clones of every batch's
    uint8 frames allocated while later batches replay): graph path with 1 and 3 lanes == eager path, bit for bit."""
    from maua_stylegan2_amd import render, seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = Generator(256, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(256, seed=5), strict=True)
    g = g.to(gpu).eval()
    n, bs = 40, 8
    lat = seeding.seeded_latents(n, g.n_latent, seed=6)
    noise = [torch.from_numpy(seeding.seeded_array(7, f"n{i}", (n, 1, r, r))) if r <= 64 else None
             for i, r in enumerate(seeding.noise_sizes(256))]

    def run(**kw):
        frames = [u8.clone() for _, u8 in render.synthesize(g, lat, noise, bs, **kw)]
        torch.cuda.synchronize()
        return torch.cat(frames).cpu().numpy()

    eager = run(use_graph=False)
    assert np.array_equal(run(lanes=lanes), eager)


def test_render_rank_shards_on_device_equal_single_rank(gpu, tmp_path, monkeypatch):
    """The multi-rank branch of render() (every batch-round pushed into a sharding.FrameStream from the graph lanes) on ONE
    GPU: rank_world and the gather collective are stubbed so that this process plays rank 0 and rank 1 of a 2-rank job in
    turn; the two blocks concatenated must equal the single-rank render, frame for frame, and rank 0's sink must have
    written its own block in order."""
    from maua_stylegan2_amd import render, sharding

    monkeypatch.setattr(render.shutil, "which", lambda name: None)
    g = build(512, gpu, 1)
    n = 11  # uneven shards: 6 + 5, batches of 2 -> graph batches and an eager tail on each rank
    lat = seeding.seeded_latents(n, g.n_latent, seed=2)
    noise = [None] * g.num_layers
    single = str(tmp_path / "single.mp4")
    assert render.render(g, lat, noise, 0, n / 30, 2, 512, single) == n
    want = np.fromfile(single + ".rgb24", dtype=np.uint8).reshape(n, 512, 512, 3)

    class _Done:
        def wait(self):
            return True

        def is_completed(self):
            return True

    def fake_gather(tensor, gather_list=None, dst=0, group=None, async_op=False):
        if gather_list is not None:  # "rank 0": its own slot arrives, the peer's slot stays as allocated
            gather_list[0].copy_(tensor)
        return _Done()

    monkeypatch.setattr(sharding.dist, "gather", fake_gather)
    shards, streams = {}, []
    real_stream = sharding.FrameStream

    class Recording(real_stream):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            streams.append(self)

    monkeypatch.setattr(sharding, "FrameStream", Recording)
    for rank in (0, 1):
        monkeypatch.setattr(sharding, "rank_world", lambda r=rank: (r, 2))
        monkeypatch.setattr(sharding, "grouped", lambda: True)  # (the collective branches are taken whenever a process group exists)
        lo, hi = sharding.shard_bounds(n, rank, 2)
        out = str(tmp_path / f"rank{rank}.mp4")
        written = render.render(g, lat, noise, 0, n / 30, 2, 512, out)
        torch.cuda.synchronize()
        stream = streams[-1]
        assert stream.rounds == 3 and stream.pushed == 3 and (stream.lo, stream.hi) == (lo, hi)
        shards[rank] = stream.mine[: hi - lo].cpu().numpy()
        if rank == 0:  # rank 0's sink consumed every frame slot in order; its own block holds real frames
            assert written == n
            raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n, 512, 512, 3)
            assert np.array_equal(raw[: hi - lo], want[: hi - lo])
        else:
            assert written == 0
    got = np.concatenate([shards[0], shards[1]])
    assert got.shape == want.shape and np.array_equal(got, want)


def test_captured_bends_equal_eager_bends_and_oracle(gpu):
    """BASELINE config 5's workload inside the captured forward: a per-frame modulated Translate at layer id 4 and a Zoom at
    layer id 5 (audioreactive/bend.py ``run_static``: the frame's inverse map is picked on the device through the frame
    source).  Graph path (3 lanes, 5 graph batches + eager tail) == eager per-batch path bit for bit (reference render.py:151-158
    rebuilds the transform per batch), and both match the oracle's transforms at the same layer ids to one grey level."""
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive import bend
    from oracle import signal_oracle
    from oracle import stylegan2_oracle as so

    size, n, bs = 64, 11, 2
    lat = seeding.seeded_latents(n, 10, seed=9)
    noise = [torch.from_numpy(seeding.seeded_array(10, f"sq{i}", (n, 1, r, r))) if r <= 32 else None
             for i, r in enumerate(seeding.noise_sizes(size))]
    sd, g = build_unsaturated(size, gpu, 8, lat, noise)  # (< 5 % of the frame on the uint8 clamp: asserted below)
    h = w = 16
    shift = torch.stack([torch.linspace(0.0, 1.5 * w, n), torch.zeros(n)], 1)  # scrolls by more than one width: the stacked pads
    zoom = 1.0 + 0.3 * torch.sin(torch.arange(n) / 2.0)
    bnoise = torch.from_numpy(seeding.seeded_array(11, "bend_noise", (1, 1, h, 5 * w))) * 0.05

    def bends():
        return [{"layer": 4, "modulation": shift.clone(), "transform": lambda b: bend.Translate(b, h, w, bnoise)},
                {"layer": 5, "modulation": zoom.clone(), "transform": lambda b: bend.Zoom(b, h, w)}]

    seq, ok = render._sequence_bends([dict(b, modulation=b["modulation"].to(gpu)) for b in bends()])
    assert ok and all(hasattr(b["transform"], "run_static") for b in seq)

    def run(use_graph):
        frames = np.zeros((n, size, size, 3), np.uint8)
        for first, u8 in render.synthesize(g, lat, noise, bs, bends=bends(), use_graph=use_graph):
            frames[first: first + u8.shape[0]] = u8.cpu().numpy()
        return frames

    graphed, eager = run(True), run(False)
    assert np.array_equal(graphed, eager)
    assert not getattr(g, "_graph_lanes", {}), "graphs captured with bends are per render, not cached on the generator"

    def o_translate(t):
        pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]
        return torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), bend._inverse_maps_translate(shift).numpy(), pads,
                                                                  bnoise.numpy())).float()

    def o_zoom(t):
        pad = max(h, w) - 1
        m = bend._inverse_maps_scale(zoom, w + 2 * pad, h + 2 * pad).numpy()
        return torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, (pad,) * 4)).float()

    noise_o = [nz if nz is not None else sd[f"noises.noise_{i}"] for i, nz in enumerate(noise)]
    want_f = so.generator_forward(sd, lat, noise_o, bends={4: o_translate, 5: o_zoom})
    assert seeding.clamped_fraction(want_f) < 0.05
    want = so.frames_to_uint8(want_f)
    diff = np.abs(graphed.astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 5e-3
    # a transform without the protocol (torch module) makes the render fall back to the eager path instead of failing
    seq, ok = render._sequence_bends([{"layer": 0, "transform": torch.nn.ReplicationPad2d((2, 2, 0, 0))}])
    assert not ok
    # a modulation shorter than the sequence must never reach a captured forward (its kernel indexes the table with frame0 + b
    # unchecked): the render keeps the eager per-batch path, which slices and validates like the reference (render.py:151-158)
    short = [{"layer": 5, "modulation": zoom[: n - 3].clone().to(gpu), "transform": lambda b: bend.Zoom(b, h, w)}]
    assert render._sequence_bends(short, n) == (None, False) and render._sequence_bends(short, n - 3)[1]
    with pytest.raises(RuntimeError, match="inverse affine maps"):
        for _ in render.synthesize(g, lat, noise, bs, bends=[dict(short[0], modulation=zoom[: n - 3].clone())]):
            pass


def test_cached_graph_lanes_serve_a_second_render_and_follow_weight_changes(gpu):
    """Graph lanes are captured once per (batch, lane) and reused by later renders with other sequences (nothing of a render is
    baked into a captured forward); loading other weights drops them."""
    from maua_stylegan2_amd import render

    size, n, bs = 32, 8, 4
    g = build(size, gpu, 4)

    def frames_of(seed, use_graph):
        lat = seeding.seeded_latents(n, g.n_latent, seed=seed)
        noise = seeding.seeded_noise(n, size, seed=seed + 1)
        out = np.zeros((n, size, size, 3), np.uint8)
        for first, u8 in render.synthesize(g, lat, noise, bs, use_graph=use_graph, lanes=2):
            out[first: first + u8.shape[0]] = u8.cpu().numpy()
        return out

    a = frames_of(20, True)
    lanes_before = dict(g._graph_lanes)
    assert len(lanes_before) == 2
    b = frames_of(30, True)
    assert all(g._graph_lanes[k] is v for k, v in lanes_before.items()), "second render re-captured its graphs"
    assert np.array_equal(a, frames_of(20, False)) and np.array_equal(b, frames_of(30, False))
    g.load_state_dict(seeding.seeded_state_dict(size, seed=5), strict=True)
    c = frames_of(20, True)
    assert all(g._graph_lanes[k] is not v for k, v in lanes_before.items()), "stale graphs survived a weight change"
    assert np.array_equal(c, frames_of(20, False)) and not np.array_equal(c, a)


@pytest.mark.parametrize("unsaturated", [True, False])
def test_bench_configuration_1024_batch8_three_lanes_vs_oracle(gpu, unsaturated):
    """The configuration bench.py times — 1024^2 generator, batches of 8 frames, 3 graph lanes, per-frame noise up to 256^2 and
    checkpoint buffers above, uint8 frames written by the last layer's fused epilogue — against the ORACLE on FULL frames, in
    FLOAT: the lanes are captured with ``tap_float_image`` (the very kernel launch that writes the frame also leaves its fp32 image),
    frame 19 (third lane's first replay) and frame 33 (second lane's second replay) must agree with the oracle within the north_star's
    1e-3 (reference models/stylegan2.py:492-576).  Two checkpoints: the plain seeded one (image std ~3: the hard case for the float
    bound, but 3/4 of its uint8 frame clamps) and an unsaturated one (< 5 % of the frame on the clamp, asserted), on which the uint8
    frame itself is compared as well (<= 1 grey level, render.py:40-43) and equals the cast of the tapped float image bit for bit."""
    from maua_stylegan2_amd import render
    from oracle import stylegan2_oracle as so

    size, n, bs, n_lanes = 1024, 40, 8, 3
    lat = seeding.seeded_latents(n, 18, seed=100)
    noise = [torch.from_numpy(seeding.seeded_array(200, f"n{i}", (n, 1, r, r))) if r <= 256 else None
             for i, r in enumerate(seeding.noise_sizes(size))]
    if unsaturated:
        sd, g = build_unsaturated(size, gpu, 0, lat, noise)
    else:
        sd, g = seeding.seeded_state_dict(size, seed=0), build(size, gpu, 0)
    g.tap_float_image = True
    picks = {19: None, 33: None}
    batches = 0
    for first, u8 in render.synthesize(g, lat, noise, bs, lanes=n_lanes):
        assert u8.shape == (bs, size, size, 3)
        lane = lane_of(g, bs, batches, n_lanes)
        assert lane.u8 is u8 and lane.image is not None and tuple(lane.image.shape) == (bs, 3, size, size)
        for i in picks:
            if first <= i < first + bs:
                torch.cuda.current_stream().synchronize()
                picks[i] = (u8[i - first].cpu().numpy(), lane.image[i - first].cpu())
                assert np.array_equal(so.frames_to_uint8(picks[i][1][None])[0], picks[i][0]), "frame != cast of the tapped float image"
        batches += 1
    assert batches == 5 and len(g._graph_lanes) == 3
    for i, (got_u8, got_f) in picks.items():
        noise_i = [sd[f"noises.noise_{k}"] if nz is None else nz[i: i + 1] for k, nz in enumerate(noise)]
        want = so.generator_forward(sd, lat[i: i + 1], noise_i)[0]
        err = float((got_f - want).abs().max())
        clamped = seeding.clamped_fraction(want)
        print(f"[bench configuration, frame {i}, {'unsaturated' if unsaturated else 'plain'} checkpoint] image std {float(want.std()):.3f}, "
              f"max |hip - oracle| = {err:.3e} (float, full frame), clamped pixels {100 * clamped:.1f} %")
        assert err < 1e-3, (i, err)
        if unsaturated:
            assert clamped < 0.05, clamped
            diff = np.abs(got_u8.astype(np.int16) - so.frames_to_uint8(want[None])[0].astype(np.int16))
            assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


def test_config5_bends_1024_batch8_three_lanes_vs_oracle(gpu):
    """BASELINE config 5's workload AT FULL SIZE against the oracle (round 3 checked 64^2 against the oracle and 1024^2 only graph
    vs eager): the bends bench.py --bends times — a per-frame modulated Translate at layer id 4 and Zoom at layer id 5 (16x16
    features of the 1024^2 generator, usage audioreactive/examples/tauceti.py:94-159, transforms audioreactive/bend.py:52-102) —
    inside the captured forward, batches of 8 on 3 lanes.  Frames 11 and 21 in float against the oracle generator with the oracle's
    own warps at the same layer ids (1e-3), and as uint8 frames on an unsaturated checkpoint (<= 1 grey level)."""
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive import bend
    from oracle import signal_oracle
    from oracle import stylegan2_oracle as so

    size, n, bs, n_lanes = 1024, 24, 8, 3
    lat = seeding.seeded_latents(n, 18, seed=101)
    noise = [torch.from_numpy(seeding.seeded_array(201, f"n{i}", (n, 1, r, r))) if r <= 256 else None
             for i, r in enumerate(seeding.noise_sizes(size))]
    sd, g = build_unsaturated(size, gpu, 0, lat, noise)
    g.tap_float_image = True
    h = w = 16
    saw = (torch.arange(n, dtype=torch.float32) * 7.0 % 24.0) / 24.0 * 1.5 * w  # scrolls by more than one width: the stacked pads
    shift = torch.stack([saw, torch.zeros(n)], 1)
    zoom = 1.0 + 0.25 * torch.sin(torch.arange(n, dtype=torch.float32) / 3.0) ** 2
    bnoise = torch.from_numpy(seeding.seeded_array(11, "bend_noise", (1, 1, h, 5 * w))) * 0.2

    def bends():
        return [{"layer": 4, "modulation": shift.clone(), "transform": lambda b: bend.Translate(b, h, w, bnoise)},
                {"layer": 5, "modulation": zoom.clone(), "transform": lambda b: bend.Zoom(b, h, w)}]

    picks = {11: None, 21: None}
    taps = {}
    orig_capture = g.capture_graph

    def capture(batch, lane=0, frames_u8=False, bends=()):  # graphs with bends are per render (not cached): keep the lanes here
        taps[lane] = orig_capture(batch, lane=lane, frames_u8=frames_u8, bends=bends)
        return taps[lane]

    g.capture_graph = capture
    batches = 0
    for first, u8 in render.synthesize(g, lat, noise, bs, bends=bends(), lanes=n_lanes):
        lane = taps[batches % n_lanes]
        assert lane.u8 is u8 and lane.image is not None
        for i in picks:
            if first <= i < first + bs:
                torch.cuda.current_stream().synchronize()
                picks[i] = (u8[i - first].cpu().numpy(), lane.image[i - first].cpu())
        batches += 1
    assert batches == 3 and len(taps) == 3

    def o_translate(i):
        pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]
        m = bend._inverse_maps_translate(shift[i: i + 1]).numpy()
        return lambda t: torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, pads, bnoise.numpy())).float()

    def o_zoom(i):
        pad = max(h, w) - 1
        m = bend._inverse_maps_scale(zoom[i: i + 1], w + 2 * pad, h + 2 * pad).numpy()
        return lambda t: torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, (pad,) * 4)).float()

    for i, (got_u8, got_f) in picks.items():
        noise_i = [sd[f"noises.noise_{k}"] if nz is None else nz[i: i + 1] for k, nz in enumerate(noise)]
        want = so.generator_forward(sd, lat[i: i + 1], noise_i, bends={4: o_translate(i), 5: o_zoom(i)})[0]
        plain = so.generator_forward(sd, lat[i: i + 1], noise_i)[0]
        err = float((got_f - want).abs().max())
        clamped = seeding.clamped_fraction(want)
        print(f"[config 5 at 1024^2, frame {i}] image std {float(want.std()):.3f}, max |hip - oracle| = {err:.3e} (float, full frame), "
              f"clamped pixels {100 * clamped:.1f} %, bends move the image by {float((want - plain).abs().mean()):.3f} on average")
        assert float((want - plain).abs().mean()) > 0.05, "the bends must change the frame for this test to mean anything"
        assert err < 1e-3 and clamped < 0.05, (i, err, clamped)
        diff = np.abs(got_u8.astype(np.int16) - so.frames_to_uint8(want[None])[0].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


def test_config3_900_frames_through_generate_vs_oracle(gpu, tmp_path, monkeypatch):
    """BASELINE config 3 at FULL size through the drop-in surface: a 30 s seeded track -> ``generate()`` with the default
    audio-reactive plugin (HIP feature kernels, chroma-weighted latents, reactive noise <= 256^2, checkpoint buffers above) on a
    seeded 1024^2 checkpoint, 900 frames at 30 fps in batches of 8 (112 graph replays on 3 lanes + an eager tail of 4), delivered
    in order to a host sink that keeps three frames.  Those frames are compared with the oracle generator run on the latents /
    noise the product's callbacks returned (<= 1 grey level); every frame must arrive exactly once and in order."""
    import wave

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive.examples import default as plugin
    from oracle import stylegan2_oracle as so

    monkeypatch.chdir(tmp_path)
    size, seconds, fps = 1024, 30.0, 30
    n = int(round(seconds * fps))
    # ToRGB gain 0.12: the frames use the grey range without sitting on the clamp (asserted below on the oracle's float image)
    sd = seeding.seeded_state_dict(size, seed=0, rgb_gain=0.12)
    torch.save({"g_ema": sd}, "seeded1024.pt")
    audio = seeding.synthetic_audio(seconds)
    with wave.open("track.wav", "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((np.clip(audio, -1, 1) * 32767).astype(np.int16).tobytes())
    np.save("selection.npy", seeding.seeded_array(42, "selection", (12, 18, 512)))
    keep = {0: None, 452: None, n - 1: None}
    order = []

    class KeepingSink(render.FrameSink):
        def __init__(self, *a, **k):
            self.count = 0

        def write(self, frame):
            if self.count in keep:
                keep[self.count] = np.array(frame, copy=True)
            if self.count % 64 == 0:
                order.append((self.count, int(frame[::64, ::64].astype(np.int64).sum())))
            self.count += 1

        def close(self):
            pass

    monkeypatch.setattr(render, "FrameSink", KeepingSink)
    seen = {"noise": []}

    def get_latents(selection, args):
        seen["latents"] = plugin.get_latents(selection, args)
        return seen["latents"]

    def get_noise(height, width, scale, num_scales, args):
        nz = plugin.get_noise(height, width, scale, num_scales, args)
        seen["noise"].append(None if nz is None else nz.detach().cpu())
        return nz

    gav.generate(ckpt="seeded1024.pt", audio_file="track.wav", initialize=plugin.initialize, get_latents=get_latents,
                 get_noise=get_noise, latent_file="selection.npy", G_res=size, out_size=size, fps=fps, batch=8,
                 output_file=str(tmp_path / "o.mp4"))
    assert tuple(seen["latents"].shape) == (n, 18, 512) and len(seen["noise"]) == 17
    assert [i for i, _ in order] == list(range(0, n, 64)) and len({c for _, c in order}) > len(order) // 2  # distinct frames, in order
    for i, got in keep.items():
        assert got is not None and got.shape == (size, size, 3), i
        noise_i = [sd[f"noises.noise_{k}"] if nz is None else nz[i: i + 1] for k, nz in enumerate(seen["noise"])]
        want_f = so.generator_forward(sd, seen["latents"][i: i + 1].cpu().float(), noise_i)
        print(f"[config 3, frame {i}] oracle image std {float(want_f.std()):.3f}, clamped {100 * seeding.clamped_fraction(want_f):.1f} %")
        assert float(want_f.std()) > 0.15 and seeding.clamped_fraction(want_f) < 0.05
        want = so.frames_to_uint8(want_f)[0]
        diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


@pytest.mark.parametrize("out_size,shape", [(1920, (2, 1024, 2048, 3)), (1080, (1, 2048, 1024, 3))])
def test_wide_output_crop_resize_on_device_equals_pil(gpu, out_size, shape):
    """render.py:97-104 (2048-px frames -> crop 112 px off both ends of the long side -> PIL bilinear resize to 1920x1080 /
    1080x1920) on the device (maua_crop_resize_u8) against PIL itself on the same frames: bit-equal (<= 1 grey level is the bar);
    other frame sizes pass through untouched."""
    import PIL.Image

    from maua_stylegan2_amd import render

    frames = torch.from_numpy(np.random.default_rng(out_size).integers(0, 256, shape, dtype=np.uint8))
    scratch = {}
    got = render.crop_resize_for_delivery(frames.to(gpu), out_size, scratch)
    torch.cuda.synchronize()
    w, h = render._output_dims(out_size)
    assert tuple(got.shape) == (shape[0], h, w, 3) and len(scratch) == 1
    for i in range(shape[0]):
        f = frames[i].numpy()
        crop = f[:, 112:-112, :] if out_size == 1920 else f[112:-112, :, :]
        want = np.array(PIL.Image.fromarray(np.ascontiguousarray(crop)).resize((w, h), PIL.Image.BILINEAR))
        diff = np.abs(got[i].cpu().numpy().astype(np.int16) - want.astype(np.int16))
        assert diff.max() == 0, (i, int(diff.max()), float((diff > 0).mean()))
    square = torch.zeros(1, 512, 512, 3, dtype=torch.uint8, device=gpu)
    assert render.crop_resize_for_delivery(square, 512, scratch) is square
    # the sink accepts the delivered size directly (no host-side PIL pass any more)
    sink = render.FrameSink(None, w, h, 30)
    sink.write(got[0].cpu().numpy())
    assert sink.count == 1


def test_stylegan1_through_generate_and_render_vs_oracle(gpu, tmp_path, monkeypatch):
    """``--stylegan1`` end to end (reference generate_audiovisual.py:41-47, models/stylegan1.py:509-617): a seeded 128-px G_style
    checkpoint is probed (1024 -> 512 -> 256 -> 128) by ``generate(stylegan1=True)``, rendered at 512^2 (constant enlarged to 32x32
    and centre-cropped to 16x16, six blocks, truncation 0.7 on the first 8 layers, per-frame noise below 64 px and the generator's
    own noise buffers above) through render()'s captured lanes (+ the eager tail batch), and three delivered frames are compared with oracle/stylegan1_oracle.py
    run on the state the generator actually holds (<= 1 grey level)."""
    import wave

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.models import stylegan1 as sg1
    from oracle import stylegan1_oracle as s1o
    from oracle import stylegan2_oracle as so

    monkeypatch.chdir(tmp_path)
    torch.manual_seed(11)
    proto = sg1.G_style(output_size=512, checkpoint=None, network_resolution=128)
    state = {}
    for key, value in proto.state_dict().items():
        if key.startswith("noise_"):
            continue
        shape = (1, 512, 4, 4) if key.endswith("4x4.const") else tuple(value.shape)
        std = 0.3 if key.endswith("noise.weight") else (0.2 if key.endswith(".bias") else 1.0)
        if "torgb" in key:
            std *= 0.2  # image std ~0.35: the uint8 frames stay off the clamp (asserted below), where an error would be invisible
        state[key] = value.clone() if key.endswith("kernel") else torch.from_numpy(seeding.seeded_array(77, key, shape, std=std))
    torch.save(state, "sg1_128.pt")
    del proto
    n, fps, bs = 10, 10, 4
    audio = seeding.synthetic_audio(n / fps)
    with wave.open("track.wav", "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((np.clip(audio, -1, 1) * 32767).astype(np.int16).tobytes())
    # (G_mapping broadcasts to 18 latents whatever the network resolution, models/stylegan1.py:357-362: the synthesis network of a
    # 128-px checkpoint reads the first 12 of them, the truncation lerp covers the first 8)
    np.save("selection.npy", seeding.seeded_array(78, "selection", (4, 18, 512)))
    lat = torch.from_numpy(seeding.seeded_array(79, "latents", (n, 18, 512)))
    noise_seq = {}
    keep, held = {0: None, 5: None, n - 1: None}, {}

    class KeepingSink(render.FrameSink):
        def __init__(self, *a, **k):
            self.count = 0

        def write(self, frame):
            if self.count in keep:
                keep[self.count] = np.array(frame, copy=True)
            self.count += 1

        def close(self):
            pass

    monkeypatch.setattr(render, "FrameSink", KeepingSink)
    real_load = gav.load_generator

    def load_generator(**kw):
        held["g"] = real_load(**kw)
        return held["g"]

    monkeypatch.setattr(gav, "load_generator", load_generator)

    def get_noise(height, width, scale, num_scales, args):
        if width > 32:
            return None  # the generator's own noise_i buffer
        noise_seq[scale] = torch.from_numpy(seeding.seeded_array(80, f"nz{scale}", (args.n_frames, 1, height, width)))
        return noise_seq[scale]

    gav.generate(ckpt="sg1_128.pt", audio_file="track.wav", get_latents=lambda selection, args: lat.clone(), get_noise=get_noise,
                 latent_file="selection.npy", stylegan1=True, G_res=128, out_size=512, fps=fps, batch=bs, truncation=0.7,
                 output_file=str(tmp_path / "o.mp4"))
    g = held["g"]
    assert g.network_resolution == 128 and len(g.g_synthesis.blocks) == 6
    assert tuple(getattr(g.g_synthesis.blocks, "4x4").const.shape) == (1, 512, 16, 16)
    sd = {k: v.detach().cpu() for k, v in g.state_dict().items()}
    tl = g.truncation_latent.cpu()
    for i, got in keep.items():
        assert got is not None and got.shape == (512, 512, 3), i
        noise_i = [noise_seq[s][i: i + 1] if s in noise_seq else sd[f"noise_{s}"] for s in range(6)]
        want_f = s1o.synthesis(sd, s1o.truncate(lat[i: i + 1], tl, 0.7), noise_i)
        assert float(want_f.std()) > 0.15 and seeding.clamped_fraction(want_f) < 0.05, (float(want_f.std()), seeding.clamped_fraction(want_f))
        want = so.frames_to_uint8(want_f)[0]
        diff = np.abs(got.astype(np.int16) - want.astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


def test_stylegan1_captured_forward_equals_eager(gpu):
    """G_style.capture_graph (hipGraph stream capture of the whole StyleGAN1 forward + frame epilogue, three lanes): frames bit-equal
    to the eager forward — per-frame noise on the small blocks, the generator's buffers on the others, per-frame truncation — the
    lanes are cached on the generator and serve a second render with other sequences."""
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.models import stylegan1 as sg1

    torch.manual_seed(5)
    g = sg1.G_style(output_size=1024, checkpoint=None, network_resolution=128).cuda().eval()
    n, bs = 14, 4
    for seed in (0, 1):
        lat = torch.from_numpy(seeding.seeded_array(90 + seed, "lat", (n, 18, 512))).cuda()
        noise = [torch.from_numpy(seeding.seeded_array(91 + seed, f"nz{i}", (n if i != 1 else 1,) + tuple(getattr(g, f"noise_{i}").shape[1:]))).cuda()
                 if i < 3 else None for i in range(6)]  # per-frame on blocks 0 and 2, one shared map on block 1, the generator's buffers above
        trunc = torch.linspace(0.5, 1.0, n).cuda()
        got = {k: u8.cpu().clone() for k, u8 in render.synthesize(g, lat, noise, bs, truncation=trunc, lanes=3)}
        want = {k: u8.cpu().clone() for k, u8 in render.synthesize(g, lat, noise, bs, truncation=trunc, use_graph=False)}
        assert sorted(got) == sorted(want) == [0, 4, 8, 12]
        for k in got:
            assert torch.equal(got[k], want[k]), (seed, k)
        assert got[0].shape == (bs, 1024, 1024, 3) and not torch.equal(got[0][0], got[0][1])  # (32 x 32 constant, six blocks)
        lanes = g.__dict__["_graph_lanes"]
        assert sorted(lanes) == [(bs, 0, False), (bs, 1, False), (bs, 2, False)]
        if seed == 0:
            first = [lanes[key] for key in sorted(lanes)]
        else:
            assert all(lanes[key] is lane for key, lane in zip(sorted(lanes), first))  # nothing re-captured


def test_generate_1920_wide_output_is_delivered_as_1080p(gpu, tmp_path, monkeypatch):
    """``--out_size 1920`` end to end: a generator built for 1920 output (2:1 noise buffers) plus the layer-0 bend that widens the
    4x4 constant to 4x8 (the reference's route, examples/tauceti.py:97-100) renders 1024 x 2048 frames; the delivery crops 112 px
    per side and resizes to 1920 x 1080 ON THE DEVICE (reference render.py:97-100 does it with PIL per frame on the host).  A
    delivered frame is compared with the oracle's 1024 x 2048 frame taken through PIL's crop + resize (<= 1 grey level)."""
    import wave

    import PIL.Image

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from oracle import stylegan2_oracle as so

    monkeypatch.chdir(tmp_path)
    size, n, fps = 1024, 6, 6
    sd = seeding.seeded_state_dict(size, seed=0)
    torch.save({"g_ema": sd}, "seeded1024.pt")
    audio = seeding.synthetic_audio(n / fps)
    with wave.open("track.wav", "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((np.clip(audio, -1, 1) * 32767).astype(np.int16).tobytes())
    np.save("selection.npy", seeding.seeded_array(42, "selection", (4, 18, 512)))
    lat = torch.from_numpy(seeding.seeded_array(43, "latents", (n, 18, 512)))
    noise_seq, frames, held = {}, [], {}

    class KeepingSink(render.FrameSink):
        def __init__(self, output_file, width, height, *a, **k):
            self.count, self.w, self.h = 0, width, height

        def write(self, frame):
            assert frame.shape == (1080, 1920, 3)  # already resized: the host-side PIL pass is not taken
            frames.append(np.array(frame, copy=True))
            self.count += 1

        def close(self):
            pass

    monkeypatch.setattr(render, "FrameSink", KeepingSink)
    real_load = gav.load_generator

    def load_generator(**kw):
        held["g"] = real_load(**kw)
        return held["g"]

    monkeypatch.setattr(gav, "load_generator", load_generator)

    def get_noise(height, width, scale, num_scales, args):
        if width > 64:
            return None
        noise_seq[scale] = torch.from_numpy(seeding.seeded_array(44, f"nz{scale}", (args.n_frames, 1, height, width)))
        return noise_seq[scale]

    pad = torch.nn.ReplicationPad2d((2, 2, 0, 0))
    gav.generate(ckpt="seeded1024.pt", audio_file="track.wav", get_latents=lambda selection, args: lat.clone(), get_noise=get_noise,
                 get_bends=lambda args: [{"layer": 0, "transform": pad}], latent_file="selection.npy", G_res=size, out_size=1920,
                 fps=fps, batch=4, output_file=str(tmp_path / "o.mp4"))
    assert len(frames) == n
    g = held["g"]
    i = 4  # second batch (eager: the layer-0 torch bend is not capturable)
    noise_i = [noise_seq[s][i: i + 1] if s in noise_seq else getattr(g.noises, f"noise_{s}").cpu() for s in range(g.num_layers)]
    wide = so.frames_to_uint8(so.generator_forward(sd, lat[i: i + 1], noise_i, bends={0: pad}))[0]
    assert wide.shape == (1024, 2048, 3)
    want = np.array(PIL.Image.fromarray(np.ascontiguousarray(wide[:, 112:-112, :])).resize((1920, 1080), PIL.Image.BILINEAR))
    diff = np.abs(frames[i].astype(np.int16) - want.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (int(diff.max()), float((diff > 0).mean()))


def test_generator_is_kept_across_generate_calls_and_follows_the_checkpoint_file(gpu, tmp_path, monkeypatch):
    """Round 6: a process that renders job after job from one checkpoint (reference generate_audiovisual.py:37-56 reloads per call) keeps
    the generator — packed weights, captured lanes — as long as the FILE is the same (path, mtime, size) and the architecture flags are;
    the frames of the second job equal the first's bit for bit; a rewritten checkpoint is loaded again; MAUA_GENERATOR_CACHE=0 and a
    process group always load.  The module is built on the device (no CPU copy first)."""
    import scipy.io.wavfile

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive.examples import default as plugin

    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(render.shutil, "which", lambda name: None)
    monkeypatch.delenv("MAUA_GENERATOR_CACHE", raising=False)
    gav._GENERATOR_CACHE.clear()
    sr = 22050
    scipy.io.wavfile.write("track.wav", sr, (seeding.synthetic_audio(1.0, sr) * 32767).astype(np.int16))
    np.save("lat.npy", seeding.seeded_latents(12, 16, seed=3).numpy())
    torch.save({"g_ema": seeding.seeded_state_dict(512, seed=1)}, "a.pt")
    loads = []
    real_load = gav.load_generator
    monkeypatch.setattr(gav, "load_generator", lambda **kw: loads.append(kw["ckpt"]) or real_load(**kw))

    def job(name, ckpt="a.pt"):
        torch.manual_seed(11), np.random.seed(11)  # (the plugin's noise draws)
        out = gav.generate(ckpt=ckpt, audio_file="track.wav", initialize=plugin.initialize, get_latents=plugin.get_latents,
                           get_noise=plugin.get_noise, latent_file="lat.npy", G_res=512, out_size=512, fps=12, batch=4,
                           output_file=str(tmp_path / name))
        return np.fromfile(out + ".rgb24", dtype=np.uint8)

    first = job("1.mp4")
    g1 = gav._GENERATOR_CACHE["entry"][1]
    assert all(p.is_cuda for p in g1.parameters()) and len(loads) == 1
    lanes1 = dict(g1._graph_lanes)
    second = job("2.mp4")
    assert len(loads) == 1 and gav._GENERATOR_CACHE["entry"][1] is g1
    assert all(g1._graph_lanes[k] is v for k, v in lanes1.items())  # the captured lanes served the second job as they were
    assert np.array_equal(first, second)
    torch.save({"g_ema": seeding.seeded_state_dict(512, seed=2)}, "a.pt")  # the same path, another checkpoint
    third = job("3.mp4")
    assert len(loads) == 2 and gav._GENERATOR_CACHE["entry"][1] is not g1 and not np.array_equal(first, third)
    monkeypatch.setenv("MAUA_GENERATOR_CACHE", "0")
    fourth = job("4.mp4")
    assert len(loads) == 3 and np.array_equal(third, fourth)
    gav._GENERATOR_CACHE.clear()
