#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE (run in the build container only).

    python tests/golden/make_golden.py [--full]

Imports /root/reference (read-only) with the nvcc JIT loader stubbed (SURVEY.md §8c recipe) so that the
reference's own pure-PyTorch CPU fallbacks run (op/upfirdn2d.py:159-200, op/fused_act.py:87-94), feeds them
seeded inputs, asserts the repo's oracle/ restatement agrees (<= 1e-5 abs), and writes small .npz files
holding *inputs (or their seeds) and the reference's outputs only*.  Generator weights are never stored:
they are regenerated from numpy default_rng streams by maua_stylegan2_amd/seeding.py.

The reference cannot travel to the GPU box; these fixtures + this script are what travels.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REPO)

torch.set_grad_enabled(False)
torch.manual_seed(0)


def import_reference():
    import torch.utils.cpp_extension as cpp_ext

    cpp_ext.load = lambda name, sources, **kw: types.SimpleNamespace()  # op/*.py call load() at import
    for name in [
        "librosa", "librosa.display", "madmom", "kornia", "kornia.augmentation", "kornia.geometry",
        "kornia.geometry.transform", "ffmpeg", "torchvision", "torchvision.utils",
    ]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
    sys.path.insert(0, REF)
    import models.stylegan2 as ref_sg2  # noqa
    import op as ref_op  # noqa
    import audioreactive.signal as ref_signal_mod  # noqa  (module attr is shadowed by scipy.signal, use sys.modules)
    import audioreactive.latent as ref_latent  # noqa
    import generate_audiovisual as ref_gav  # noqa

    return ref_sg2, ref_op, sys.modules["audioreactive.signal"], sys.modules["audioreactive.latent"], ref_gav


def rng(seed):
    return np.random.default_rng(seed)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a.astype(np.float32)))


def check(name, mine, ref, tol=1e-5):
    err = float((mine - ref).abs().max()) if mine.numel() else 0.0
    print(f"  {name:58s} oracle-vs-reference max|err| = {err:.3e}")
    assert err <= tol, (name, err)
    return err


def perlin_fixture(ref_latent, signal_oracle):
    """perlin_noise (audioreactive/latent.py:188-246) hard-codes ``.cuda()`` (:207,219) and draws its gradient angles from
    numpy's GLOBAL generator (:209-210).  Here Tensor.cuda is replaced by the identity for the duration of the call, so
    the reference's own arithmetic runs on the CPU, and the global generator is seeded so the same two draws can be
    repeated for the oracle.  Stored: the angles (inputs) and the reference's output."""
    print("perlin_noise (reference run with Tensor.cuda -> identity, numpy global RNG seeded)")
    out, cases = {}, []
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for name, shape, res, tileable, seed in [
            ("time_tiled", (24, 8, 12), (4, 2, 3), (True, False, False), 7),      # the default: loops along time
            ("no_tiling", (10, 6, 6), (2, 3, 1), (False, False, False), 8),
            ("all_tiled", (12, 12, 8), (3, 4, 2), (True, True, True), 9),
            ("one_period", (16, 4, 4), (1, 1, 1), (True, False, False), 10),
        ]:
            np.random.seed(seed)
            y_ref = ref_latent.perlin_noise(shape, res, tileable)
            np.random.seed(seed)
            dims = (res[0] + 1, res[1] + 1, res[2] + 1)
            theta = 2 * np.pi * np.random.rand(*dims)
            phi = 2 * np.pi * np.random.rand(*dims)
            y_mine = signal_oracle.perlin_noise(shape, res, theta.copy(), phi.copy(), tileable)
            check("perlin_noise." + name, torch.from_numpy(np.asarray(y_mine)), y_ref.double(), tol=1e-12)
            out[f"{name}.theta"], out[f"{name}.phi"], out[f"{name}.y"] = theta, phi, y_ref.numpy()
            out[f"{name}.cfg"] = np.array(list(shape) + list(res) + [int(v) for v in tileable] + [seed], dtype=np.int64)
            cases.append(name)
    finally:
        torch.Tensor.cuda = real_cuda
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "perlin.npz"), **out)


def mapping_fixture(ref_sg2, seeding, so):
    """The z-input side of the reference Generator (models/stylegan2.py:388-393,511-526): mapping network outputs and
    one- / two-z forwards (style mixing at a fixed inject_index) of a seeded 32^2 generator."""
    print("mapping network / input_is_latent=False")
    size = 32
    sd = seeding.seeded_state_dict(size, seed=3)
    g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(sd, strict=True)
    g.eval()
    g.truncation_latent = torch.zeros(1, 512)
    z = torch.from_numpy(seeding.seeded_array(4, "z", (5, 512)))
    w_ref = g.style(z)
    check("mapping_network", so.mapping_network(sd, z), w_ref, tol=1e-5)
    z1 = torch.from_numpy(seeding.seeded_array(4, "z1", (2, 512)))
    z2 = torch.from_numpy(seeding.seeded_array(4, "z2", (2, 512)))
    noise = seeding.seeded_noise(2, size, seed=9)
    ones = torch.ones(2)
    out = {"w": w_ref.numpy(), "inject_index": np.int64(3)}
    for tag, zs, idx in (("one", [z1], None), ("mix", [z1, z2], 3)):
        img_ref, lat_ref = g(list(zs), return_latents=True, inject_index=idx, truncation=ones, noise=list(noise),
                             randomize_noise=False, input_is_latent=False)
        lat_mine = so.latents_from_z(sd, zs, g.n_latent, idx)
        check(f"latents_from_z.{tag}", lat_mine, lat_ref, tol=1e-5)
        check(f"forward_from_z.{tag}", so.generator_forward(sd, lat_mine, noise), img_ref, tol=1e-4)
        out[f"{tag}.latents"], out[f"{tag}.image"] = lat_ref.numpy(), img_ref.numpy()
    np.savez_compressed(os.path.join(HERE, "mapping.npz"), **out)


def signatures_fixture(ref_sg2, ref_op, ref_signal, ref_latent, ref_gav):
    """Parameter names and (simple) defaults of the reference's public callables on the path, read with inspect from the
    imported reference: the interface the drop-in mirrors.  Data only: {qualified name: [[param, default
    repr or null], ...]}."""
    import inspect
    import json

    import render as ref_render  # noqa

    def describe(fn):
        out = []
        for name, prm in inspect.signature(fn).parameters.items():
            if name == "self":
                continue
            d = prm.default
            simple = d is None or isinstance(d, (bool, int, float, str, tuple, list, dict))
            out.append([name, None if d is inspect.Parameter.empty else (repr(d) if simple else "<object>")])
        return out

    table = {
        "generate_audiovisual.generate": describe(ref_gav.generate),
        "generate_audiovisual.get_noise_range": describe(ref_gav.get_noise_range),
        "generate_audiovisual.load_generator": describe(ref_gav.load_generator),
        "render.render": describe(ref_render.render),
        "op.upfirdn2d": describe(ref_op.upfirdn2d),
        "op.fused_leaky_relu": describe(ref_op.fused_leaky_relu),
        "op.FusedLeakyReLU.__init__": describe(ref_op.FusedLeakyReLU.__init__),
    }
    for cls in ("Generator", "ModulatedConv2d", "StyledConv", "ToRGB", "EqualLinear", "Blur", "Upsample", "NoiseInjection",
                "ConstantInput", "ManipulationLayer"):
        c = getattr(ref_sg2, cls)
        table[f"models.stylegan2.{cls}.__init__"] = describe(c.__init__)
        table[f"models.stylegan2.{cls}.forward"] = describe(c.forward)
    for name in ("onsets", "rms", "raw_chroma", "chroma", "laplacian_segmentation", "normalize", "percentile", "percentile_clip",
                 "compress", "expand", "gaussian_filter", "load_audio", "set_SMF"):
        table[f"audioreactive.signal.{name}"] = describe(getattr(ref_signal, name))
    for name in ("chroma_weight_latents", "slerp", "slerp_loops", "spline_loops", "wrapping_slice", "generate_latents",
                 "save_latents", "load_latents", "perlin_noise"):
        table[f"audioreactive.latent.{name}"] = describe(getattr(ref_latent, name))
    import audioreactive.bend as ref_bend  # noqa  (kornia is a stub module: only the constructors' signatures are read)

    for cls in ("NetworkBend", "AddNoise", "Translate", "Zoom", "Rotate"):
        table[f"audioreactive.bend.{cls}.__init__"] = describe(getattr(ref_bend, cls).__init__)
    # every ``ar.<name>`` / ``ar.<name>(...)`` attribute the reference's shipped plugins touch, and everything its
    # audioreactive package star-exports: the namespace an unmodified plugin file expects to find
    import ast

    used = set()
    ex_dir = os.path.join(REF, "audioreactive", "examples")
    for fname in sorted(os.listdir(ex_dir)):
        if fname.endswith(".py"):
            for node in ast.walk(ast.parse(open(os.path.join(ex_dir, fname)).read())):
                if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "ar":
                    used.add(node.attr)
    import audioreactive as ref_ar  # noqa

    third_party = ("scipy", "numpy", "torch", "sklearn", "librosa", "madmom", "kornia", "matplotlib", "joblib", "builtins")
    exported = sorted(n for n in dir(ref_ar) if not n.startswith("_") and callable(getattr(ref_ar, n))
                      and not str(getattr(getattr(ref_ar, n), "__module__", "")).startswith(third_party))
    table["__namespace__"] = {"used_by_example_plugins": sorted(used), "exported_callables": exported}
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print(f"signatures: {len(table)} callables")


def meta_fixture(ref_sg2):
    """Shape bookkeeping of the reference Generator constructor (models/stylegan2.py:395-470): n_latent, num_layers and
    the noise-buffer shapes after the output_size / base_res_factor resize (:461-470) that load_generator relies on for
    1920 / 1080 output (generate_audiovisual.py:37-56)."""
    print("generator constructor bookkeeping")
    rows = []
    for size, output_size, factor in [(32, 32, 1), (32, 1920, 1), (32, 1080, 1), (64, 1024, 1), (64, 64, 2), (32, 1920, 0.5),
                                      (128, 128, 1), (256, 1080, 1)]:
        g = ref_sg2.Generator(size, 512, 2, channel_multiplier=2, constant_input=True, output_size=output_size,
                              base_res_factor=factor)
        shapes = [tuple(getattr(g.noises, f"noise_{i}").shape[-2:]) for i in range(g.num_layers)]
        flat = [v for hw in shapes for v in hw]
        rows.append([size, output_size, int(round(factor * 100)), g.n_latent, g.num_layers] + flat + [0] * (40 - len(flat)))
    np.savez_compressed(os.path.join(HERE, "generator_meta.npz"), rows=np.array(rows, dtype=np.int64))
    import json

    layout = {}
    for size in (256, 1024):  # the two checkpoints of BASELINE.json's configs
        g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        layout[str(size)] = [[k, list(v.shape)] for k, v in g.state_dict().items()]
    with open(os.path.join(HERE, "state_dict_layout.json"), "w") as f:
        json.dump(layout, f)
    print("state dict layout:", {k: len(v) for k, v in layout.items()}, "tensors")


class FlipX(torch.nn.Module):
    def forward(self, x):
        return x.flip(-1)


class Gain(torch.nn.Module):
    def __init__(self, gain):
        super().__init__()
        self.gain = gain

    def forward(self, x):
        return x * self.gain


def bends_fixture(ref_sg2, seeding, so):
    """Where the reference's ManipulationLayers sit (models/stylegan2.py:297-307, ids :417-449) and its route to a 2:1 frame:
    transform_dict_list with torch-only transforms (kornia is absent) on a seeded 32^2 generator — ReplicationPad2d widening
    the constant at layer 0 (examples/tauceti.py:97-100), a horizontal flip after conv1 (id 1), a gain after the second and
    the last StyledConv (ids 3 and 7).  Noise maps are 2:1 to match."""
    print("network bends: layer ids / wide output")
    size, batch = 32, 2
    sd = seeding.seeded_state_dict(size, seed=12)
    g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(sd, strict=True)
    g.eval()
    g.truncation_latent = torch.zeros(1, 512)
    lat = seeding.seeded_latents(batch, g.n_latent, seed=13)
    noise = [torch.from_numpy(seeding.seeded_array(14, f"wn{i}", (batch, 1, r, 2 * r))) for i, r in enumerate(seeding.noise_sizes(size))]
    transforms = {0: torch.nn.ReplicationPad2d((2, 2, 0, 0)), 1: FlipX(), 3: Gain(0.5), 7: Gain(-1.25)}
    tdl = [{"layer": k, "transform": v} for k, v in transforms.items()]
    img_ref, _ = g(lat, noise=list(noise), truncation=torch.ones(batch), transform_dict_list=tdl, randomize_noise=False,
                   input_is_latent=True)
    assert tuple(img_ref.shape) == (batch, 3, size, 2 * size)
    img_mine = so.generator_forward(sd, lat, noise, bends=transforms)
    check("generator_forward(bends)", img_mine, img_ref, tol=1e-4)
    plain = so.generator_forward(sd, lat, [nz[..., : nz.shape[-2]].contiguous() for nz in noise])
    assert plain.shape[-1] == size  # (and the bends really change the picture)
    np.savez_compressed(os.path.join(HERE, "bends.npz"), image=img_ref.numpy(), layers=np.array(sorted(transforms)),
                        seeds=np.array([12, 13, 14]))


RENDER_SUB = (slice(3, None, 8), slice(5, None, 8))  # stored pixels of every frame: rows 3::8, columns 5::8


def render_fixture(ref_sg2, seeding, so):
    """The reference's own render loop (render.py:14-192) on the CPU: ``ffmpeg`` is replaced by an object that records the
    bytes written to its stdin, ``Tensor.cuda`` / ``Tensor.pin_memory`` by the identity, ``th.cuda.FloatTensor`` (the float
    truncation branch, models/stylegan2.py:538) by ``th.FloatTensor``.  Two clips of 5 frames at 512^2 (the smallest size
    render accepts), batch 2 (ragged tail): (a) checkpoint noise buffers + float truncation 1.0, (b) per-frame noise for
    the scales <= 64 px, buffers above, per-frame truncation tensor.  Stored: seeds, the 64x64 pixel subsample RENDER_SUB of
    every frame and per-frame byte sums."""
    print("render.render (reference loop, frames captured from the ffmpeg pipe)")
    captured = []

    class _Pipe:
        def write(self, data):
            captured.append(np.frombuffer(data, dtype=np.uint8).copy())

        def close(self):
            pass

    class _Chain:
        stdin = _Pipe()

        def output(self, *a, **k):
            return self

        def global_args(self, *a, **k):
            return self

        def overwrite_output(self):
            return self

        def run_async(self, **k):
            return self

        def wait(self):
            return 0

    ff = sys.modules["ffmpeg"]
    ff.input = lambda *a, **k: _Chain()
    import render as ref_render  # noqa  (/root/reference/render.py)

    real = (torch.Tensor.cuda, torch.Tensor.pin_memory, torch.cuda.FloatTensor)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    size, n, batch = 512, 5, 2
    out = {"cfg": np.array([size, n, batch, 1, 2, 3], dtype=np.int64)}  # size, frames, batch, seeds: weights, latents, noise
    try:
        sd = seeding.seeded_state_dict(size, seed=1)
        g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        g.load_state_dict(sd, strict=True)
        g.eval()
        lat = seeding.seeded_latents(n, g.n_latent, seed=2)
        per_frame = seeding.seeded_noise(n, size, seed=3)
        trunc = torch.linspace(0.6, 1.0, n)
        tl = torch.from_numpy(seeding.seeded_array(5, "truncation_latent", (1, 512)))
        for tag, noise, truncation in (("a", [None] * g.num_layers, 1.0),
                                       ("b", [nz if nz.shape[-1] <= 64 else None for nz in per_frame], trunc)):
            captured.clear()
            g.truncation_latent = tl.clone()
            ref_render.render(generator=g, latents=lat.clone(), noise=list(noise), offset=0, duration=n / 30,
                              batch_size=batch, out_size=size, output_file="unused.mp4", truncation=truncation)
            assert len(captured) == n, f"reference render delivered {len(captured)} of {n} frames"
            frames = np.stack([c.reshape(size, size, 3) for c in captured])
            mine = so.frames_to_uint8(so.generator_forward(
                sd, lat, noise, truncation=None if isinstance(truncation, float) else truncation, truncation_latent=tl))
            diff = np.abs(frames.astype(np.int16) - np.asarray(mine).astype(np.int16))
            print(f"  render.{tag}: oracle-vs-reference frames max|diff| = {diff.max()} grey levels, "
                  f"{(diff > 0).mean():.2e} of the bytes differ")
            assert diff.max() <= 1 and (diff > 0).mean() < 1e-4
            out[f"{tag}.sub"] = frames[:, RENDER_SUB[0], RENDER_SUB[1], :]
            out[f"{tag}.sums"] = frames.reshape(n, -1).sum(1).astype(np.int64)
        out["b.truncation"] = trunc.numpy()
    finally:
        torch.Tensor.cuda, torch.Tensor.pin_memory, torch.cuda.FloatTensor = real
    np.savez_compressed(os.path.join(HERE, "render_512.npz"), **out)


def variants_fixture(ref_sg2, seeding, so):
    """Generator variants off the default path: LatentInput (``--noconst``, models/stylegan2.py:281-294,409-412) and
    ``min_rgb_size`` above 4 (:553-568, no ToRGB below that resolution).  Seeded 32^2 generators, reference images."""
    print("generator variants: LatentInput (--noconst), min_rgb_size")
    size, batch = 32, 2
    out = {"seeds": np.array([21, 22, 23, 24], dtype=np.int64)}
    lat = seeding.seeded_latents(batch, 8, seed=22)
    noise = seeding.seeded_noise(batch, size, seed=23)
    trunc = torch.tensor([0.8, 1.0])
    tl = torch.from_numpy(seeding.seeded_array(24, "truncation_latent", (1, 512)))
    # (a) LatentInput
    sd = seeding.seeded_state_dict(size, seed=21, constant_input=False)
    g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=False)
    g.load_state_dict(sd, strict=True)
    g.eval()
    g.truncation_latent = tl.clone()
    img_ref, _ = g(lat, noise=list(noise), truncation=trunc, randomize_noise=False, input_is_latent=True)
    check("generator(noconst)", so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl), img_ref, tol=1e-4)
    out["noconst.image"] = img_ref.numpy()
    # (b) min_rgb_size 16
    sd = seeding.seeded_state_dict(size, seed=21)
    g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True, min_rgb_size=16)
    g.load_state_dict(sd, strict=True)
    g.eval()
    g.truncation_latent = tl.clone()
    img_ref, _ = g(lat, noise=list(noise), truncation=trunc, randomize_noise=False, input_is_latent=True)
    check("generator(min_rgb_size=16)", so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl,
                                                            min_rgb_size=16), img_ref, tol=1e-4)
    out["min_rgb16.image"] = img_ref.numpy()
    np.savez_compressed(os.path.join(HERE, "generator_variants.npz"), **out)


def plugin_fixtures(ref_sg2, ref_gav, seeding):
    """The reference's own glue around the (unavailable) audio features: its default plugin
    (audioreactive/examples/default.py:6-45) and ``generate()`` (generate_audiovisual.py:59-231) run here on the CPU with
    ``ar.onsets`` / ``ar.chroma`` / ``ar.load_audio`` replaced by the seeded stand-ins of tests/golden/plugin_stubs.py,
    ``torch.randn`` by its shape-keyed seeded generator, ``Tensor.cuda`` / ``Module.cuda`` / ``pin_memory`` by the identity,
    ``th.cuda.FloatTensor`` by ``th.FloatTensor`` and ffmpeg by a byte recorder.  Stored: what the reference produced
    (latents / noise summaries, frame subsamples); the stand-ins regenerate the inputs on the GPU box."""
    import argparse as _argparse
    import queue as _queue
    import tempfile

    sys.path.insert(0, HERE)
    import plugin_stubs as stubs

    import audioreactive as ref_ar  # noqa  (the reference package: /root/reference is first on sys.path)
    from audioreactive.examples import default as ref_plugin  # noqa
    import render as ref_render  # noqa

    real = dict(cuda=torch.Tensor.cuda, mcuda=torch.nn.Module.cuda, pin=torch.Tensor.pin_memory, ft=torch.cuda.FloatTensor,
                randn=torch.randn, onsets=ref_ar.onsets, chroma=ref_ar.chroma, load=ref_ar.load_audio, q=ref_render.queue)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor

    class PatientQueue(_queue.Queue):  # render.py:37,97 give up after 5 s without a batch: a CPU forward can take longer
        def get(self, block=True, timeout=None):
            return super().get(block, None if timeout is None else 30)

    ref_render.queue = types.SimpleNamespace(Queue=PatientQueue, Empty=_queue.Empty)
    cwd = os.getcwd()
    try:
        # ------------------------------------------------------------ (A) the plugin callbacks on their own, long clip
        print("default plugin (reference callbacks, stubbed features)")
        n_frames = 600
        feats = stubs.Features(n_frames)
        ref_ar.onsets, ref_ar.chroma = feats.onsets, feats.chroma
        ref_ar.set_SMF(1)
        torch.randn = stubs.SeededRandn(41)
        args = _argparse.Namespace(audio=np.zeros(8, np.float32), sr=22050, n_frames=n_frames, fps=30)
        args = ref_plugin.initialize(args)
        assert [c[0] for c in feats.calls] == ["onsets", "onsets"], feats.calls
        out = {"n_frames": np.int64(n_frames), "onset_calls": np.array([list(c[1:]) for c in feats.calls], dtype=np.float64)}
        selection = torch.from_numpy(seeding.seeded_array(42, "selection", (12, 16, 512)))
        lat = ref_plugin.get_latents(selection, args)
        assert tuple(lat.shape) == (n_frames, 16, 512)
        for k, v in stubs.summary(lat).items():
            out[f"latents.{k}"] = v
        sizes = [(4, 4), (8, 8), (16, 32), (32, 32), (64, 64), (128, 128), (512, 512)]
        for h, w in sizes:
            nz = ref_plugin.get_noise(h, w, 0, len(sizes), args)
            if nz is None:
                out[f"noise_{h}x{w}.none"] = np.int64(1)
                continue
            assert tuple(nz.shape) == (n_frames, 1, h, w)
            for k, v in stubs.summary(nz).items():
                out[f"noise_{h}x{w}.{k}"] = v
            print(f"  noise {h}x{w}: std {float(nz.std()):.4f}")
        out["noise_sizes"] = np.array(sizes, dtype=np.int64)
        np.savez_compressed(os.path.join(HERE, "default_plugin.npz"), **out)

        # ------------------------------------------------------------ (B) generate() end to end, 512^2, 10 frames
        print("generate() end to end (reference orchestrator + plugin + render loop, stubbed features)")
        captured = []

        class _Pipe:
            def write(self, data):
                captured.append(np.frombuffer(data, dtype=np.uint8).copy())

            def close(self):
                pass

        class _Chain:
            stdin = _Pipe()

            def output(self, *a, **k):
                return self

            def global_args(self, *a, **k):
                return self

            def overwrite_output(self):
                return self

            def run_async(self, **k):
                return self

            def wait(self):
                return 0

        sys.modules["ffmpeg"].input = lambda *a, **k: _Chain()
        size, n, batch, fps = 512, 10, 4, 30
        out = {"cfg": np.array([size, n, batch, fps, 43, 42], dtype=np.int64)}  # ..., weight seed, selection seed
        tmp = tempfile.mkdtemp(prefix="maua_golden_")
        os.chdir(tmp)
        os.makedirs("workspace")
        torch.save({"g_ema": seeding.seeded_state_dict(size, seed=43)}, "seeded512.pt")
        np.save("selection.npy", seeding.seeded_array(42, "selection", (12, 16, 512)))
        for tag, truncation in (("a", 1.0), ("b", 0.7)):
            feats = stubs.Features(n, fps)
            ref_ar.onsets, ref_ar.chroma, ref_ar.load_audio = feats.onsets, feats.chroma, feats.load_audio
            torch.randn = stubs.SeededRandn(44)
            seen = {}

            def get_latents(selection, args):
                seen["latents"] = ref_plugin.get_latents(selection, args)
                return seen["latents"]

            def get_noise(height, width, scale, num_scales, args):
                nz = ref_plugin.get_noise(height, width, scale, num_scales, args)
                seen.setdefault("noise", []).append(nz)
                return nz

            captured.clear()
            ref_gav.generate(ckpt="seeded512.pt", audio_file="clip.wav", initialize=ref_plugin.initialize,
                             get_latents=get_latents, get_noise=get_noise, latent_file="selection.npy", G_res=size,
                             out_size=size, fps=fps, batch=batch, truncation=truncation, output_file="unused.mp4")
            assert len(captured) == n, f"reference generate() delivered {len(captured)} of {n} frames"
            frames = np.stack([c.reshape(size, size, 3) for c in captured])
            out[f"{tag}.sub"] = frames[:, RENDER_SUB[0], RENDER_SUB[1], :]
            out[f"{tag}.sums"] = frames.reshape(n, -1).sum(1).astype(np.int64)
            for k, v in stubs.summary(seen["latents"]).items():
                out[f"{tag}.latents.{k}"] = v
            out[f"{tag}.noise_is_none"] = np.array([nz is None for nz in seen["noise"]])
            for i, nz in enumerate(seen["noise"]):
                if nz is not None:
                    out[f"{tag}.noise_{i}.stats"] = stubs.summary(nz)["stats"]
            assert np.array_equal(np.load("workspace/last-latents.npy"), np.load("selection.npy"))
            print(f"  generate.{tag}: truncation {truncation}, frames mean {frames.mean():.2f} std {frames.std():.2f}")
        np.savez_compressed(os.path.join(HERE, "generate_e2e.npz"), **out)
    finally:
        os.chdir(cwd)
        torch.Tensor.cuda, torch.nn.Module.cuda, torch.Tensor.pin_memory = real["cuda"], real["mcuda"], real["pin"]
        torch.cuda.FloatTensor, torch.randn = real["ft"], real["randn"]
        ref_ar.onsets, ref_ar.chroma, ref_ar.load_audio = real["onsets"], real["chroma"], real["load"]
        ref_render.queue = real["q"]


def latent_utils_fixture(ref_latent, ref_signal):
    """Host-side latent sequencing helpers (audioreactive/latent.py:29-133): slerp values, wrapping_slice index sets incl.
    its single-wrap quirk, and slerp_loops — which in the reference dies in gaussian_filter's float32 conv1d because its
    interpolant is float64; it is run here with that one cast added in front of the filter."""
    print("latent helpers: slerp / wrapping_slice / slerp_loops")
    r = rng(104)
    out = {}
    a, b = r.standard_normal(16), r.standard_normal(16)
    vals = np.array([0.0, 0.1, 0.5, 0.77, 1.0])
    out["slerp.a"], out["slerp.b"], out["slerp.vals"] = a, b, vals
    out["slerp.y"] = np.stack([ref_latent.slerp(v, a, b) for v in vals])
    out["slerp.parallel"] = np.stack([ref_latent.slerp(v, a, 2.0 * a) for v in vals])
    cases = [(10, 0, 10), (10, 7, 6), (10, 9, 10), (10, 3, 4), (1, 0, 1), (1, 0, 5), (7, 5, 16), (6, 0, 13),
             (10, 10, 4), (7, 7, 9)]  # start == n: the reference restarts at index 0 (start > n raises there)
    out["wrap.cases"] = np.array(cases, dtype=np.int64)
    for n, start, length in cases:
        out[f"wrap.{n}_{start}_{length}"] = ref_latent.wrapping_slice(torch.arange(n), start, length, return_indices=True).numpy()
    sel = r.standard_normal((4, 18, 16)).astype(np.float32)
    real_gf = ref_latent.gaussian_filter
    ref_latent.gaussian_filter = lambda x, sigma, causal=None: ref_signal.gaussian_filter(x.float(), sigma, causal)
    try:
        ref_signal.set_SMF(1)
        out["slerp_loops.sel"] = sel
        for tag, (n_frames, n_loops, smoothing, loop) in {"a": (120, 2, 1, True), "b": (100, 1, 3, False), "c": (53, 2, 1, True)}.items():
            y = ref_latent.slerp_loops(sel, n_frames, n_loops, smoothing, loop)
            out[f"slerp_loops.{tag}.cfg"] = np.array([n_frames, n_loops, smoothing, int(loop)], dtype=np.int64)
            out[f"slerp_loops.{tag}.y"] = y.numpy()[:, ::6, :]  # the layer axis is a plain repeat: keep 3 of 18
            out[f"slerp_loops.{tag}.shape"] = np.array(y.shape, dtype=np.int64)
    finally:
        ref_latent.gaussian_filter = real_gf
    np.savez_compressed(os.path.join(HERE, "latent_utils.npz"), **out)


def stylegan1_fixture(seeding):
    """StyleGAN1 (`--stylegan1`, models/stylegan1.py): (a) a seeded narrow G_synthesis(resolution=256, fmap_base=512, fmap_max=64)
    — 64..4 channels, so the CPU forward takes seconds; the 128 / 256 px blocks take the reference's fused conv_transpose2d
    branch (:83-93), the others nearest upscale + conv — with per-block noise and 14 style rows; (b) G_mapping on seeded z;
    (c) the constructor bookkeeping of G_style for a 128-px checkpoint and 1920 output (constant widened to 8 columns, one noise
    buffer per block, state-dict keys).  Stored: reference outputs, shapes, the key list."""
    import json
    import tempfile

    import models.stylegan1 as ref_sg1  # noqa  (/root/reference)
    from oracle import stylegan1_oracle as s1o

    print("StyleGAN1: G_synthesis / G_mapping / G_style bookkeeping")
    out = {}
    torch.manual_seed(0)
    gs = ref_sg1.G_synthesis(resolution=256, fmap_base=512, fmap_max=64)
    sd = {}
    for key, v in gs.state_dict().items():
        if key.endswith("intermediate.kernel"):
            sd[key] = v.clone()
        elif key.endswith("noise.weight"):
            sd[key] = torch.from_numpy(seeding.seeded_array(31, key, tuple(v.shape), std=0.3))
        elif key.endswith(".bias"):
            sd[key] = torch.from_numpy(seeding.seeded_array(31, key, tuple(v.shape), std=0.2))
        else:
            sd[key] = torch.from_numpy(seeding.seeded_array(31, key, tuple(v.shape)))
    gs.load_state_dict(sd, strict=True)
    gs.eval()
    n_blocks = len(gs.blocks)
    dl = torch.from_numpy(seeding.seeded_array(32, "dlatents", (2, 2 * n_blocks, 512)))
    noise = [torch.from_numpy(seeding.seeded_array(33, f"noise_{i}", (2 if i % 2 else 1, 1, 4 * 2 ** i, 4 * 2 ** i))) for i in range(n_blocks)]
    x = None
    for i, blk in enumerate(gs.blocks.values()):  # G_style.forward's loop (:600-605): one noise tensor per block
        x = blk(dl[:, 2 * i: 2 * i + 2], noise=noise[i]) if i == 0 else blk(x, dl[:, 2 * i: 2 * i + 2], noise=noise[i])
    img_ref = gs.torgb(x)
    img_mine = s1o.synthesis(sd, dl, noise, prefix="")
    check("stylegan1 G_synthesis(256, narrow)", img_mine, img_ref, tol=2e-4)
    out["synth.image"] = img_ref.numpy()
    out["synth.seeds"] = np.array([31, 32, 33], dtype=np.int64)
    out["synth.keys"] = np.array(list(sd.keys()))
    out["synth.shapes"] = np.array([";".join(str(d) for d in v.shape) for v in sd.values()])
    # (b) mapping network
    gm = ref_sg1.G_mapping()
    msd = {k: torch.from_numpy(seeding.seeded_array(34, k, tuple(v.shape), std=0.2 if k.endswith("bias") else 1.0))
           for k, v in gm.state_dict().items()}
    gm.load_state_dict(msd, strict=True)
    z = torch.from_numpy(seeding.seeded_array(35, "z", (3, 512)))
    w_ref = gm(z)
    check("stylegan1 G_mapping", s1o.mapping({f"g_mapping.{k}": v for k, v in msd.items()}, z), w_ref, tol=1e-5)
    out["mapping.w"] = w_ref[:, 0].numpy()
    tl = torch.from_numpy(seeding.seeded_array(36, "tl", (1, 18, 512)))
    st = torch.from_numpy(seeding.seeded_array(36, "styles", (2, 18, 512)))
    interp = torch.lerp(tl, st, 0.7)
    out["trunc.y"] = torch.where((torch.arange(18) < 8).view(1, -1, 1), interp, st).numpy()[:, ::3, ::64]
    check("stylegan1 truncation", s1o.truncate(st, tl, 0.7), torch.where((torch.arange(18) < 8).view(1, -1, 1), interp, st), tol=0)
    # (c) G_style constructor bookkeeping: 128-px checkpoint, 1920 output
    tmp = tempfile.mkdtemp(prefix="maua_sg1_")
    small = ref_sg1.G_style.__new__(ref_sg1.G_style)
    torch.nn.Sequential.__init__(small)
    small.g_mapping = ref_sg1.G_mapping()
    small.g_synthesis = ref_sg1.G_synthesis(resolution=128)
    ckpt = os.path.join(tmp, "sg1_128.pt")
    torch.save(small.state_dict(), ckpt)
    del small
    g = ref_sg1.G_style(output_size=1920, checkpoint=ckpt)
    meta = {"const": list(getattr(g.g_synthesis.blocks, "4x4").const.shape), "blocks": list(g.g_synthesis.blocks.keys()),
            "noise": [list(getattr(g, f"noise_{i}").shape) for i in range(len(g.g_synthesis.blocks))],
            "truncation_latent": list(g.truncation_latent.shape), "keys": list(g.state_dict().keys())}
    with open(os.path.join(HERE, "stylegan1_meta.json"), "w") as f:
        json.dump(meta, f)
    print("  G_style(1920, 128-px checkpoint): const", meta["const"], "noise", meta["noise"][0], "...", meta["noise"][-1])
    np.savez_compressed(os.path.join(HERE, "stylegan1.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also run the 256^2 / 1024^2 generators (minutes on CPU)")
    ap.add_argument("--only-perlin", action="store_true", help="(re)generate perlin.npz only")
    ap.add_argument("--only-render", action="store_true", help="(re)generate render_512.npz only")
    ap.add_argument("--only-mapping", action="store_true", help="(re)generate mapping.npz only")
    ap.add_argument("--only-bends", action="store_true", help="(re)generate bends.npz only")
    ap.add_argument("--only-meta", action="store_true", help="(re)generate generator_meta.npz only")
    ap.add_argument("--only-signatures", action="store_true", help="(re)generate signatures.json only")
    ap.add_argument("--only-variants", action="store_true", help="(re)generate generator_variants.npz only")
    ap.add_argument("--only-latent-utils", action="store_true", help="(re)generate latent_utils.npz only")
    ap.add_argument("--only-stylegan1", action="store_true", help="(re)generate stylegan1.npz / stylegan1_meta.json only")
    ap.add_argument("--only-plugin", action="store_true", help="(re)generate default_plugin.npz / generate_e2e.npz only")
    args = ap.parse_args()

    ref_sg2, ref_op, ref_signal, ref_latent, ref_gav = import_reference()
    from maua_stylegan2_amd import seeding
    from oracle import ops_oracle, signal_oracle, stylegan2_oracle as so

    if args.only_perlin:
        perlin_fixture(ref_latent, signal_oracle)
        return
    if args.only_render:
        render_fixture(ref_sg2, seeding, so)
        return
    if args.only_mapping:
        mapping_fixture(ref_sg2, seeding, so)
        return
    if args.only_bends:
        bends_fixture(ref_sg2, seeding, so)
        return
    if args.only_meta:
        meta_fixture(ref_sg2)
        return
    if args.only_signatures:
        signatures_fixture(ref_sg2, ref_op, ref_signal, ref_latent, ref_gav)
        return
    if args.only_variants:
        variants_fixture(ref_sg2, seeding, so)
        return
    if args.only_latent_utils:
        latent_utils_fixture(ref_latent, ref_signal)
        return
    if args.only_stylegan1:
        stylegan1_fixture(seeding)
        return
    if args.only_plugin:
        plugin_fixtures(ref_sg2, ref_gav, seeding)
        return

    # ------------------------------------------------------------------ (1) upfirdn2d
    print("upfirdn2d")
    cases = []
    r = rng(100)
    blur = seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)
    plain = seeding.fir_kernel_2d((1, 3, 3, 1), 1.0)
    specs = [
        # name, shape, kernel, up, down, pad
        ("blur_after_upconv", (2, 5, 17, 17), blur, 1, 1, (1, 1)),
        ("blur_rect", (1, 3, 33, 65), blur, 1, 1, (1, 1)),
        ("skip_upsample", (2, 3, 8, 8), blur, 2, 1, (2, 1)),
        ("skip_upsample_rect", (1, 3, 5, 12), blur, 2, 1, (2, 1)),
        ("downsample2", (2, 4, 16, 16), plain, 1, 2, (1, 1)),
        ("blur_pad22", (1, 2, 9, 9), plain, 1, 1, (2, 2)),
        ("asym4x4", (2, 3, 11, 13), r.standard_normal((4, 4)).astype(np.float32), 1, 1, (1, 1)),
        ("asym3x3", (1, 4, 10, 7), r.standard_normal((3, 3)).astype(np.float32), 1, 1, (1, 1)),
        ("asym3x3_up2", (1, 2, 6, 6), r.standard_normal((3, 3)).astype(np.float32), 2, 1, (1, 1)),
        ("asym4x4_up2_down2", (1, 2, 7, 9), r.standard_normal((4, 4)).astype(np.float32), 2, 2, (2, 1)),
        ("negative_pad", (1, 2, 12, 12), r.standard_normal((4, 4)).astype(np.float32), 1, 1, (-1, -2)),
        ("mixed_pad", (1, 2, 12, 12), r.standard_normal((4, 4)).astype(np.float32), 2, 1, (3, -1)),
        ("asym2x2", (1, 2, 6, 8), r.standard_normal((2, 2)).astype(np.float32), 2, 1, (1, 0)),
        ("big5x5_generic", (1, 2, 9, 9), r.standard_normal((5, 5)).astype(np.float32), 1, 1, (2, 2)),
        ("up3_generic", (1, 2, 5, 5), r.standard_normal((4, 4)).astype(np.float32), 3, 2, (2, 2)),
    ]
    out = {}
    for name, shape, k, up, down, pad in specs:
        x = r.standard_normal(shape).astype(np.float32)
        y_ref = ref_op.upfirdn2d(t(x), t(k), up=up, down=down, pad=pad)
        y_mine = ops_oracle.upfirdn2d(t(x), t(k), up=up, down=down, pad=pad)
        check(name, y_mine, y_ref)
        y_loops = ops_oracle.upfirdn2d_loops(x, k, up, down, pad)
        assert np.abs(y_loops - y_ref.numpy()).max() < 1e-4, name
        out[f"{name}.x"], out[f"{name}.k"], out[f"{name}.y"] = x, k, y_ref.numpy()
        out[f"{name}.cfg"] = np.array([up, down, pad[0], pad[1]], dtype=np.int64)
        cases.append(name)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(HERE, "ops_upfirdn2d.npz"), **out)

    # ------------------------------------------------------------------ (2) fused_leaky_relu
    print("fused_leaky_relu")
    out = {}
    r = rng(101)
    for name, shape in [("nchw", (2, 7, 5, 5)), ("nc", (4, 16)), ("nchw_odd", (3, 5, 3, 7)), ("ncl", (2, 6, 10))]:
        x = r.standard_normal(shape).astype(np.float32)
        b = r.standard_normal(shape[1]).astype(np.float32)
        y_ref = ref_op.fused_leaky_relu(t(x), t(b))
        check(name, ops_oracle.fused_leaky_relu(t(x), t(b)), y_ref)
        k_sem = ops_oracle.fused_bias_act_kernel_semantics(x, b, None, 3, 0, 0.2, 2 ** 0.5)
        assert np.abs(k_sem - y_ref.numpy()).max() < 1e-6
        out[f"{name}.x"], out[f"{name}.b"], out[f"{name}.y"] = x, b, y_ref.numpy()
    out["cases"] = np.array(["nchw", "nc", "nchw_odd", "ncl"])
    np.savez_compressed(os.path.join(HERE, "ops_fused_leaky_relu.npz"), **out)

    # ------------------------------------------------------------------ (3,4) layers
    print("ModulatedConv2d / StyledConv / ToRGB")
    out = {}
    r = rng(102)
    layer_cases = []
    for name, cin, cout, k, up, demod, hw in [
        ("plain3x3", 8, 6, 3, False, True, (7, 9)),
        ("up3x3", 8, 6, 3, True, True, (5, 6)),
        ("rgb1x1", 8, 3, 1, False, False, (6, 6)),
        ("plain3x3_wide", 16, 40, 3, False, True, (4, 4)),
        ("up3x3_wide", 12, 34, 3, True, True, (4, 4)),
    ]:
        m = ref_sg2.ModulatedConv2d(cin, cout, k, 32, demodulate=demod, upsample=up)
        w = r.standard_normal((1, cout, cin, k, k)).astype(np.float32)
        mw = r.standard_normal((cin, 32)).astype(np.float32)
        mb = (1 + 0.1 * r.standard_normal(cin)).astype(np.float32)
        m.weight.copy_(t(w)), m.modulation.weight.copy_(t(mw)), m.modulation.bias.copy_(t(mb))
        x = r.standard_normal((2, cin) + hw).astype(np.float32)
        s = r.standard_normal((2, 32)).astype(np.float32)
        y_ref = m(t(x), t(s))
        y_mine = so.modulated_conv2d(t(x), t(s), t(w), t(mw), t(mb), demodulate=demod, upsample=up,
                                     blur_kernel=t(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)))
        check("modconv." + name, y_mine, y_ref)
        for key, val in dict(x=x, s=s, w=w, mw=mw, mb=mb, y=y_ref.numpy()).items():
            out[f"modconv.{name}.{key}"] = val
        out[f"modconv.{name}.cfg"] = np.array([cin, cout, k, int(up), int(demod)], dtype=np.int64)
        layer_cases.append(name)
    out["modconv.cases"] = np.array(layer_cases)

    for name, up in [("styled_plain", False), ("styled_up", True)]:
        cin, cout = 8, 10
        m = ref_sg2.StyledConv(cin, cout, 3, 32, upsample=up)
        sd = {
            "L.conv.weight": r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32),
            "L.conv.modulation.weight": r.standard_normal((cin, 32)).astype(np.float32),
            "L.conv.modulation.bias": (1 + 0.1 * r.standard_normal(cin)).astype(np.float32),
            "L.noise.weight": np.array([0.37], dtype=np.float32),
            "L.activate.bias": (0.3 * r.standard_normal(cout)).astype(np.float32),
        }
        if up:
            sd["L.conv.blur.kernel"] = seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)
        m.load_state_dict({k[2:]: t(v) for k, v in sd.items()}, strict=True)
        h = 6
        x = r.standard_normal((2, cin, h, h)).astype(np.float32)
        s = r.standard_normal((2, 32)).astype(np.float32)
        oh = 2 * h if up else h
        nz = r.standard_normal((2, 1, oh, oh)).astype(np.float32)
        y_ref = m(t(x), t(s), noise=t(nz))
        y_mine = so.styled_conv({k: t(v) for k, v in sd.items()}, "L", t(x), t(s), t(nz), up)
        check(name, y_mine, y_ref)
        for k, v in sd.items():
            out[f"{name}.sd.{k}"] = v
        out[f"{name}.x"], out[f"{name}.s"], out[f"{name}.noise"], out[f"{name}.y"] = x, s, nz, y_ref.numpy()

    for name, skip in [("torgb_noskip", False), ("torgb_skip", True)]:
        cin = 8
        m = ref_sg2.ToRGB(cin, 32, upsample=True)
        sd = {
            "L.bias": (0.3 * r.standard_normal((1, 3, 1, 1))).astype(np.float32),
            "L.upsample.kernel": seeding.fir_kernel_2d((1, 3, 3, 1), 4.0),
            "L.conv.weight": r.standard_normal((1, 3, cin, 1, 1)).astype(np.float32),
            "L.conv.modulation.weight": r.standard_normal((cin, 32)).astype(np.float32),
            "L.conv.modulation.bias": (1 + 0.1 * r.standard_normal(cin)).astype(np.float32),
        }
        m.load_state_dict({k[2:]: t(v) for k, v in sd.items()}, strict=True)
        x = r.standard_normal((2, cin, 8, 8)).astype(np.float32)
        s = r.standard_normal((2, 32)).astype(np.float32)
        sk = r.standard_normal((2, 3, 4, 4)).astype(np.float32) if skip else None
        y_ref = m(t(x), t(s), t(sk) if skip else None)
        y_mine = so.to_rgb({k: t(v) for k, v in sd.items()}, "L", t(x), t(s), t(sk) if skip else None)
        check(name, y_mine, y_ref)
        for k, v in sd.items():
            out[f"{name}.sd.{k}"] = v
        out[f"{name}.x"], out[f"{name}.s"], out[f"{name}.y"] = x, s, y_ref.numpy()
        if skip:
            out[f"{name}.skip"] = sk
    np.savez_compressed(os.path.join(HERE, "layers.npz"), **out)

    # ------------------------------------------------------------------ (5) end-to-end generators
    print("Generator end-to-end")
    sizes = [(8, 2, 1), (16, 2, 1), (64, 2, 1)]
    if args.full:
        sizes += [(256, 2, 4), (1024, 1, 16)]
    for size, batch, stride in sizes:
        path = os.path.join(HERE, f"gen_{size}.npz")
        sd = seeding.seeded_state_dict(size, seed=0)
        g = ref_sg2.Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        g.load_state_dict(sd, strict=True)  # proves key names/shapes equal the reference checkpoint layout
        g.eval()
        n_latent = g.n_latent
        lat = seeding.seeded_latents(batch, n_latent, seed=1)
        noise = seeding.seeded_noise(batch, size, seed=2)
        trunc = torch.full((batch,), 0.7)
        tl = torch.from_numpy(seeding.seeded_array(5, "truncation_latent", (1, 512)))
        g.truncation_latent = tl
        img_ref, acts_ref = g(styles=lat, noise=list(noise), truncation=trunc, randomize_noise=False,
                              input_is_latent=True, return_activation_maps=True)
        img_mine, acts_mine = so.generator_forward(sd, lat, noise, truncation=trunc, truncation_latent=tl,
                                                   return_activations=True)
        check(f"generator {size} image", img_mine, img_ref, tol=2e-4)
        for a, bb in zip(acts_mine, acts_ref):
            assert float((a - bb).abs().max()) < 2e-4
        # second run: checkpoint noise buffers (noise=None), truncation 1 -> identity lerp
        g.truncation_latent = torch.zeros(1, 512)
        img_ref2, _ = g(styles=lat, noise=None, truncation=torch.ones(batch), randomize_noise=False, input_is_latent=True)
        img_mine2 = so.generator_forward(sd, lat, None)
        check(f"generator {size} image (buffer noise)", img_mine2, img_ref2, tol=2e-4)
        fix = {
            "size": np.int64(size), "batch": np.int64(batch), "stride": np.int64(stride),
            "seeds": np.array([0, 1, 2, 5], dtype=np.int64), "truncation": np.float32(0.7),
            "image": img_ref.numpy()[:, :, ::stride, ::stride].copy(),
            "image_buffer_noise": img_ref2.numpy()[:, :, ::stride, ::stride].copy(),
            "image_mean_std": np.array([img_ref.mean().item(), img_ref.std().item()], dtype=np.float64),
            "act_mean_abs": np.array([a.abs().mean().item() for a in acts_ref], dtype=np.float64),
            "act_sum": np.array([a.double().sum().item() for a in acts_ref], dtype=np.float64),
        }
        np.savez_compressed(path, **fix)
        print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")
        del g, sd

    # ------------------------------------------------------------------ (6) audioreactive torch stages
    print("audioreactive: gaussian_filter / percentile_clip / chroma_weight_latents / get_noise_range")
    out = {}
    r = rng(103)
    gcases = []
    for name, shape, sigma, causal, smf in [
        ("env_s5", (200,), 5, 0, 1.0),
        ("env_s2_c02", (150,), 2, 0.2, 1.0),
        ("lat_s4", (120, 3, 16), 4, None, 1.0),
        ("lat_s2_c02_smf2", (90, 2, 8), 2, 0.2, 2.0),
        ("noise_s5", (64, 1, 4, 6), 5, None, 1.0),
        ("short_radius_gt_n", (10, 2, 4), 5, None, 1.0),  # radius 20 > n_frames 10 -> :350-355 branch
        ("causal_int", (80,), 3, 1, 1.0),  # non-float causal -> right half zeroed (:340-341)
    ]:
        x = r.standard_normal(shape).astype(np.float32)
        ref_signal.set_SMF(smf)
        y_ref = ref_signal.gaussian_filter(t(x), sigma, causal=causal)
        y_mine = signal_oracle.gaussian_filter(t(x), sigma, causal=causal, smf=smf)
        check("gaussian_filter." + name, y_mine, y_ref)
        out[f"gf.{name}.x"], out[f"gf.{name}.y"] = x, y_ref.numpy()
        out[f"gf.{name}.cfg"] = np.array([sigma, -1.0 if causal is None else float(causal), smf,
                                          0.0 if causal is None else (2.0 if isinstance(causal, float) else 1.0)])
        gcases.append(name)
    ref_signal.set_SMF(1)
    out["gf.cases"] = np.array(gcases)
    for name, n, p in [("p97", 300, 97), ("p50", 257, 50), ("p100", 100, 100)]:
        x = np.abs(r.standard_normal(n)).astype(np.float32)
        y_ref = ref_signal.percentile_clip(t(x).clone(), p)
        check("percentile_clip." + name, signal_oracle.percentile_clip(t(x).clone(), p), y_ref)
        out[f"pc.{name}.x"], out[f"pc.{name}.y"], out[f"pc.{name}.p"] = x, y_ref.numpy(), np.int64(p)
    out["pc.cases"] = np.array(["p97", "p50", "p100"])
    x = r.standard_normal(64).astype(np.float32)
    out["normalize.x"], out["normalize.y"] = x, ref_signal.normalize(t(x).clone()).numpy()
    check("normalize", signal_oracle.normalize(t(x).clone()), t(out["normalize.y"]))
    out["compress.x"] = x
    out["compress.y"] = ref_signal.compress(t(x).clone(), 0.5, 0.25).numpy()
    check("compress", signal_oracle.compress(t(x).clone(), 0.5, 0.25), t(out["compress.y"]))
    chroma = np.abs(r.standard_normal((40, 12))).astype(np.float32)
    chroma /= chroma.sum(1, keepdims=True)
    lats = r.standard_normal((12, 6, 16)).astype(np.float32)
    y_ref = ref_latent.chroma_weight_latents(t(chroma), t(lats))
    check("chroma_weight_latents", signal_oracle.chroma_weight_latents(t(chroma), t(lats)), y_ref)
    out["cwl.chroma"], out["cwl.latents"], out["cwl.y"] = chroma, lats, y_ref.numpy()
    rows = []
    for out_size, g_res in [(1024, 1024), (256, 256), (512, 512), (1920, 1024), (1080, 1024), (512, 256), (1024, 512)]:
        lo, hi, fn = ref_gav.get_noise_range(out_size, g_res, False)
        sides = [2 ** fn(s) for s in range(lo, hi)]
        assert signal_oracle.noise_side_lengths(out_size, g_res) == sides, (out_size, g_res)
        rows.append([out_size, g_res, lo, hi] + sides + [0] * (20 - len(sides)))
    out["noise_range"] = np.array(rows, dtype=np.int64)
    wl = ref_latent.wrapping_slice(torch.arange(10), 7, 6)
    out["wrapping_slice_10_7_6"] = wl.numpy()
    sel = r.standard_normal((5, 3, 4)).astype(np.float32)
    out["spline.sel"] = sel
    out["spline.y"] = ref_latent.spline_loops(sel, 37, 2).numpy()
    np.savez_compressed(os.path.join(HERE, "audioreactive_torch.npz"), **out)

    # ------------------------------------------------------------------ (7) uint8 post-process (render.py:40-43)
    print("frame post-process")
    edge = np.array([-1.0, 1.0, -1.01, 1.01, 0.0, 0.5, -0.5, 0.999999, -0.999999, 1 / 127.5 - 1, 0.0039, 0.33333334,
                     -0.00392157, 0.00392157, 0.9921569, 0.9960784], dtype=np.float32)
    vals = np.concatenate([edge, r.uniform(-1.2, 1.2, 3 * 4 * 8 - edge.size).astype(np.float32)]).reshape(1, 3, 4, 8)
    imgs = (t(vals).clone().clamp_(-1, 1) + 1) * 127.5
    u8_ref = imgs.permute(0, 2, 3, 1).numpy().astype(np.uint8)
    assert (so.frames_to_uint8(t(vals)) == u8_ref).all()
    np.savez_compressed(os.path.join(HERE, "postprocess.npz"), x=vals, y=u8_ref)

    # ------------------------------------------------------------------ (8) Perlin noise
    perlin_fixture(ref_latent, signal_oracle)

    # ------------------------------------------------------------------ (9) mapping network, z inputs
    mapping_fixture(ref_sg2, seeding, so)

    # ------------------------------------------------------------------ (9b) interface signatures
    signatures_fixture(ref_sg2, ref_op, ref_signal, ref_latent, ref_gav)

    # ------------------------------------------------------------------ (10) constructor bookkeeping
    meta_fixture(ref_sg2)

    # ------------------------------------------------------------------ (10b) network bends
    bends_fixture(ref_sg2, seeding, so)

    # ------------------------------------------------------------------ (11) render loop
    render_fixture(ref_sg2, seeding, so)

    # ------------------------------------------------------------------ (12) generator variants
    variants_fixture(ref_sg2, seeding, so)

    # ------------------------------------------------------------------ (12b) latent sequencing helpers
    latent_utils_fixture(ref_latent, ref_signal)

    # ------------------------------------------------------------------ (12c) StyleGAN1
    stylegan1_fixture(seeding)

    # ------------------------------------------------------------------ (13) default plugin + generate() end to end
    plugin_fixtures(ref_sg2, ref_gav, seeding)
    print("done")


if __name__ == "__main__":
    main()
