"""GPU parity, end to end: the HIP generator vs the reference's outputs (golden fixtures, all sizes up to 1024^2)
and vs the oracle on fresh seeded inputs; hipGraph replay equals eager."""
import numpy as np
import pytest
import torch

from maua_stylegan2_amd import seeding

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1e-3  # north_star: match the reference CPU fallback within 1e-3 fp32


def build(size, dev, seed=0):
    from maua_stylegan2_amd.models.stylegan2 import Generator

    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(size, seed=seed), strict=True)
    return g.to(dev).eval()


@pytest.mark.parametrize("size", [8, 16, 64, 256, 1024])
def test_generator_matches_reference_golden(gpu, golden, size):
    gold = golden(f"gen_{size}.npz")
    batch, stride = int(gold["batch"]), int(gold["stride"])
    s_sd, s_lat, s_noise, s_tl = (int(v) for v in gold["seeds"])
    g = build(size, gpu, s_sd)
    lat = seeding.seeded_latents(batch, g.n_latent, seed=s_lat).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(batch, size, seed=s_noise)]
    g.truncation_latent = torch.from_numpy(seeding.seeded_array(s_tl, "truncation_latent", (1, 512))).to(gpu)
    img, acts = g(styles=lat, noise=noise, truncation=torch.full((batch,), float(gold["truncation"]), device=gpu),
                  randomize_noise=False, input_is_latent=True, return_activation_maps=True)
    assert img.shape == (batch, 3, size, size)
    got = img.cpu().numpy()[:, :, ::stride, ::stride]
    err = np.abs(got - gold["image"]).max()
    assert err < TOL, f"{size}: max abs err {err}"
    np.testing.assert_allclose([a.abs().mean().item() for a in acts], gold["act_mean_abs"], rtol=1e-3)
    np.testing.assert_allclose(np.array([img.mean().item(), img.std().item()]), gold["image_mean_std"], atol=1e-3)
    # checkpoint noise buffers + truncation 1 (identity)
    g.truncation_latent = None
    img2, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
    err2 = np.abs(img2.cpu().numpy()[:, :, ::stride, ::stride] - gold["image_buffer_noise"]).max()
    assert err2 < TOL, f"{size}: max abs err {err2}"


def test_generator_256_batch8_vs_oracle(gpu):
    """BASELINE config 1 shape (256^2, batch 8) on fresh seeds against the oracle, full frames."""
    from oracle import stylegan2_oracle as so

    sd = seeding.seeded_state_dict(256, seed=3)
    g = build(256, gpu, 3)
    lat = seeding.seeded_latents(8, g.n_latent, seed=4)
    noise = seeding.seeded_noise(8, 256, seed=5)
    want = so.generator_forward(sd, lat, noise)
    got, _ = g(styles=lat.to(gpu), noise=[n.to(gpu) for n in noise], truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert float((got.cpu() - want).abs().max()) < TOL
    u8 = so.frames_to_uint8(want)
    from maua_stylegan2_amd import _lib
    out = torch.empty((8, 256, 256, 3), dtype=torch.uint8, device=gpu)
    _lib.check(_lib.load().maua_frames_to_u8(got.data_ptr(), out.data_ptr(), 8, 256, 256, _lib.stream_ptr()), "u8")
    diff = np.abs(out.cpu().numpy().astype(np.int16) - u8.astype(np.int16))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3  # 1e-3 float noise can flip a truncating cast by one level


def test_hipgraph_replay_equals_eager(gpu):
    g = build(64, gpu, 1)
    batch = 2
    lat = seeding.seeded_latents(batch, g.n_latent, seed=9).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(batch, 64, seed=10)]
    eager, _ = g(styles=lat, noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
    eager = eager.clone()
    lat2 = seeding.seeded_latents(batch, g.n_latent, seed=11).to(gpu)
    # the captured forward reads its inputs through a frame source: bind a 3-batch sequence (batches 0 and 2 = lat, 1 = lat2;
    # noise maps repeated accordingly), replay at different frame offsets of the SAME graph
    seq_lat = torch.cat([lat, lat2, lat]).contiguous()
    seq_noise = [torch.cat([n, n, n]).contiguous() for n in noise]
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        lane = g.capture_graph(batch)
        lane.bind(seq_lat, seq_noise)
        lane.replay(0)
        stream.synchronize()
        assert torch.equal(lane.image, eager)
        lane.replay(batch)  # new inputs through the same graph: only the frame index moved
        stream.synchronize()
        replayed = lane.image.clone()
        lane.replay(2 * batch)
        stream.synchronize()
        assert torch.equal(lane.image, eager)
        with pytest.raises(RuntimeError, match="outside the bound sequences"):
            lane.replay(2 * batch + 1)
        # a second render through the same graph: other sequences, shared [1,1,h,w] maps and a checkpoint buffer (None) mixed
        mixed = [None if i % 3 == 0 else (n[:1].contiguous() if i % 3 == 1 else n) for i, n in enumerate(noise)]
        lane.bind(lat2, mixed)
        lane.replay(0)
        stream.synchronize()
        rebound = lane.image.clone()
        with pytest.raises(RuntimeError, match="does not match"):
            lane.bind(lat2, [n[..., :-1] for n in noise])
    eager2, _ = g(styles=lat2, noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert torch.equal(replayed, eager2)
    eager3, _ = g(styles=lat2, noise=mixed, truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert torch.equal(rebound, eager3)


def test_randomize_noise_and_float_truncation(gpu):
    g = build(16, gpu, 2)
    lat = seeding.seeded_latents(2, g.n_latent, seed=1).to(gpu)
    a, _ = g(styles=lat, truncation=0.5, randomize_noise=True, input_is_latent=True)
    a = a.clone()
    b, _ = g(styles=lat, truncation=0.5, randomize_noise=True, input_is_latent=True)
    assert a.shape == (2, 3, 16, 16) and not torch.equal(a, b)
    assert g.truncation_latent is not None and g.truncation_latent.shape == (1, 512)


def test_fused_torgb_and_winograd_match_separate_direct_kernels(gpu):
    """1024^2: the default path (Winograd for >= 64-channel plain layers, ToRGB folded into the 64- and 32-channel
    conv epilogues) against the same generator with both switched off (direct 3x3 kernel + separate ToRGB launch)."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    g = build(1024, gpu, 2)
    lat = seeding.seeded_latents(1, g.n_latent, seed=6).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(1, 1024, seed=7)]
    fast, _ = g(styles=lat, noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
    fast = fast.clone()
    keep = (ModulatedConv2d.winograd_min_cout, ModulatedConv2d.winograd43_min_cout, ModulatedConv2d.winograd2d_min_cout)
    try:
        ModulatedConv2d.winograd_min_cout = ModulatedConv2d.winograd43_min_cout = ModulatedConv2d.winograd2d_min_cout = 1 << 30
        g.disable_rgb_fusion = True
        plain, _ = g(styles=lat, noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
        plain = plain.clone()
        # ... and with the 2-D Winograd kernel on EVERY qualifying layer (incl. the fused-ToRGB 32/64-channel ones)
        ModulatedConv2d.winograd_min_cout, ModulatedConv2d.winograd43_min_cout = keep[0], keep[1]
        ModulatedConv2d.winograd2d_min_cout = 32
        g.disable_rgb_fusion = False
        all2d, _ = g(styles=lat, noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
    finally:
        ModulatedConv2d.winograd_min_cout, ModulatedConv2d.winograd43_min_cout, ModulatedConv2d.winograd2d_min_cout = keep
        g.disable_rgb_fusion = False
    assert float((fast - plain).abs().max()) < 5e-4
    assert float((all2d - plain).abs().max()) < 5e-4


def test_generator_512_channel_multiplier_1_vs_oracle(gpu):
    """A narrower generator (channel_multiplier 1: 128 / 64 / 32 / 16 channels at 64..512^2) exercises the tile configs
    below the Winograd thresholds (16-channel layers: direct mode on a 32-row tile) next to every Winograd mode."""
    from maua_stylegan2_amd.models.stylegan2 import Generator
    from oracle import stylegan2_oracle as so

    sd = seeding.seeded_state_dict(512, seed=11, channel_multiplier=1)
    g = Generator(512, 512, 8, channel_multiplier=1, constant_input=True)
    g.load_state_dict(sd, strict=True)
    g = g.to(gpu).eval()
    lat = seeding.seeded_latents(2, g.n_latent, seed=12)
    noise = seeding.seeded_noise(2, 512, seed=13)
    want = so.generator_forward(sd, lat, noise)
    got, _ = g(styles=lat.to(gpu), noise=[n.to(gpu) for n in noise], truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert float((got.cpu() - want).abs().max()) < TOL


def test_generator_z_inputs_vs_reference_golden(gpu, golden):
    """The input_is_latent=False side of Generator.forward (mapping network on the device, one z or two z mixed at
    inject_index, return_latents) against outputs of the reference generator (tests/golden/mapping.npz)."""
    from maua_stylegan2_amd.models.stylegan2 import Generator

    fx = golden("mapping.npz")
    g = Generator(32, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(32, seed=3), strict=True)
    g = g.to(gpu).eval()
    z = torch.from_numpy(seeding.seeded_array(4, "z", (5, 512))).to(gpu)
    np.testing.assert_allclose(g.get_latent(z).cpu().numpy(), fx["w"], atol=2e-4, rtol=1e-3)
    z1 = torch.from_numpy(seeding.seeded_array(4, "z1", (2, 512))).to(gpu)
    z2 = torch.from_numpy(seeding.seeded_array(4, "z2", (2, 512))).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(2, 32, seed=9)]
    for tag, zs, idx in (("one", [z1], None), ("mix", [z1, z2], int(fx["inject_index"]))):
        img, lat = g(list(zs), return_latents=True, inject_index=idx, truncation=1.0, noise=list(noise),
                     randomize_noise=False, input_is_latent=False)
        assert tuple(lat.shape) == (2, g.n_latent, 512)
        np.testing.assert_allclose(lat.cpu().numpy(), fx[f"{tag}.latents"], atol=2e-4, rtol=1e-3, err_msg=tag)
        # 2e-3: the mapping network's own fp32 rounding (device GEMM vs CPU) feeds into the generator here
        np.testing.assert_allclose(img.cpu().numpy(), fx[f"{tag}.image"], atol=2e-3, err_msg=tag)


def test_forward_outputs_are_private_copies_and_noise_is_validated(gpu):
    """Reference-style use of the public call: results of successive calls are kept side by side (the reference returns
    freshly allocated tensors), and a noise map that does not match the feature map raises (the reference fails with a
    broadcast error, models/stylegan2.py:266) instead of being read out of bounds."""
    from maua_stylegan2_amd.models.stylegan2 import Generator

    size = 16
    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(size, seed=0))
    g = g.to(gpu).eval()
    lat = seeding.seeded_latents(2, g.n_latent, seed=1).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(1, size, seed=2)]
    a, acts_a = g(styles=lat[:1], noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True,
                  return_activation_maps=True)
    keep, keep_act = a.clone(), acts_a[-1].clone()
    b, _ = g(styles=lat[1:], noise=noise, truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert torch.equal(a, keep) and torch.equal(acts_a[-1], keep_act) and not torch.equal(a, b)
    for bad in (torch.zeros(1, 1, 8, 4, device=gpu), torch.zeros(3, 1, 8, 8, device=gpu), torch.zeros(1, 2, 8, 8, device=gpu)):
        wrong = list(noise)
        wrong[1] = bad  # first 8x8 layer (the up-sampling StyledConv)
        with pytest.raises(RuntimeError, match="noise"):
            g(styles=lat[:1], noise=wrong, truncation=1.0, randomize_noise=False, input_is_latent=True)
        wrong = list(noise)
        wrong[2] = bad  # plain StyledConv
        with pytest.raises(RuntimeError, match="noise"):
            g(styles=lat[:1], noise=wrong, truncation=1.0, randomize_noise=False, input_is_latent=True)


def test_generator_variants_match_reference_golden(gpu, golden):
    """LatentInput generators (``--noconst``, reference models/stylegan2.py:281-294) and ``min_rgb_size`` (:553-568) against
    images of the reference classes (tests/golden/generator_variants.npz), eager and through a captured hipGraph."""
    from maua_stylegan2_amd.models.stylegan2 import Generator

    fx = golden("generator_variants.npz")
    s_w, s_lat, s_noise, s_tl = (int(v) for v in fx["seeds"])
    size, batch = 32, 2
    lat = seeding.seeded_latents(batch, 8, seed=s_lat).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(batch, size, seed=s_noise)]
    trunc = torch.tensor([0.8, 1.0], device=gpu)
    tl = torch.from_numpy(seeding.seeded_array(s_tl, "truncation_latent", (1, 512))).to(gpu)
    for key, kwargs in (("noconst", dict(constant_input=False)), ("min_rgb16", dict(constant_input=True, min_rgb_size=16))):
        g = Generator(size, 512, 8, channel_multiplier=2, **kwargs)
        g.load_state_dict(seeding.seeded_state_dict(size, seed=s_w, constant_input=kwargs["constant_input"]), strict=True)
        g = g.to(gpu).eval()
        g.truncation_latent = tl
        img, _ = g(styles=lat, noise=noise, truncation=trunc, randomize_noise=False, input_is_latent=True)
        err = float((img.cpu() - torch.from_numpy(fx[f"{key}.image"])).abs().max())
        assert err < TOL, (key, err)
        with pytest.raises(RuntimeError, match="non-default stream"):
            g.capture_graph(batch)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            lane = g.capture_graph(batch)
            lane.bind(lat, noise, trunc)
            lane.replay(0)
            stream.synchronize()
            assert torch.equal(lane.image, img), key


def test_stylegan1_synthesis_matches_reference_golden(gpu, golden):
    """`--stylegan1`: the mirror's G_synthesis on the HIP path (shared-weight MFMA convs, upfirdn2d upscale / blur, fused epilogue)
    against the image of the reference's G_synthesis class (stylegan1.npz: seeded narrow 256-px network, per-block noise with
    batch 1 and 2, fused-upscale branch at 128 / 256 px), then G_style's wrapper logic (default noise buffers, truncation on the
    first 8 layers, (image, None) return) against the oracle."""
    from maua_stylegan2_amd.models import stylegan1 as sg1
    from oracle import stylegan1_oracle as s1o
    from test_oracle_golden import _sg1_state_dict

    fx = golden("stylegan1.npz")
    sd = _sg1_state_dict(fx)
    s_w, s_l, s_n = (int(v) for v in fx["synth.seeds"])
    gs = sg1.G_synthesis(resolution=256, fmap_base=512, fmap_max=64)
    gs.load_state_dict(sd, strict=True)
    gs = gs.to(gpu).eval()
    n_blocks = len(gs.blocks)
    dl = torch.from_numpy(seeding.seeded_array(s_l, "dlatents", (2, 2 * n_blocks, 512)))
    noise = [torch.from_numpy(seeding.seeded_array(s_n, f"noise_{i}", (2 if i % 2 else 1, 1, 4 * 2 ** i, 4 * 2 ** i))) for i in range(n_blocks)]
    img = gs(dl.to(gpu), [n.to(gpu) for n in noise])
    err = float((img.cpu() - torch.from_numpy(fx["synth.image"])).abs().max())
    assert img.shape == (2, 3, 256, 256) and err < TOL, err
    # G_style wrapper around the same synthesis network
    g = sg1.G_style.__new__(sg1.G_style)
    torch.nn.Sequential.__init__(g)
    g.g_mapping = sg1.G_mapping()
    g.g_synthesis = gs
    for i, nz in enumerate(noise):
        g.register_buffer(f"noise_{i}", nz[:1].clone())
    g.truncation_latent = torch.from_numpy(seeding.seeded_array(36, "tl", (1, 18, 512)))
    g = g.to(gpu)
    styles = torch.from_numpy(seeding.seeded_array(37, "styles", (2, 18, 512)))
    out, none = g(styles=styles.to(gpu), noise=[None] * n_blocks, truncation=0.7, transform_dict_list=[], randomize_noise=False,
                  input_is_latent=True)
    assert none is None
    want = s1o.synthesis(sd, s1o.truncate(styles, g.truncation_latent.cpu(), 0.7), [n[:1] for n in noise], prefix="")
    assert float((out.cpu() - want).abs().max()) < TOL
    wrong = [None] * n_blocks
    wrong[2] = torch.zeros(1, 1, 8, 8, device=gpu)
    with pytest.raises(RuntimeError, match="noise"):
        g(styles=styles.to(gpu), noise=wrong)


def test_surplus_latent_rows_are_ignored_as_in_the_reference(gpu):
    """The reference forward only indexes ``latent[:, i]`` for i < n_latent (models/stylegan2.py:549-569), so an 18-layer latent
    file fed to a smaller generator works there.  Eager forward (row stride taken from the tensor) and captured lanes (sequence cut
    once per render) must give exactly what the first n_latent rows give; fewer rows than n_latent raise."""
    from maua_stylegan2_amd import render

    size, n, bs = 64, 6, 2
    g = build(size, gpu, 2)
    assert g.n_latent == 10
    lat18 = seeding.seeded_latents(n, 18, seed=31)
    lat10 = lat18[:, :10].contiguous()
    noise = seeding.seeded_noise(n, size, seed=32)
    a, _ = g(styles=lat18[:bs].to(gpu), noise=[z[:bs].to(gpu) for z in noise], randomize_noise=False, input_is_latent=True)
    b, _ = g(styles=lat10[:bs].to(gpu), noise=[z[:bs].to(gpu) for z in noise], randomize_noise=False, input_is_latent=True)
    assert torch.equal(a, b)
    _, lat_out = g(styles=lat18[:bs].to(gpu), noise=[z[:bs].to(gpu) for z in noise], randomize_noise=False, input_is_latent=True,
                   return_latents=True)
    assert tuple(lat_out.shape) == (bs, 18, 512)  # handed back as given (:571-572)

    def frames(lat):
        out = np.zeros((n, size, size, 3), np.uint8)
        for first, u8 in render.synthesize(g, lat, noise, bs, lanes=2):
            out[first: first + u8.shape[0]] = u8.cpu().numpy()
        return out

    assert np.array_equal(frames(lat18), frames(lat10))
    with pytest.raises(RuntimeError, match="do not match"):
        g(styles=lat18[:bs, :9].to(gpu), noise=None, randomize_noise=False, input_is_latent=True)


@pytest.mark.parametrize("size", [256, 1024])
def test_generator_with_split_bf16_conv_layers_matches_reference_golden(gpu, golden, size):
    """SIDE MEASUREMENT switched ON (ModulatedConv2d.split_bf16_min_cout = 128, split_bf16_up_min_cout = 32: the wide plain layers and the
    transposed layers from 32^2 inputs up run split-bf16 products on the bf16 matrix cores, csrc/modconv_sbf16.hip): the generator must
    still reproduce the REFERENCE's own image within the north_star tolerance of 1e-3; the measured error and the number of switched
    layers are printed."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    gold = golden(f"gen_{size}.npz")
    batch, stride = int(gold["batch"]), int(gold["stride"])
    s_sd, s_lat, s_noise, s_tl = (int(v) for v in gold["seeds"])
    g = build(size, gpu, s_sd)
    lat = seeding.seeded_latents(batch, g.n_latent, seed=s_lat).to(gpu)
    noise = [n.to(gpu) for n in seeding.seeded_noise(batch, size, seed=s_noise)]
    g.truncation_latent = torch.from_numpy(seeding.seeded_array(s_tl, "truncation_latent", (1, 512))).to(gpu)
    trunc = torch.full((batch,), float(gold["truncation"]), device=gpu)
    keep = ModulatedConv2d.split_bf16_min_cout, ModulatedConv2d.split_bf16_up_min_cout
    try:
        fp32, _ = g(styles=lat, noise=noise, truncation=trunc, randomize_noise=False, input_is_latent=True)
        ModulatedConv2d.split_bf16_min_cout, ModulatedConv2d.split_bf16_up_min_cout = 128, 32
        switched = sum(c.conv.conv_mode(4 * 2 ** ((i + 1) // 2) if i % 2 else 4 * 2 ** (i // 2),
                                        4 * 2 ** ((i + 1) // 2) if i % 2 else 4 * 2 ** (i // 2)) in (7, 8) for i, c in enumerate(g.convs))
        img, _ = g(styles=lat, noise=noise, truncation=trunc, randomize_noise=False, input_is_latent=True)
    finally:
        ModulatedConv2d.split_bf16_min_cout, ModulatedConv2d.split_bf16_up_min_cout = keep
    assert switched >= (5 if size == 256 else 9), switched
    err = np.abs(img.cpu().numpy()[:, :, ::stride, ::stride] - gold["image"]).max()
    err32 = np.abs(fp32.cpu().numpy()[:, :, ::stride, ::stride] - gold["image"]).max()
    print(f"[split-bf16 generator {size}] {switched} conv layers on the bf16 cores: max |image - reference| = {err:.2e} "
          f"(fp32 path: {err32:.2e}; image std {float(fp32.std()):.2f}), max |split - fp32| = {float((img - fp32).abs().max()):.2e}")
    assert err < TOL, f"{size}: max abs err {err}"


@pytest.mark.parametrize("size", [256, 1024])
def test_style_fold_on_equals_off_and_the_reference_image(gpu, golden, size):
    """Round 6, the style fold (include/maua_hip.h): every producer from the first 2-D Winograd / F(2,2)^2 consumer on stores its map
    multiplied by the next convolution's styles (reference models/stylegan2.py:220-221 reassociated).  Whole generator, fold on (the
    default) against fold off — same kernels otherwise — and against the reference's own image; the number of folded layers is checked
    so that the switch cannot silently do nothing."""
    gold = golden(f"gen_{size}.npz")
    batch, stride = int(gold["batch"]), int(gold["stride"])
    s_sd, s_lat, s_noise, _ = (int(v) for v in gold["seeds"])
    g = build(size, gpu, s_sd)
    lat = seeding.seeded_latents(batch, g.n_latent, seed=s_lat).to(gpu)
    assert g.style_fold
    on, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
    folded = [i for i, c in enumerate(g.convs) if c.posted]
    g.style_fold = False
    off, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert not any(c.posted for c in g.convs)
    g.style_fold = True
    # 1024^2: convs.4's tail (16^2 -> 32^2) is the first producer whose consumer (convs.5, 2-D Winograd) takes a pre-scaled map; every
    # layer up to convs.14 follows.  256^2: convs.4 .. convs.10.
    assert folded == list(range(4, len(g.convs) - 1)), folded
    err = float((on - off).abs().max())
    print(f"[style fold, {size}^2] {len(folded)} folded layers; fold on vs off {err:.2e} at image std {float(off.std()):.2f}")
    assert err < 1e-4
    assert np.abs(on.cpu().numpy()[:, :, ::stride, ::stride] - gold["image_buffer_noise"]).max() < TOL


def test_style_fold_steps_aside_for_bends_and_activation_maps(gpu):
    """A bend on a layer id, or return_activation_maps, reads the un-scaled map: the producer of that map must not fold (the layers
    either side still do)."""
    import torch.nn as nn

    g = build(256, gpu, 1)
    lat = seeding.seeded_latents(2, g.n_latent, seed=2).to(gpu)

    class Gain(nn.Module):
        def forward(self, x):
            return x * 1.25

    plain, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
    assert g.convs[6].posted and g.convs[7].posted
    bends = [{"layer": 8, "transform": Gain()}, {"layer": 9, "transform": Gain()}]  # layer ids 8 / 9 = convs.6 / convs.7
    bent, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True, transform_dict_list=bends)
    assert not g.convs[6].posted and not g.convs[7].posted and g.convs[5].posted and g.convs[8].posted
    g.style_fold = False
    bent_off, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True, transform_dict_list=bends)
    g.style_fold = True
    assert float((bent - bent_off).abs().max()) < 1e-4 and float((bent - plain).abs().max()) > 1e-2
    _, acts = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True, return_activation_maps=True)
    assert not any(c.posted for c in g.convs)
    g.style_fold = False
    _, acts_off = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True, return_activation_maps=True)
    for a, b in zip(acts, acts_off):
        assert torch.equal(a, b)


@pytest.mark.parametrize("batch,noise_mode", [(1, "per_frame"), (8, "per_frame"), (11, "shared"), (3, "buffer")])
def test_const_conv_equals_the_convolution_on_the_repeated_constant(gpu, batch, noise_mode):
    """conv1 on the ConstantInput as y = T s (maua_const_styledconv_f32, T = conv1's weight applied to the constant once per checkpoint)
    against the layer run as a convolution over const.repeat(batch) (lowres_fusion off: reference models/stylegan2.py:547-549 literally):
    conv1's activation map, to_rgb1's contribution to the image and the whole image; and a changed constant / weight rebuilds T.  (The
    golden-image tests of this file run with the switch on: they pin the same path against the reference's images.)"""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    g = build(16, gpu, seed=3)
    lat = seeding.seeded_latents(batch, g.n_latent, seed=5).to(gpu)
    per = [n.to(gpu) for n in seeding.seeded_noise(batch, 16, seed=7)]
    noise = per if noise_mode == "per_frame" else [n[:1] for n in per] if noise_mode == "shared" else None
    outs = {}
    try:
        for on in (True, False):
            StyledConv.lowres_fusion = on
            img, acts = g(styles=lat, noise=noise, randomize_noise=False, input_is_latent=True, return_activation_maps=True)
            assert g.conv1.last_path == ("const" if on else "plain")
            outs[on] = (img.cpu().numpy(), acts[0].cpu().numpy())
        scale = float(np.abs(outs[False][1]).max())
        np.testing.assert_allclose(outs[True][1], outs[False][1], atol=2e-6 * scale, rtol=2e-5)
        np.testing.assert_allclose(outs[True][0], outs[False][0], atol=2e-5, rtol=1e-4)
        # a new constant (in place) and a new weight (swapped Parameter, as model rewriting does) both invalidate T
        StyledConv.lowres_fusion = True
        g.input.input.mul_(1.5)
        g.conv1.conv.weight = torch.nn.Parameter(g.conv1.conv.weight.detach().flip(1).contiguous())
        img_on = g(styles=lat, noise=noise, randomize_noise=False, input_is_latent=True)[0].cpu().numpy()
        StyledConv.lowres_fusion = False
        img_off = g(styles=lat, noise=noise, randomize_noise=False, input_is_latent=True)[0].cpu().numpy()
        assert np.abs(img_off - outs[False][0]).max() > 1e-3
        np.testing.assert_allclose(img_on, img_off, atol=2e-5, rtol=1e-4)
    finally:
        StyledConv.lowres_fusion = True
