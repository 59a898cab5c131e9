"""GPU property tests (hypothesis): randomly drawn shapes / pads / factors through the C ABI against the oracle.  These sweep
the dispatch logic (tile kernels vs generic gather, Winograd / polyphase / direct conv modes, ragged tiles, split-K) far more
broadly than the hand-picked parametrisations; example counts are small so that the whole file stays under a minute."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

from oracle import ops_oracle, stylegan2_oracle as so

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
COMMON = dict(deadline=None, derandomize=True, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(max_examples=40, **COMMON)
@given(n=st.integers(1, 3), c=st.integers(1, 5), h=st.integers(1, 70), w=st.integers(1, 300), kh=st.integers(1, 5),
       kw=st.integers(1, 5), up=st.integers(1, 3), down=st.integers(1, 3), p0=st.integers(-2, 4), p1=st.integers(-2, 4),
       seed=st.integers(0, 1 << 16))
def test_upfirdn2d_random_calls_vs_oracle(gpu, n, c, h, w, kh, kw, up, down, p0, p1, seed):
    from maua_stylegan2_amd.op import upfirdn2d

    out_h = ops_oracle.upfirdn2d_out_size(h, up, down, p0, p1, kh)
    out_w = ops_oracle.upfirdn2d_out_size(w, up, down, p0, p1, kw)
    if out_h < 1 or out_w < 1 or h * up + p0 + p1 < kh or w * up + p0 + p1 < kw:
        return  # empty outputs are covered by the edge-case tests
    r = np.random.default_rng(seed)
    x = torch.from_numpy(r.standard_normal((n, c, h, w)).astype(np.float32))
    k = torch.from_numpy(r.standard_normal((kh, kw)).astype(np.float32))
    want = ops_oracle.upfirdn2d(x, k, up=up, down=down, pad=(p0, p1)).numpy()
    got = upfirdn2d(x.to(gpu), k.to(gpu), up=up, down=down, pad=(p0, p1)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-5, rtol=1e-5)


@settings(max_examples=50, **COMMON)
@given(cin=st.sampled_from([3, 8, 20, 32, 40, 64, 96, 128]), cout=st.sampled_from([3, 16, 24, 32, 40, 64, 72, 128, 160]),
       h=st.integers(3, 48), w=st.sampled_from([4, 6, 9, 16, 20, 30, 32, 36, 48, 64, 66, 68, 96, 100]),
       batch=st.integers(1, 3), up=st.booleans(), seed=st.integers(0, 1 << 16))
def test_styled_conv_random_shapes_vs_oracle(gpu, cin, cout, h, w, batch, up, seed):
    """StyledConv (modulated 3x3 conv + noise + bias + leaky ReLU, plain and up-sampling) on random layer shapes: whatever
    kernel mode / tile config the host mirror picks must agree with the reference formulation."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    if up and (h > 24 or w > 48):
        h, w = min(h, 24), min(w, 48)  # keep the CPU oracle's conv_transpose + blur quick
    r = np.random.default_rng(seed)
    m = StyledConv(cin, cout, 3, 512, upsample=up)
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.31]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    if up:
        sd["L.conv.blur.kernel"] = m.conv.blur.kernel.clone()
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    nz = torch.from_numpy(r.standard_normal((batch, 1, oh, ow)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, up).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=2e-4, err_msg=f"mode {m.conv.conv_mode(h, w)}")


@settings(max_examples=40, **COMMON)
@given(cin=st.sampled_from([4, 8, 20, 32, 64, 96, 128]), cout=st.sampled_from([32, 64, 128, 192]), hb=st.integers(1, 6),
       wb=st.integers(1, 4), batch=st.integers(1, 3), shared_noise=st.booleans(), seed=st.integers(0, 1 << 16))
def test_winograd2d_random_shapes_vs_oracle(gpu, cin, cout, hb, wb, batch, shared_noise, seed):
    """The 2-D Winograd kernel (mode 5) on random qualifying shapes — H a multiple of 8, W of 32, any Cin % 4 == 0, 32 / 64 / 128 /
    192 output channels (both tile shapes, 1..3 weight tiles), per-frame or shared noise — against the oracle."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    h, w = hb * 8, wb * 32
    r = np.random.default_rng(seed)
    m = StyledConv(cin, cout, 3, 512)
    m.conv.winograd2d_min_cout = 32
    assert m.conv.conv_mode(h, w) == 5
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.27]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((1 if shared_noise else batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=2e-4)


@settings(max_examples=25, **COMMON)
@given(cin=st.sampled_from([4, 8, 20, 64, 96, 256]), cout=st.sampled_from([32, 64, 96, 128]), hb=st.integers(1, 5),
       wb=st.integers(1, 3), batch=st.integers(1, 3), seed=st.integers(0, 1 << 16))
def test_upconv2d_random_shapes_vs_oracle(gpu, cin, cout, hb, wb, batch, seed):
    """The two-axis F(2,2) transposed-conv kernel (mode 6) on random qualifying shapes — H a multiple of 8, W of 32, Cin % 4 == 0,
    Cout % 32 == 0 — through the whole up-sampling StyledConv (blur + noise + bias + activation behind it) against the oracle."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    h, w = hb * 8, wb * 32
    r = np.random.default_rng(seed)
    m = StyledConv(cin, cout, 3, 512, upsample=True)
    assert m.conv.conv_mode(h, w) == 6
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.blur.kernel": m.conv.blur.kernel.clone(),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.27]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, 2 * h, 2 * w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, True).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=2e-4)


@settings(max_examples=30, **COMMON)
@given(cin=st.sampled_from([16, 32, 48, 80, 128, 256]), cout=st.sampled_from([32, 64, 96, 128, 256, 384]), hb=st.integers(1, 4),
       wb=st.integers(1, 3), batch=st.integers(1, 3), up=st.booleans(), shared_noise=st.booleans(), seed=st.integers(0, 1 << 16))
def test_split_bf16_random_shapes_vs_oracle(gpu, cin, cout, hb, wb, batch, up, shared_noise, seed):
    """SIDE MEASUREMENT kernels (csrc/modconv_sbf16.hip, off by default) on random qualifying shapes — Cin % 16 == 0, H a multiple of 8, W of
    32; plain (mode 7): Cout % 128 == 0; transposed (mode 8): Cout % 32 == 0 — through the whole StyledConv (tail, and blur for the
    transposed form) against the oracle, same bound as the fp32 kernels."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    if not up:
        cout = 128 * max(1, cout // 128)
    h, w = hb * 8, wb * 32
    r = np.random.default_rng(seed)
    m = StyledConv(cin, cout, 3, 512, upsample=up)
    m.conv.split_bf16_min_cout, m.conv.split_bf16_up_min_cout = 128, 32
    assert m.conv.conv_mode(h, w) == (8 if up else 7)
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.27]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    if up:
        sd["L.conv.blur.kernel"] = m.conv.blur.kernel.clone()
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    nz = torch.from_numpy(r.standard_normal((1 if shared_noise else batch, 1, oh, ow)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, up).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=2e-4)


@settings(max_examples=25, **COMMON)
@given(cin=st.sampled_from([3, 16, 32, 40, 64, 128]), h=st.integers(2, 40), w=st.sampled_from([2, 4, 6, 10, 16, 30, 32, 64, 72]),
       batch=st.integers(1, 3), skip=st.booleans(), seed=st.integers(0, 1 << 16))
def test_to_rgb_random_shapes_vs_oracle(gpu, cin, h, w, batch, skip, seed):
    """ToRGB (1x1 modulated conv without demodulation + bias + 2x FIR-upsampled skip) on random shapes."""
    from maua_stylegan2_amd.models.stylegan2 import ToRGB

    if skip and (h % 2 or w % 2):
        h, w = h + h % 2, w + w % 2
    r = np.random.default_rng(seed)
    m = ToRGB(cin, 512)
    sd = {
        "L.bias": torch.from_numpy((0.2 * r.standard_normal((1, 3, 1, 1))).astype(np.float32)),
        "L.upsample.kernel": m.upsample.kernel.clone(),
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, 3, cin, 1, 1)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    sk = torch.from_numpy(r.standard_normal((batch, 3, h // 2, w // 2)).astype(np.float32)) if skip else None
    want = so.to_rgb(sd, "L", x, s, sk).numpy()
    got = m(x.to(gpu), s.to(gpu), None if sk is None else sk.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-4, rtol=1e-4)


@settings(max_examples=25, **COMMON)
@given(shape=st.sampled_from([(1, 1, 1, 1), (2, 7, 5, 5), (4, 16), (1, 3, 17, 33), (3, 32, 8, 130), (1, 512, 4, 4), (2, 5, 1, 257)]),
       slope=st.sampled_from([0.2, 0.1]), scale=st.sampled_from([2 ** 0.5, 1.0, 0.5]), seed=st.integers(0, 1 << 16))
def test_fused_leaky_relu_random_shapes_vs_oracle(gpu, shape, slope, scale, seed):
    from maua_stylegan2_amd.op import fused_leaky_relu

    r = np.random.default_rng(seed)
    x = torch.from_numpy(r.standard_normal(shape).astype(np.float32))
    b = torch.from_numpy(r.standard_normal(shape[1]).astype(np.float32))
    # the reference's CPU path hard-codes slope 0.2 (op/fused_act.py:91); the CUDA kernel honours the argument — the
    # kernel-semantics oracle is the definition for slopes other than 0.2
    want = np.asarray(ops_oracle.fused_bias_act_kernel_semantics(x.numpy(), b.numpy(), None, act=3, grad=0, alpha=slope,
                                                                 scale=scale)).reshape(shape)
    got = fused_leaky_relu(x.to(gpu), b.to(gpu), negative_slope=slope, scale=scale).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=1e-6, rtol=1e-6)


@settings(max_examples=12, **COMMON)
@given(size=st.sampled_from([8, 16, 32, 64, 128]), cm=st.sampled_from([1, 2]), batch=st.integers(1, 3),
       trunc=st.sampled_from([1.0, 0.7, 0.3]), per_frame_noise=st.booleans(), seed=st.integers(0, 1 << 10))
def test_generator_random_configs_vs_oracle(gpu, size, cm, batch, trunc, per_frame_noise, seed):
    """Whole generator forward on random (size, channel multiplier, batch, truncation, noise source) against the oracle."""
    from maua_stylegan2_amd import seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    sd = seeding.seeded_state_dict(size, seed=seed, channel_multiplier=cm)
    g = Generator(size, 512, 8, channel_multiplier=cm, constant_input=True)
    g.load_state_dict(sd, strict=True)
    g = g.to(gpu).eval()
    lat = seeding.seeded_latents(batch, g.n_latent, seed=seed + 1)
    noise = seeding.seeded_noise(batch, size, seed=seed + 2) if per_frame_noise else None
    tl = torch.from_numpy(seeding.seeded_array(seed + 3, "truncation_latent", (1, 512)))
    want = so.generator_forward(sd, lat, noise, truncation=None if trunc == 1.0 else torch.full((batch,), trunc),
                                truncation_latent=tl)
    g.truncation_latent = tl.to(gpu)
    got, _ = g(styles=lat.to(gpu), noise=None if noise is None else [n.to(gpu) for n in noise],
               truncation=trunc if trunc == 1.0 else torch.full((batch,), trunc, device=gpu), randomize_noise=False,
               input_is_latent=True)
    assert float((got.cpu() - want).abs().max()) < 1e-3


@settings(max_examples=40, **COMMON)
@given(n=st.integers(2, 300), trail=st.sampled_from([(), (3,), (2, 5), (1, 4, 6)]), sigma=st.sampled_from([0.5, 1, 2, 4.5, 5, 17, 64, 128]),
       causal=st.sampled_from([None, 0, 1, 0.2, 0.75]), smf=st.sampled_from([1.0, 0.8, 2.0]), seed=st.integers(0, 1 << 16))
def test_gaussian_filter_random_vs_oracle(gpu, n, trail, sigma, causal, smf, seed):
    """Temporal Gaussian FIR (audioreactive/signal.py:319-368): every (length, trailing shape, sigma, causal factor, SMF) draw
    incl. radius > n_frames (the reference's wrap-around padding) and integer `causal` (which zeroes the future half)."""
    from maua_stylegan2_amd.audioreactive import signal as sig
    from oracle import signal_oracle

    r = np.random.default_rng(seed)
    x = torch.from_numpy(r.standard_normal((n,) + trail).astype(np.float32))
    want = signal_oracle.gaussian_filter(x, sigma, causal=causal, smf=smf).numpy()
    sig.set_SMF(smf)
    try:
        got = sig.gaussian_filter(x.to(gpu), sigma, causal=causal).cpu().numpy()
    finally:
        sig.set_SMF(1)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=3e-5, rtol=1e-5)


@settings(max_examples=25, **COMMON)
@given(batch=st.integers(1, 3), h=st.integers(1, 40), w=st.integers(1, 70), seed=st.integers(0, 1 << 16))
def test_frames_to_uint8_random_shapes_bit_exact(gpu, batch, h, w, seed):
    """The uint8 NHWC frame epilogue (render.py:40-43) is integer work: bit-exact for every width (packed 4-pixel path and
    the scalar path), values straddling the clamp and exact half-levels included."""
    from maua_stylegan2_amd import render

    r = np.random.default_rng(seed)
    x = (1.3 * r.standard_normal((batch, 3, h, w))).astype(np.float32)
    x.flat[:: 7] = np.round(x.flat[:: 7] * 127.5) / 127.5  # exact grey levels
    x.flat[:: 11] = np.float32(1.0)
    x.flat[:: 13] = np.float32(-1.0)
    want = so.frames_to_uint8(torch.from_numpy(x))
    got = render.frames_to_uint8(torch.from_numpy(x).to(gpu)).cpu().numpy()
    assert np.array_equal(got, want)


@settings(max_examples=25, **COMMON)
@given(b=st.integers(1, 3), c=st.integers(1, 5), h=st.integers(2, 20), w=st.integers(2, 24), kind=st.sampled_from(["zoom", "rotate"]),
       seed=st.integers(0, 1 << 16))
def test_bends_random_vs_oracle(gpu, b, c, h, w, kind, seed):
    """Zoom / Rotate bends (ReflectionPad2d -> affine warp -> CenterCrop, audioreactive/bend.py:52-102) with random
    per-frame parameters against the oracle's composition."""
    from maua_stylegan2_amd.audioreactive import bend
    from oracle import signal_oracle

    r = np.random.default_rng(seed)
    x = r.standard_normal((b, c, h, w)).astype(np.float32)
    xd = torch.from_numpy(x).to(gpu)
    if kind == "zoom":
        z = torch.from_numpy(r.uniform(0.4, 2.5, b).astype(np.float32))
        pad = max(h, w) - 1
        want = signal_oracle.affine_reflect_warp(x, bend._inverse_maps_scale(z, w + 2 * pad, h + 2 * pad).numpy(), (pad,) * 4)
        got = bend.Zoom(z, h, w)(xd).cpu().numpy()
    else:
        a = torch.from_numpy(r.uniform(-180, 180, b).astype(np.float32))
        pad = int(max(h, w) * (1 - np.sqrt(2) / 2))
        want = signal_oracle.affine_reflect_warp(x, bend._inverse_maps_rotate(a, w + 2 * pad, h + 2 * pad).numpy(), (pad,) * 4)
        got = bend.Rotate(a, h, w)(xd).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)
