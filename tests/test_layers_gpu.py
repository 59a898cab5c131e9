"""GPU parity: ModulatedConv2d / StyledConv / ToRGB on the HIP path vs golden vectors from the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from maua_stylegan2_amd import _lib, seeding

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1e-4  # fp32 MFMA sums over K <= 16*9; north_star budget 1e-3


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_modulated_conv2d_golden(gpu, golden):
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    g = golden("layers.npz")
    for name in g["modconv.cases"]:
        cin, cout, k, up, demod = (int(v) for v in g[f"modconv.{name}.cfg"])
        m = ModulatedConv2d(cin, cout, k, 32, demodulate=bool(demod), upsample=bool(up)).to(gpu)
        # style_dim 32 is below the kernel's 64-lane granularity -> zero-pad style and modulation weight to 64
        m.weight.copy_(t(g[f"modconv.{name}.w"], gpu))
        mw = np.zeros((cin, 64), np.float32)
        mw[:, :32] = g[f"modconv.{name}.mw"] * np.sqrt(64 / 32)  # EqualLinear scale is 1/sqrt(in_dim)
        m.modulation.weight = torch.nn.Parameter(t(mw, gpu))
        m.modulation.bias.copy_(t(g[f"modconv.{name}.mb"], gpu))
        s = np.zeros((2, 64), np.float32)
        s[:, :32] = g[f"modconv.{name}.s"]
        y = m(t(g[f"modconv.{name}.x"], gpu), t(s, gpu))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"modconv.{name}.y"], atol=TOL, err_msg=str(name))


def _pad_mod(sd, prefix, dev):
    mw = sd[f"{prefix}.conv.modulation.weight"]
    out = np.zeros((mw.shape[0], 64), np.float32)
    out[:, :32] = mw * np.sqrt(2.0)
    return torch.nn.Parameter(t(out, dev))


@pytest.mark.parametrize("name,up", [("styled_plain", False), ("styled_up", True)])
def test_styled_conv_golden(gpu, golden, name, up):
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    g = golden("layers.npz")
    sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + ".sd.")}
    m = StyledConv(8, 10, 3, 32, upsample=up).to(gpu)
    m.conv.weight.copy_(t(sd["L.conv.weight"], gpu))
    m.conv.modulation.weight = _pad_mod(sd, "L", gpu)
    m.conv.modulation.bias.copy_(t(sd["L.conv.modulation.bias"], gpu))
    m.noise.weight.copy_(t(sd["L.noise.weight"], gpu))
    m.activate.bias.copy_(t(sd["L.activate.bias"], gpu))
    s = np.zeros((2, 64), np.float32)
    s[:, :32] = g[f"{name}.s"]
    y = m(t(g[f"{name}.x"], gpu), t(s, gpu), noise=t(g[f"{name}.noise"], gpu))
    np.testing.assert_allclose(y.cpu().numpy(), g[f"{name}.y"], atol=TOL)


@pytest.mark.parametrize("name", ["torgb_noskip", "torgb_skip"])
def test_to_rgb_golden(gpu, golden, name):
    from maua_stylegan2_amd.models.stylegan2 import ToRGB

    g = golden("layers.npz")
    sd = {k[len(name) + 4:]: g[k] for k in g.files if k.startswith(name + ".sd.")}
    m = ToRGB(8, 32, upsample=True).to(gpu)
    m.bias.copy_(t(sd["L.bias"], gpu))
    m.conv.weight.copy_(t(sd["L.conv.weight"], gpu))
    m.conv.modulation.weight = _pad_mod(sd, "L", gpu)
    m.conv.modulation.bias.copy_(t(sd["L.conv.modulation.bias"], gpu))
    s = np.zeros((2, 64), np.float32)
    s[:, :32] = g[f"{name}.s"]
    skip = t(g[f"{name}.skip"], gpu) if f"{name}.skip" in g.files else None
    y = m(t(g[f"{name}.x"], gpu), t(s, gpu), skip)
    np.testing.assert_allclose(y.cpu().numpy(), g[f"{name}.y"], atol=TOL)


@pytest.mark.parametrize("m_tiles,h,w,batch,with_skip", [(8, 64, 64, 2, True), (2, 256, 256, 1, True), (4, 32, 96, 3, False), (1, 8, 4, 1, True),
                                                         (3, 6, 20, 2, True)])
def test_torgb_plane_sum_form(gpu, m_tiles, h, w, batch, with_skip):
    """maua_torgb_f32 with w = s = NULL (the sum of per-tile partial ToRGB planes + bias + up-sampled skip) against the same call with a
    0 / 1 selection matrix and unit styles (the general kernel, which the golden vectors pin) and against the oracle's upfirdn2d."""
    from maua_stylegan2_amd.models.stylegan2 import Upsample
    from oracle import ops_oracle as oo

    r = np.random.default_rng(m_tiles * 1000 + h + w)
    part = r.standard_normal((batch, 3 * m_tiles, h, w)).astype(np.float32)
    bias = r.standard_normal(3).astype(np.float32)
    skip = r.standard_normal((batch, 3, h // 2, w // 2)).astype(np.float32) if with_skip else None
    up = Upsample([1, 3, 3, 1]).to(gpu)
    k4 = up.kernel
    lib = _lib.load()
    d_part, d_bias = t(part, gpu), t(bias, gpu)
    d_skip = t(skip, gpu) if with_skip else None
    got = torch.full((batch, 3, h, w), float("nan"), device=gpu)
    _lib.check(lib.maua_torgb_f32(d_part.data_ptr(), None, None, 0, d_bias.data_ptr(), _lib.ptr(d_skip), k4.data_ptr() if with_skip else None,
                                  got.data_ptr(), batch, 3 * m_tiles, h, w, 1.0, _lib.stream_ptr(gpu)), "maua_torgb_f32")
    sel = np.zeros((3, 3 * m_tiles), np.float32)
    for m in range(m_tiles):
        for c in range(3):
            sel[c, 3 * m + c] = 1.0
    ones = torch.ones(batch, 3 * m_tiles, device=gpu)
    ref = torch.full((batch, 3, h, w), float("nan"), device=gpu)
    _lib.check(lib.maua_torgb_f32(d_part.data_ptr(), t(sel, gpu).data_ptr(), ones.data_ptr(), 3 * m_tiles, d_bias.data_ptr(), _lib.ptr(d_skip),
                                  k4.data_ptr() if with_skip else None, ref.data_ptr(), batch, 3 * m_tiles, h, w, 1.0, _lib.stream_ptr(gpu)),
               "maua_torgb_f32")
    want = part.reshape(batch, m_tiles, 3, h, w).sum(1) + bias[None, :, None, None]
    if with_skip:
        want = want + oo.upfirdn2d(torch.from_numpy(skip), k4.cpu(), up=2, down=1, pad=up.pad).numpy()
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), atol=2e-6 * m_tiles, rtol=0)
    np.testing.assert_allclose(got.cpu().numpy(), want, atol=1e-5, rtol=1e-5)
    # rejected: a plane count that is not 3 M, a width that is not a multiple of 4, one of w / s alone
    assert lib.maua_torgb_f32(d_part.data_ptr(), None, None, 0, None, None, None, got.data_ptr(), batch, 3 * m_tiles + 1, h, w, 1.0, None) == -22
    assert lib.maua_torgb_f32(d_part.data_ptr(), None, None, 0, None, None, None, got.data_ptr(), batch, 3 * m_tiles, h, w + 2, 1.0, None) == -22
    assert lib.maua_torgb_f32(d_part.data_ptr(), d_bias.data_ptr(), None, 0, None, None, None, got.data_ptr(), batch, 3 * m_tiles, h, w, 1.0, None) == -22


@pytest.mark.parametrize("cin,cout,hw,up,batch", [
    (512, 512, 4, False, 3), (512, 512, 8, True, 2), (512, 512, 16, False, 1), (512, 256, 32, True, 1),
    (128, 128, 64, False, 2), (128, 64, 64, True, 1), (64, 64, 128, False, 1), (64, 32, 96, True, 1),
    (32, 32, 160, False, 2), (40, 24, 20, False, 1), (24, 72, 12, True, 2), (16, 8, 128, True, 2), (8, 40, 130, True, 1),
    (72, 40, 9, True, 3), (512, 512, 8, True, 8), (40, 96, 7, True, 2),  # position grids of 100 / 81 (two flat runs per image) / 64 (one 2-D tile)
])
def test_modconv_shapes_vs_oracle(gpu, cin, cout, hw, up, batch):
    """Every tile configuration (BM 32/64/128, split-K, polyphase) against the oracle's reference formulation."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(cin * 7 + cout + hw + up)
    m = ModulatedConv2d(cin, cout, 3, 512, upsample=up)
    w = r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)
    mw = r.standard_normal((cin, 512)).astype(np.float32)
    mb = (1 + 0.1 * r.standard_normal(cin)).astype(np.float32)
    m.weight.copy_(torch.from_numpy(w)), m.modulation.weight.copy_(torch.from_numpy(mw)), m.modulation.bias.copy_(torch.from_numpy(mb))
    m = m.to(gpu)
    x = r.standard_normal((batch, cin, hw, hw + (4 if hw % 8 == 0 and hw > 16 else 0))).astype(np.float32)
    s = r.standard_normal((batch, 512)).astype(np.float32)
    want = so.modulated_conv2d(torch.from_numpy(x), torch.from_numpy(s), torch.from_numpy(w), torch.from_numpy(mw),
                               torch.from_numpy(mb), upsample=up, blur_kernel=m.blur.kernel.cpu() if up else None).numpy()
    got = m(t(x, gpu), t(s, gpu)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-4)


def test_styled_conv_up_wide_vs_oracle(gpu):
    """Up-sampling StyledConv at 256 / 264-wide outputs: the 16-byte blur + noise + bias + act tail (fir_vec4_kernel),
    per-frame noise and broadcast (checkpoint buffer) noise."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(21)
    for cin, cout, h, w, bcast in [(8, 6, 128, 128, False), (16, 4, 33, 132, True)]:
        m = StyledConv(cin, cout, 3, 512, upsample=True)
        sd = {
            "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
            "L.conv.blur.kernel": m.conv.blur.kernel.clone(),
            "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
            "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
            "L.noise.weight": torch.tensor([0.41]),
            "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
        }
        m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
        m = m.to(gpu)
        x = torch.from_numpy(r.standard_normal((2, cin, h, w)).astype(np.float32))
        s = torch.from_numpy(r.standard_normal((2, 512)).astype(np.float32))
        nz = torch.from_numpy(r.standard_normal((1 if bcast else 2, 1, 2 * h, 2 * w)).astype(np.float32))
        want = so.styled_conv(sd, "L", x, s, nz, True).numpy()
        got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (128, 128, 64, 64, 2),    # FAST path, whole tiles
    (256, 256, 32, 36, 1),    # ragged pair grid (18 pairs -> 2 tiles of 16), split-K
    (136, 200, 40, 70, 3),    # Cout not a multiple of the 128-row weight tile, Cin % 4 == 0 only -> generic loads
    (64, 160, 9, 34, 2),      # short map, several images... one image per tile, ragged rows
    (130, 128, 16, 32, 2),    # Cin % 4 != 0 -> generic path with a partial last chunk
    (32, 32, 72, 96, 2),      # 32-row weight tile (BM 32), 8-row tiles with a ragged last row block
    (64, 64, 48, 64, 1),      # 64-row weight tile
    (24, 40, 20, 38, 2),      # generic path: Cin % 8 != 0, Cout not a multiple of 32
])
def test_modconv_winograd_vs_oracle(gpu, cin, cout, h, w, batch):
    """Plain 3x3 layers with >= 128 output channels run the Winograd F(2,3) mode (mode 2 of maua_modconv3x3_f32);
    the StyledConv tail (noise + bias + leaky ReLU) is applied to both outputs of a pair in the epilogue."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(cin + 3 * cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=False)
    m.conv.winograd43_min_cout = m.conv.winograd2d_min_cout = 1 << 30  # this test pins the F(2,3) mode
    assert m.conv.conv_mode(h, w) == 2
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.37]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-4)
    # and against the direct (mode 0) kernel of the same layer: the two differ only by fp32 rounding
    m.conv.winograd_min_cout = m.conv.winograd43_min_cout = 1 << 30
    assert m.conv.conv_mode(h, w) == 0
    direct = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, direct, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (128, 128, 64, 64, 2),    # 128-row weight tile, FAST path, whole tiles
    (64, 64, 40, 128, 1),     # 64-row tile (TM 2 in one wave row), ragged row blocks
    (136, 200, 24, 72, 2),    # generic loads: Cout not a multiple of the tile, ragged quads (18 per row -> 2 sub-tiles)
    (256, 256, 8, 64, 1),     # short map, split-K
    (32, 32, 48, 192, 2),     # 32-row weight tile (three patch slots per thread)
    (18, 32, 20, 68, 1),      # generic path at the 32-row tile: Cin % 4 != 0
])
def test_modconv_winograd43_vs_oracle(gpu, cin, cout, h, w, batch):
    """Plain 3x3 layers with >= 64 output channels on maps >= 64 wide (W % 4 == 0) run Winograd F(4,3) (mode 3).  Its
    transform constants span 1/24 .. 8, so the fp32 error is ~2e-5 of the output scale instead of ~1e-6: the tolerance
    against the oracle is 5e-4 (north_star budget 1e-3), and the direct kernel must agree to 2e-4."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(2 * cin + cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=False)
    m.conv.winograd2d_min_cout = 1 << 30  # this test pins the 1-D F(4,3) mode
    assert m.conv.conv_mode(h, w) == 3
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.29]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=1e-4)
    m.conv.winograd_min_cout = m.conv.winograd43_min_cout = 1 << 30
    direct = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, direct, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (64, 64, 32, 32, 2),      # one m-tile group (BM 64), 1 x 4 tiles per image
    (128, 128, 16, 64, 1),    # two weight tiles per position tile, 2 x 2 tiles
    (32, 32, 32, 64, 2),      # 32-channel layer: the wave-complete kernel (16-row tiles, one n-tile per wave)
    (32, 32, 24, 32, 1),      # 32-channel layer whose height is not a multiple of 16: the y-frequency-split <2, 2, 3> instance
    (16, 32, 16, 32, 3),      # wave-complete kernel, 4 K steps, a single tile per image
    (512, 256, 8, 32, 1),     # deep K (128 chunks), a single 8-row tile
    (64, 192, 24, 96, 1),     # three weight tiles, 3 x 3 position tiles
    (36, 64, 40, 32, 3),      # Cin = 9 chunks, ragged nothing: H % 8 == 0, three images
])
def test_modconv_winograd2d_vs_oracle(gpu, cin, cout, h, w, batch):
    """Plain 3x3 layers whose shape qualifies run 2-D Winograd F(2x4, 3x3) (mode 5, csrc/modconv_w2d.hip: F(2,3) along y
    on top of F(4,3) along x, the four waves of a workgroup splitting the y-frequencies).  Same tolerances as the 1-D
    F(4,3) mode: 5e-4 against the oracle (north_star budget 1e-3), 2e-4 against the direct kernel; also without the fused
    tail (ModulatedConv2d.forward) and with a shared [1,1,H,W] noise map."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(5 * cin + cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=False)
    m.conv.winograd2d_min_cout = 32  # (the generator itself uses mode 5 from 128 channels up)
    assert m.conv.conv_mode(h, w) == 5
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.31]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=1e-4)
    shared = m(x.to(gpu), s.to(gpu), noise=nz[:1].to(gpu)).cpu().numpy()
    np.testing.assert_allclose(shared, so.styled_conv(sd, "L", x, s, nz[:1], False).numpy(), atol=5e-4, rtol=1e-4)
    raw = m.conv(x.to(gpu), s.to(gpu)).cpu().numpy()  # no tail: demodulated convolution only
    want_raw = so.modulated_conv2d(x, s, sd["L.conv.weight"], sd["L.conv.modulation.weight"], sd["L.conv.modulation.bias"]).numpy()
    np.testing.assert_allclose(raw, want_raw, atol=5e-4, rtol=1e-4)
    m.conv.winograd_min_cout = m.conv.winograd43_min_cout = m.conv.winograd2d_min_cout = 1 << 30
    assert m.conv.conv_mode(h, w) == 0
    direct = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(got, direct, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,with_skip", [(64, 64, 32, 64, True), (32, 32, 32, 32, True), (128, 64, 16, 32, False),
                                                     (32, 32, 48, 64, True), (64, 128, 32, 32, True), (128, 256, 16, 64, False),
                                                     (32, 512, 8, 32, True), (32, 32, 40, 32, True), (64, 32, 16, 64, False)])
def test_fused_torgb_epilogue_equals_separate_launches(gpu, cin, cout, h, w, with_skip):
    """StyledConv + ToRGB folded into one launch (maua_styledconv_torgb_f32: <= 64-channel plain layers, every kernel mode that
    the layer shape selects — 2-D Winograd here) against the same two layers run as separate launches and against the oracle;
    with ``store`` off the feature map is not written at all.  Wider layers (2, 4, 8 output-channel tiles): the conv leaves
    per-tile partial ToRGB sums (maua_styledconv_torgb_partial_f32) and maua_torgb_f32 adds them up."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv, ToRGB
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(cin + cout + h + w)
    b = 2
    conv, rgb = StyledConv(cin, cout, 3, 512), ToRGB(cout, 512)
    sd = {
        "C.conv.weight": r.standard_normal((1, cout, cin, 3, 3)), "C.conv.modulation.weight": r.standard_normal((cin, 512)),
        "C.conv.modulation.bias": 1 + 0.1 * r.standard_normal(cin), "C.noise.weight": np.array([0.23]),
        "C.activate.bias": 0.3 * r.standard_normal(cout),
        "T.bias": 0.3 * r.standard_normal((1, 3, 1, 1)), "T.upsample.kernel": seeding.fir_kernel_2d((1, 3, 3, 1), 4.0),
        "T.conv.weight": r.standard_normal((1, 3, cout, 1, 1)), "T.conv.modulation.weight": r.standard_normal((cout, 512)),
        "T.conv.modulation.bias": 1 + 0.1 * r.standard_normal(cout),
    }
    sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in sd.items()}
    conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("C.")}, strict=True)
    rgb.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("T.")}, strict=True)
    conv, rgb = conv.to(gpu), rgb.to(gpu)
    conv.conv.winograd2d_min_cout = 32
    assert conv.conv.conv_mode(h, w) == 5
    x = torch.from_numpy(r.standard_normal((b, cin, h, w)).astype(np.float32))
    s1 = torch.from_numpy(r.standard_normal((b, 512)).astype(np.float32))
    s2 = torch.from_numpy(r.standard_normal((b, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((b, 1, h, w)).astype(np.float32))
    skip = torch.from_numpy(r.standard_normal((b, 3, h // 2, w // 2)).astype(np.float32)) if with_skip else None
    feat_want = so.styled_conv(sd, "C", x, s1, nz, False)
    img_want = so.to_rgb(sd, "T", feat_want, s2, skip).numpy()
    # separate launches (the public modules)
    feat = conv(x.to(gpu), s1.to(gpu), noise=nz.to(gpu))
    img_sep = rgb(feat, s2.to(gpu), skip.to(gpu) if with_skip else None).cpu().numpy()
    np.testing.assert_allclose(img_sep, img_want, atol=2e-3, rtol=1e-4)
    # fused launch through StyledConv.run on precomputed styles
    lib = _lib.load()
    from maua_stylegan2_amd.models.stylegan2 import _style_table

    entries = [conv.conv.table_entry(0, 0, 0), rgb.conv.table_entry(1, cin, b * cout)]
    table = _style_table(entries, gpu)
    lat = torch.stack([s1, s2], 1).to(gpu).contiguous()
    styles = torch.empty(b, cin + cout, device=gpu)
    demod = torch.empty(b * cout, device=gpu)
    st = _lib.stream_ptr(gpu)
    _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), b, 2, 512, None, None, table.data_ptr(), 2, max(cin, cout),
                                         styles.data_ptr(), cin + cout, None, st), "affine")
    _lib.check(lib.maua_demod_f32(table.data_ptr(), 2, cout, styles.data_ptr(), cin + cout, demod.data_ptr(), b, st), "demod")
    for store in ((True, False) if cout <= 64 else (True,)):
        out_img = torch.full((b, 3, h, w), float("nan"), device=gpu)
        feat_buf = {}

        def bufs(name, shape):
            feat_buf[name] = torch.full(shape, float("nan"), device=gpu)
            return feat_buf[name]

        fuse = dict(module=rgb, s_off=cin, skip=skip.to(gpu) if with_skip else None, out=out_img, store=store)
        conv.run(x.to(gpu), styles, 0, demod.view(b, cout), nz.to(gpu), bufs, "f", rgb=fuse)
        assert fuse.get("done"), "the layer was expected to take the fused path"
        np.testing.assert_allclose(out_img.cpu().numpy(), img_want, atol=2e-3, rtol=1e-4)
        if store:
            np.testing.assert_allclose(feat_buf["f"].cpu().numpy(), feat_want.numpy(), atol=5e-4, rtol=1e-4)
        else:
            assert torch.isnan(feat_buf["f"]).all()  # never written


def _lowres_setup(gpu, cin, cout, h, w, b, up, seed):
    """A StyledConv (+ ToRGB for the plain case) with random weights, styles / demodulation factors from the table kernels, inputs."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv, ToRGB, _style_table

    r = np.random.default_rng(seed)
    conv, rgb = StyledConv(cin, cout, 3, 512, upsample=up), ToRGB(cout, 512)
    sd = {
        "C.conv.weight": r.standard_normal((1, cout, cin, 3, 3)), "C.conv.modulation.weight": r.standard_normal((cin, 512)),
        "C.conv.modulation.bias": 1 + 0.1 * r.standard_normal(cin), "C.noise.weight": np.array([0.23]),
        "C.activate.bias": 0.3 * r.standard_normal(cout),
        "T.bias": 0.3 * r.standard_normal((1, 3, 1, 1)), "T.upsample.kernel": seeding.fir_kernel_2d((1, 3, 3, 1), 4.0),
        "T.conv.weight": r.standard_normal((1, 3, cout, 1, 1)), "T.conv.modulation.weight": r.standard_normal((cout, 512)),
        "T.conv.modulation.bias": 1 + 0.1 * r.standard_normal(cout),
    }
    if up:
        sd["C.conv.blur.kernel"] = seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)
    sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in sd.items()}
    conv.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("C.")}, strict=True)
    rgb.load_state_dict({k[2:]: v for k, v in sd.items() if k.startswith("T.")}, strict=True)
    conv, rgb = conv.to(gpu), rgb.to(gpu)
    lib = _lib.load()
    s1 = torch.from_numpy(r.standard_normal((b, 512)).astype(np.float32))
    s2 = torch.from_numpy(r.standard_normal((b, 512)).astype(np.float32))
    entries = [conv.conv.table_entry(0, 0, 0), rgb.conv.table_entry(1, cin, b * cout)]
    table = _style_table(entries, gpu)
    lat = torch.stack([s1, s2], 1).to(gpu).contiguous()
    styles = torch.empty(b, cin + cout, device=gpu)
    demod = torch.empty(b * cout, device=gpu)
    st = _lib.stream_ptr(gpu)
    _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), b, 2, 512, None, None, table.data_ptr(), 2, max(cin, cout), styles.data_ptr(),
                                         cin + cout, None, st), "affine")
    _lib.check(lib.maua_demod_f32(table.data_ptr(), 2, cout, styles.data_ptr(), cin + cout, demod.data_ptr(), b, st), "demod")
    x = torch.from_numpy(r.standard_normal((b, cin, h, w)).astype(np.float32))
    return conv, rgb, sd, styles, demod, x, s1, s2, r


@pytest.mark.parametrize("cin,cout,h,w,b,noise_b,post", [(512, 512, 4, 4, 8, 8, False), (512, 512, 8, 8, 8, 1, False), (512, 512, 16, 16, 3, 3, True),
                                                           (24, 40, 5, 7, 2, 2, True), (64, 32, 16, 8, 1, 1, False), (32, 64, 1, 1, 2, 2, False),
                                                           (512, 512, 16, 16, 8, 8, False), (64, 32, 16, 16, 1, 1, True), (72, 96, 16, 16, 2, 1, False),
                                                           (64, 32, 16, 16, 2, 2, False), (64, 64, 16, 16, 3, 3, False)])
def test_upconv_blur_lowres_equals_three_launches_and_oracle(gpu, cin, cout, h, w, b, noise_b, post):
    """The low-resolution entry of an up-sampling StyledConv (polyphase convolution -> split-K slabs, then slab sum + demodulation + blur +
    noise + bias + leaky ReLU in one launch) against the three-launch path it replaces — BIT-identical: the slab sum keeps reduce_tail_kernel's
    association, the blur fir_tile_kernel's order — and against the oracle; with the style fold's scale on the stored map."""
    from oracle import stylegan2_oracle as so

    conv, _, sd, styles, demod, x, s1, _, r = _lowres_setup(gpu, cin, cout, h, w, b, True, cin + cout + h + w)
    assert conv.conv.conv_mode(h, w) == 1 and _lib.load().maua_lowres_ok(cin, cout, h, w, 1) == 1
    nz = torch.from_numpy(r.standard_normal((noise_b, 1, 2 * h, 2 * w)).astype(np.float32))
    post_s = torch.from_numpy(r.standard_normal((b, cin + cout)).astype(np.float32)).to(gpu) if post else None
    full = styles if not post else torch.cat([styles, post_s], 1).contiguous()  # (post_off indexes the same table: a second block of styles)
    ok6 = _lib.load().maua_lowres_ok(cin, cout, h, w, 6) == 1  # 16-wide inputs: the F(2,2)^2 kernel on 16 x 16-position tiles, K split
    outs = {}
    for name, fused, up2d in (("pair", False, False), ("low1", True, False)) + ((("low6", True, True),) if ok6 else ()):
        conv.lowres_fusion, conv.lowres_up2d = fused, up2d
        held = {}

        def bufs(name, shape):
            held[name] = torch.full(tuple(int(v) for v in shape), float("nan"), device=gpu)
            return held[name]

        y = conv.run(x.to(gpu), full, 0, demod.view(b, cout), nz.to(gpu), bufs, "u", post_off=(cin + cout) if post else None)
        assert conv.last_path == ("lowres" if fused else "pair") and conv.posted == post
        if fused:
            assert ("up2d" in _lib.last_modconv_instance()) == up2d
        outs[name] = y.cpu().numpy()
    if _lib.load().maua_modconv_ws_floats(b, cin, cout, h, w, 1) > 0:  # K is split: the same slabs, the same association
        assert np.array_equal(outs["low1"], outs["pair"])
    else:                                                              # (wscale * d as one factor in the pair, as two here)
        np.testing.assert_allclose(outs["low1"], outs["pair"], atol=1e-5, rtol=1e-5)
    outs[True] = outs["low6"] if ok6 else outs["low1"]
    if ok6:
        np.testing.assert_allclose(outs["low6"], outs["pair"], atol=2e-4 * max(1.0, float(np.abs(outs["pair"]).max()) / 8), rtol=2e-4)
    want = so.styled_conv(sd, "C", x, s1, nz.expand(b, -1, -1, -1) if noise_b == 1 else nz, True).numpy()
    if post:
        want = want * post_s[:, :cout].cpu().numpy()[:, :, None, None]
    np.testing.assert_allclose(outs[True], want, atol=5e-4 * max(1.0, float(np.abs(want).max()) / 8), rtol=2e-4)


@pytest.mark.parametrize("small_wino", [False, True])
@pytest.mark.parametrize("cin,cout,h,w,b,with_skip", [(512, 512, 4, 4, 8, False), (512, 512, 8, 8, 8, True), (512, 512, 16, 16, 2, True),
                                                        (40, 96, 4, 8, 3, True), (64, 32, 16, 16, 1, False), (32, 64, 4, 4, 2, False),
                                                        (128, 256, 8, 16, 3, True)])
def test_styledconv_rgbpart_lowres_equals_separate_launches_and_oracle(gpu, cin, cout, h, w, b, with_skip, small_wino):
    """The low-resolution entry of a plain StyledConv + ToRGB (direct convolution -> split-K slabs; slab sum + tail + per-group partial ToRGB sums;
    plane sum) against convolution, reduce + tail, ToRGB as separate launches: the feature map BIT-identical, the image within rounding of the
    re-associated channel sum; and against the oracle."""
    from oracle import stylegan2_oracle as so

    conv, rgb, sd, styles, demod, x, s1, s2, r = _lowres_setup(gpu, cin, cout, h, w, b, False, 3 * cin + cout + h + w)
    conv.conv.winograd_min_cout = conv.conv.winograd43_min_cout = conv.conv.winograd2d_min_cout = 1 << 30
    # small_wino: the 8^2 / 16^2 layers of 128 and more channels through Winograd F(2,3) along x (the generator's default), else the direct form
    conv.conv.winograd_small_min_cout = 128 if small_wino else 1 << 30
    want_mode = 2 if (small_wino and cout >= 128 and 8 <= w < 32 and h >= 8) else 0
    if small_wino and want_mode == 0:
        pytest.skip("the shape stays on the direct form either way")
    assert conv.conv.conv_mode(h, w) == want_mode and _lib.load().maua_lowres_ok(cin, cout, h, w, want_mode) == 1
    nz = torch.from_numpy(r.standard_normal((b, 1, h, w)).astype(np.float32))
    skip = torch.from_numpy(r.standard_normal((b, 3, h // 2, w // 2)).astype(np.float32)) if with_skip else None
    feat_want = so.styled_conv(sd, "C", x, s1, nz, False)
    img_want = so.to_rgb(sd, "T", feat_want, s2, skip).numpy()
    got = {}
    for fused in (True, False):
        conv.lowres_fusion = fused
        held = {}

        def bufs(name, shape):
            held[name] = torch.full(tuple(int(v) for v in shape), float("nan"), device=gpu)
            return held[name]

        img = torch.full((b, 3, h, w), float("nan"), device=gpu)
        fuse = dict(module=rgb, s_off=cin, skip=skip.to(gpu) if with_skip else None, out=img, store=True)
        y = conv.run(x.to(gpu), styles, 0, demod.view(b, cout), nz.to(gpu), bufs, "p", rgb=fuse)
        assert bool(fuse.get("done")) == fused and conv.last_path == ("lowres" if fused else "plain")
        if not fused:
            rgb.run(y, styles, cin, skip.to(gpu) if with_skip else None, img)
        got[fused] = (y.cpu().numpy(), img.cpu().numpy())
    if _lib.load().maua_modconv_ws_floats(b, cin, cout, h, w, want_mode) > 0:
        assert np.array_equal(got[True][0], got[False][0])
    else:
        np.testing.assert_allclose(got[True][0], got[False][0], atol=1e-5, rtol=1e-5)
    scale = max(1.0, float(np.abs(img_want).max()))
    np.testing.assert_allclose(got[True][1], got[False][1], atol=2e-5 * scale, rtol=1e-5)
    np.testing.assert_allclose(got[True][0], feat_want.numpy(), atol=5e-4, rtol=2e-4)
    np.testing.assert_allclose(got[True][1], img_want, atol=2e-3, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (128, 64, 64, 64, 2),     # 64-row tile, flat pair runs, FAST path
    (64, 32, 48, 96, 1),      # 32-row tile (three patch slots per thread)
    (24, 72, 12, 20, 2),      # generic loads (Cin % 8 != 0), Cout not a multiple of the tile, small ragged grid
    (512, 256, 16, 16, 1),    # split-K
    (64, 32, 130, 128, 1),    # large grid: the size at which the generator itself switches to mode 4
])
def test_upconv_winograd_matches_polyphase_and_oracle(gpu, cin, cout, h, w, batch):
    """Up-sampling ModulatedConv2d: mode 4 (F(2,2) on the even x-phase) against the oracle's conv_transpose2d + blur and
    against the plain polyphase kernel (mode 1) of the same layer."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(cin + cout + h + w)
    m = ModulatedConv2d(cin, cout, 3, 512, upsample=True)
    m.conv_mode = lambda hh, ww, _m=m: 4 if _m.upconv_winograd else 1  # force the mode under test at every size
    wgt = r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)
    mw = r.standard_normal((cin, 512)).astype(np.float32)
    mb = (1 + 0.1 * r.standard_normal(cin)).astype(np.float32)
    m.weight.copy_(torch.from_numpy(wgt)), m.modulation.weight.copy_(torch.from_numpy(mw)), m.modulation.bias.copy_(torch.from_numpy(mb))
    m = m.to(gpu)
    x = r.standard_normal((batch, cin, h, w)).astype(np.float32)
    s = r.standard_normal((batch, 512)).astype(np.float32)
    want = so.modulated_conv2d(torch.from_numpy(x), torch.from_numpy(s), torch.from_numpy(wgt), torch.from_numpy(mw),
                               torch.from_numpy(mb), upsample=True, blur_kernel=m.blur.kernel.cpu()).numpy()
    got = m(t(x, gpu), t(s, gpu)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-4)
    m.upconv_winograd = False
    ref = m(t(x, gpu), t(s, gpu)).cpu().numpy()
    np.testing.assert_allclose(got, ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (64, 32, 8, 32, 1),       # one tile per image and m-tile
    (128, 64, 64, 64, 2),     # two m-tiles, 8 x 2 tiles
    (512, 256, 16, 32, 1),    # long K loop (128 steps), 8 m-tiles
    (32, 32, 40, 96, 2),      # tile counts that are not powers of two
    (8, 64, 8, 64, 3),        # two K steps only: prologue / epilogue dominated
    (64, 32, 256, 256, 1),    # generator-sized grid (convs.12's shape class)
])
def test_upconv_two_axis_f22_matches_polyphase_and_oracle(gpu, cin, cout, h, w, batch):
    """Up-sampling ModulatedConv2d on mode 6 (csrc/modconv_up2d.hip: F(2,2) on both axes of the polyphase form + the two edge
    lines) against the oracle's conv_transpose2d + blur and against the plain polyphase kernel (mode 1); the RAW (2H+1) x (2W+1)
    output is compared as well, on a NaN-prefilled buffer: every element must be written, by exactly the right formula."""
    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(cin + cout + h + w)
    m = ModulatedConv2d(cin, cout, 3, 512, upsample=True)
    assert m.conv_mode(h, w) == 6
    wgt = r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)
    mw = r.standard_normal((cin, 512)).astype(np.float32)
    mb = (1 + 0.1 * r.standard_normal(cin)).astype(np.float32)
    m.weight.copy_(torch.from_numpy(wgt)), m.modulation.weight.copy_(torch.from_numpy(mw)), m.modulation.bias.copy_(torch.from_numpy(mb))
    m = m.to(gpu)
    x = r.standard_normal((batch, cin, h, w)).astype(np.float32)
    s = r.standard_normal((batch, 512)).astype(np.float32)
    want = so.modulated_conv2d(torch.from_numpy(x), torch.from_numpy(s), torch.from_numpy(wgt), torch.from_numpy(mw),
                               torch.from_numpy(mb), upsample=True, blur_kernel=m.blur.kernel.cpu()).numpy()
    got = m(t(x, gpu), t(s, gpu)).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=3e-4, rtol=1e-4)
    # raw transposed convolution (unit demod), mode 6 vs mode 1, every element written
    xs = t(x, gpu)
    styles = t(r.standard_normal((batch, cin)).astype(np.float32), gpu)
    raws = {}
    for mode in (6, 1):
        m.conv_mode = lambda hh, ww, _mode=mode: _mode
        raw = torch.full((batch, cout, 2 * h + 1, 2 * w + 1), float("nan"), device=gpu)
        n_ws = _lib.load().maua_modconv_ws_floats(batch, cin, cout, h, w, mode)
        ws = torch.empty(max(n_ws, 1), device=gpu)
        m.run(xs, styles, 0, None, raw, ws if n_ws else None)
        raws[mode] = raw.cpu().numpy()
    assert np.isfinite(raws[6]).all(), "mode 6 left output elements unwritten"
    scale = np.abs(raws[1]).max()
    np.testing.assert_allclose(raws[6], raws[1], atol=2e-5 * scale, rtol=1e-4)
    ref = F.conv_transpose2d((torch.from_numpy(x) * styles.cpu()[:, :, None, None]), torch.from_numpy(wgt[0]).transpose(0, 1).contiguous(),
                             stride=2).numpy() * m.scale
    np.testing.assert_allclose(raws[6], ref, atol=3e-5 * scale, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (128, 128, 16, 64, 2),    # one weight tile, 2 x 2 position tiles, 8 K chunks
    (256, 256, 32, 32, 1),    # two weight tiles
    (512, 128, 8, 32, 1),     # deep K (32 chunks), a single tile
    (48, 384, 24, 96, 2),     # three K chunks, three weight tiles, 3 x 3 tiles: image borders on every side
])
def test_modconv_split_bf16_vs_oracle(gpu, cin, cout, h, w, batch):
    """SIDE MEASUREMENT (mode 7, csrc/modconv_sbf16.hip; off by default): the plain 3x3 modulated convolution in its direct 9-tap
    form with split-bf16 products on the bf16 matrix cores (a b ~= a_h b_h + a_h b_l + a_l b_h, fp32 accumulation).  Same bounds
    as the fp32 Winograd modes — 5e-4 against the oracle, 2e-4 against the fp32 direct kernel — with and without the fused tail;
    the measured maxima are printed."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(7 * cin + cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=False)
    m.conv.split_bf16_min_cout = 128
    assert m.conv.conv_mode(h, w) == 7
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.31]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    raw = m.conv(x.to(gpu), s.to(gpu)).cpu().numpy()  # no tail: demodulated convolution only
    want_raw = so.modulated_conv2d(x, s, sd["L.conv.weight"], sd["L.conv.modulation.weight"], sd["L.conv.modulation.bias"]).numpy()
    m.conv.split_bf16_min_cout = m.conv.winograd_min_cout = m.conv.winograd43_min_cout = m.conv.winograd2d_min_cout = 1 << 30
    assert m.conv.conv_mode(h, w) == 0
    direct = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    print(f"[split-bf16 {cin}->{cout} @{h}x{w}] max |mode 7 - oracle| = {np.abs(got - want).max():.2e} (output std {want.std():.2f}), "
          f"raw {np.abs(raw - want_raw).max():.2e}, vs fp32 direct kernel {np.abs(got - direct).max():.2e}; "
          f"fp32 direct vs oracle {np.abs(direct - want).max():.2e}")
    np.testing.assert_allclose(got, want, atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(raw, want_raw, atol=5e-4, rtol=1e-4)
    np.testing.assert_allclose(got, direct, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,h,w,batch", [
    (128, 128, 16, 32, 2),    # one weight tile, 2 x 1 position tiles
    (256, 256, 8, 64, 1),     # two weight tiles, 1 x 2 position tiles
    (48, 384, 24, 96, 1),     # three K chunks, twelve m-tiles, 3 x 3 tiles
    (64, 32, 16, 32, 2),      # a single 32-channel m-tile (convs.14's shape class)
])
def test_upconv_split_bf16_matches_polyphase_and_oracle(gpu, cin, cout, h, w, batch):
    """SIDE MEASUREMENT (mode 8, csrc/modconv_sbf16.hip; off by default): the stride-2 transposed convolution (models/stylegan2.py:229-237)
    in its polyphase form — all four output phases of a position in one workgroup — on the bf16 matrix cores with split-bf16 products + the
    fp32 edge lines, written into a NaN-prefilled
    raw map: every element of [B, Cout, 2H+1, 2W+1] must be written and agree with the fp32 polyphase kernel (mode 1) within 2e-4; then
    the whole StyledConv (blur + noise + bias + act) against the oracle within 5e-4."""
    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.models.stylegan2 import StyledConv, _style_table
    from oracle import stylegan2_oracle as so

    r = np.random.default_rng(11 * cin + cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=True)
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.blur.kernel": torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)),
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
    lib = _lib.load()

    def raw_map(mode_switch):
        m.conv.split_bf16_up_min_cout = mode_switch
        mode = m.conv.conv_mode(h, w)
        xs, ss = x.to(gpu), s.to(gpu)
        st = torch.empty((batch, cin), device=gpu)
        dm = torch.empty((batch, cout), device=gpu)
        table = _style_table([m.conv.table_entry(0, 0, 0)], gpu)
        lat = ss.reshape(batch, 1, -1)
        _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), batch, 1, 512, None, None, table.data_ptr(), 1, cin, st.data_ptr(), cin, None,
                                             _lib.stream_ptr(gpu)), "affine")
        _lib.check(lib.maua_demod_f32(table.data_ptr(), 1, cout, st.data_ptr(), cin, dm.data_ptr(), batch, _lib.stream_ptr(gpu)), "demod")
        out = torch.full((batch, cout, 2 * h + 1, 2 * w + 1), float("nan"), device=gpu)
        n_ws = lib.maua_modconv_ws_floats(batch, cin, cout, h, w, mode)
        ws = torch.empty(max(n_ws, 1), device=gpu)
        m.conv.run(xs, st, 0, dm, out, ws if n_ws else None)
        return mode, out.cpu().numpy()

    mode8, got = raw_map(32)
    assert mode8 == 8 and np.isfinite(got).all(), "an element of the raw map was not written"
    m.conv.upwino2d_min_cout = 1 << 30
    m.conv.upconv_winograd = False
    mode1, ref = raw_map(1 << 30)
    assert mode1 == 1
    print(f"[split-bf16 transposed {cin}->{cout} @{h}x{w}] max |mode 8 - mode 1 (fp32 polyphase)| = {np.abs(got - ref).max():.2e} "
          f"(raw std {ref.std():.2f})")
    np.testing.assert_allclose(got, ref, atol=2e-4, rtol=1e-4)
    m.conv.split_bf16_up_min_cout = 32
    full = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
    want = so.styled_conv(sd, "L", x, s, nz, True).numpy()
    print(f"[split-bf16 transposed {cin}->{cout} @{h}x{w}] StyledConv: max |hip - oracle| = {np.abs(full - want).max():.2e} (std {want.std():.2f})")
    np.testing.assert_allclose(full, want, atol=5e-4, rtol=1e-4)


# ---- round 5: the whole up-sampling StyledConv as one kernel (maua_upconv_blur_f32, csrc/modconv_up2d.hip FUSE == 2) --------------------


def _styled_up(cin, cout, seed, dev):
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    r = np.random.default_rng(seed)
    m = StyledConv(cin, cout, 3, 512, upsample=True)
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.blur.kernel": m.conv.blur.kernel.clone(),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.41]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    return m.to(dev), sd, r


@pytest.mark.parametrize("cin,cout,h,w,batch,noise_batch", [
    (64, 32, 8, 32, 1, 1),       # one x tile pair, two y tiles (the second holds raw row 2H only)
    (64, 32, 32, 32, 3, 3),      # several vertical segments: seam rows through the second launch
    (128, 64, 64, 64, 2, 1),     # two m-tiles, shared noise map (a checkpoint buffer)
    (32, 32, 40, 96, 2, 2),      # tile counts that are not powers of two, wide map: four x tiles
    (256, 64, 24, 64, 1, 0),     # long K loop, no noise map
    (64, 32, 256, 256, 1, 1),    # generator-sized grid (convs.12's shape class): 9 x 33 tiles
])
def test_upconv_blur_fused_equals_two_launches_and_oracle(gpu, cin, cout, h, w, batch, noise_batch):
    """reference models/stylegan2.py:229-238,262-266,338-343 — transposed conv -> blur -> noise -> bias -> leaky ReLU — as ONE kernel against
    (a) the two-launch path (mode-6 transposed convolution writing the raw (2H+1) x (2W+1) map, then maua_blur_noise_act_f32) on a
    NaN-prefilled output: every element written, agreement to 1e-5 of the output scale (the separable blur sums in another order);
    (b) the oracle's StyledConv (3e-4)."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv
    from oracle import stylegan2_oracle as so

    m, sd, r = _styled_up(cin, cout, cin + cout + h + w, gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((noise_batch, 1, 2 * h, 2 * w)).astype(np.float32)) if noise_batch else None
    keep = StyledConv.fused_blur_min_width
    outs = {}
    try:
        for name, width in (("fused", 32), ("pair", 1 << 30)):
            StyledConv.fused_blur_min_width = width
            launches = []
            real = m.conv.run
            m.conv.run = lambda *a, _real=real, **k: (launches.append("conv.run"), _real(*a, **k))[1]
            noise = torch.zeros(1, 1, 2 * h, 2 * w, device=gpu) if nz is None else nz.to(gpu)
            outs[name] = m(x.to(gpu), s.to(gpu), noise=noise).cpu().numpy()
            m.conv.run = real
            assert (launches == []) == (name == "fused"), (name, launches)  # the fused path never runs the raw-map convolution
    finally:
        StyledConv.fused_blur_min_width = keep
    assert np.isfinite(outs["fused"]).all()
    scale = np.abs(outs["pair"]).max()
    np.testing.assert_allclose(outs["fused"], outs["pair"], atol=1e-5 * scale, rtol=0)
    want = so.styled_conv(sd, "L", x, s, torch.zeros(1, 1, 2 * h, 2 * w) if nz is None else nz, True).numpy()
    np.testing.assert_allclose(outs["fused"], want, atol=3e-4, rtol=1e-4)


def test_upconv_blur_fused_c_abi_contract(gpu):
    """maua_upconv_blur_f32 through the C ABI: unwritten elements would stay NaN; noise through a frame source (the captured forward's
    way) equals noise through the argument; shapes outside maua_upconv_blur_ok are refused with MAUA_ENOSYS, missing operands with
    MAUA_EINVAL; a non-separable blur kernel keeps the layer on the two-launch path."""
    import ctypes

    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    lib = _lib.load()
    cin, cout, h, w, b = 64, 32, 32, 64, 2
    m, sd, r = _styled_up(cin, cout, 77, gpu)
    conv = m.conv
    x = torch.from_numpy(r.standard_normal((b, cin, h, w)).astype(np.float32)).to(gpu)
    s = torch.from_numpy(r.standard_normal((b, cin)).astype(np.float32)).to(gpu)
    d = (torch.rand(b, cout) + 0.5).to(gpu)
    frames = 5
    nz_seq = torch.from_numpy(r.standard_normal((frames, 1, 2 * h, 2 * w)).astype(np.float32)).to(gpu)
    wq = conv.packed_wino(6)
    k = conv.blur.kernel
    n_ws = lib.maua_upconv_blur_ws_floats(b, cin, cout, h, w)
    ws = torch.empty(max(n_ws, 1), device=gpu)
    st = _lib.stream_ptr(gpu)

    def call(out, noise, nstride, src=None, slot=0, cin_=cin, ws_=ws, k_=k):
        return lib.maua_upconv_blur_f32(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, d.data_ptr(), out.data_ptr(), _lib.ptr(ws_), k_.data_ptr(),
                                        _lib.ptr(noise), nstride, m.noise.weight.data_ptr(), m.activate.bias.data_ptr(), src, slot, b, cin_, cout,
                                        h, w, float(conv.scale), None, st)

    frame0 = 2
    direct = torch.full((b, cout, 2 * h, 2 * w), float("nan"), device=gpu)
    assert call(direct, nz_seq[frame0: frame0 + b].contiguous(), 4 * h * w) == 0
    src = _lib.FrameSource()
    src.frame0 = frame0
    src.noise[3] = nz_seq.data_ptr()
    src.noise_stride[3] = 4 * h * w
    dev_src = torch.frombuffer(bytearray(bytes(src)), dtype=torch.uint8).to(gpu)
    via_src = torch.full_like(direct, float("nan"))
    assert call(via_src, None, 0, src=dev_src.data_ptr(), slot=3) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(direct).all() and torch.equal(direct, via_src)
    assert lib.maua_upconv_blur_ok(cin, cout, h, w) == 1 and lib.maua_upconv_blur_ok(512, 256, 64, 64) == 0  # (cin <= 256: LDS budget)
    assert lib.maua_upconv_blur_ok(60, 32, 32, 64) == 0 and lib.maua_upconv_blur_ok(64, 32, 30, 64) == 0
    assert call(direct, None, 0, cin_=60) == -38   # MAUA_ENOSYS: the two-launch path serves the shape
    assert n_ws > 0 and call(direct, None, 0, ws_=None) == -22  # MAUA_EINVAL: several segments need their seam workspace
    # a tap matrix that is not an outer product: StyledConv keeps the two-launch path (and agrees with the generic FIR)
    odd = k.clone()
    odd[0, 0] += 0.05
    conv.blur.kernel.copy_(odd)
    assert not conv.blur_is_separable()
    keep = StyledConv.fused_blur_min_width
    try:
        StyledConv.fused_blur_min_width = 32
        launches = []
        real = conv.run
        conv.run = lambda *a, _real=real, **kw: (launches.append(1), _real(*a, **kw))[1]
        m(x, torch.from_numpy(r.standard_normal((b, 512)).astype(np.float32)).to(gpu), noise=nz_seq[:b])
        conv.run = real
        assert launches, "a non-separable blur must not reach the fused kernel"
    finally:
        StyledConv.fused_blur_min_width = keep


# ---- round 6: the style fold (include/maua_hip.h THE STYLE FOLD) — producers store their map multiplied by the consumer's styles -----------


def _fold_chain(gpu, c0, c1, c2, h, w, batch, fused_min_width, seed):
    """up(c0 -> c1) -> plain(c1 -> c1) + ToRGB -> up(c1 -> c2) as three StyledConv modules + one ToRGB with seeded weights; returns
    (modules, oracle state dict, inputs)."""
    from maua_stylegan2_amd.models.stylegan2 import StyledConv, ToRGB

    r = np.random.default_rng(seed)
    f32 = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32))  # noqa: E731
    mods, sd = {}, {}
    for name, (ci, co, up) in {"A": (c0, c1, True), "B": (c1, c1, False), "C": (c1, c2, True)}.items():
        m = StyledConv(ci, co, 3, 512, upsample=up)
        part = {"conv.weight": f32(r.standard_normal((1, co, ci, 3, 3))), "conv.modulation.weight": f32(r.standard_normal((ci, 512))),
                "conv.modulation.bias": f32(1 + 0.1 * r.standard_normal(ci)), "noise.weight": f32([0.37]),
                "activate.bias": f32(0.3 * r.standard_normal(co))}
        if up:
            part["conv.blur.kernel"] = m.conv.blur.kernel.clone()
        m.load_state_dict(part, strict=True)
        m.fused_blur_min_width = fused_min_width
        mods[name] = m.to(gpu)
        sd.update({f"{name}.{k}": v for k, v in part.items()})
    t_rgb = ToRGB(c1, 512)
    part = {"bias": f32(0.3 * r.standard_normal((1, 3, 1, 1))), "upsample.kernel": f32(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)),
            "conv.weight": f32(r.standard_normal((1, 3, c1, 1, 1))), "conv.modulation.weight": f32(r.standard_normal((c1, 512))),
            "conv.modulation.bias": f32(1 + 0.1 * r.standard_normal(c1))}
    t_rgb.load_state_dict(part, strict=True)
    mods["T"] = t_rgb.to(gpu)
    sd.update({f"T.{k}": v for k, v in part.items()})
    inputs = dict(x=f32(r.standard_normal((batch, c0, h, w))), lat=f32(r.standard_normal((batch, 4, 512))),
                  nzA=f32(r.standard_normal((batch, 1, 2 * h, 2 * w))), nzB=f32(r.standard_normal((batch, 1, 2 * h, 2 * w))),
                  nzC=f32(r.standard_normal((batch, 1, 4 * h, 4 * w))), skip=f32(r.standard_normal((batch, 3, h, w))))
    return mods, sd, inputs


@pytest.mark.parametrize("c0,c1,c2,h,w,batch,fused_min_width", [
    (64, 128, 64, 16, 32, 2, 1 << 30),   # tail -> modconv_w2d_kernel<4,2,2,PRE> (two m-tiles: partial ToRGB sums) -> modconv_up2d_kernel<8,0,PRE> + tail
    (64, 32, 32, 32, 32, 2, 32),         # fused up-sampling layer -> modconv_w2dw_kernel<PRE> (ToRGB fused) -> fused up-sampling layer <8,2,PRE>
    (128, 64, 32, 16, 32, 1, 1 << 30),   # tail -> modconv_w2d_kernel<4,2,2,PRE> (one m-tile, ToRGB fused) -> mode 6 PRE + tail
    (64, 64, 32, 24, 32, 3, 32),         # fused -> <4,2,2,PRE> -> fused (h not a power of two, three images)
    (36, 32, 32, 8, 32, 1, 1 << 30),     # modconv_up2d_kernel<4,...> (cin % 8 != 0) producer side: tail -> w2dw -> <8,0,PRE>
])
def test_style_fold_chain_equals_unfolded_chain_and_oracle(gpu, c0, c1, c2, h, w, batch, fused_min_width):
    """reference models/stylegan2.py:220-221 reassociated: every producer stores act(..) * s_next (post_s), the consumer runs the kernel
    instance without its style multiplies (s == NULL).  Three layers + ToRGB through StyledConv.run, folded against un-folded (same
    kernels otherwise: 2e-5 of the output scale — one rounding per element moves) and against the oracle's chain (the layer tests'
    tolerance); the ToRGB image must not see the fold at all (it is computed from the un-scaled value)."""
    from maua_stylegan2_amd.models.stylegan2 import _style_table
    from oracle import stylegan2_oracle as so

    mods, sd, inp = _fold_chain(gpu, c0, c1, c2, h, w, batch, fused_min_width, seed=c0 + c1 + c2 + h + w)
    A, B, C, T = mods["A"], mods["B"], mods["C"], mods["T"]
    lib = _lib.load()
    assert B.accepts_prescaled(2 * h, 2 * w) and C.accepts_prescaled(2 * h, 2 * w)
    entries, s_off, d_off = [], 0, 0
    for m, li in ((A.conv, 0), (B.conv, 1), (T.conv, 2), (C.conv, 3)):
        entries.append(m.table_entry(li, s_off, d_off))
        s_off += m.in_channel
        d_off += batch * m.out_channel if m.demodulate else 0
    table = _style_table(entries, gpu)
    styles = torch.empty(batch, s_off, device=gpu)
    demod = torch.empty(max(d_off, 1), device=gpu)
    st = _lib.stream_ptr(gpu)
    lat = inp["lat"].to(gpu).contiguous()
    _lib.check(lib.maua_style_affine_f32(lat.data_ptr(), batch, 4, 512, None, None, table.data_ptr(), 4, max(e["cin"] for e in entries),
                                         styles.data_ptr(), s_off, None, st), "affine")
    _lib.check(lib.maua_demod_f32(table.data_ptr(), 4, max(e["cout"] for e in entries), styles.data_ptr(), s_off, demod.data_ptr(), batch, st),
               "demod")
    dm = lambda e: demod[e["d_off"]: e["d_off"] + batch * e["cout"]].view(batch, e["cout"])  # noqa: E731
    g = {k: v.to(gpu) for k, v in inp.items()}

    def chain(fold):
        keep = {}

        def bufs(name, shape):
            keep[name] = torch.full(shape, float("nan"), device=gpu)
            return keep[name]

        a = A.run(g["x"], styles, entries[0]["s_off"], dm(entries[0]), g["nzA"], bufs, "A", post_off=entries[1]["s_off"] if fold else None)
        assert A.posted == fold and A.last_path == ("fused" if fused_min_width <= w and lib.maua_upconv_blur_ok(c0, c1, h, w) else "pair")
        img = torch.full((batch, 3, 2 * h, 2 * w), float("nan"), device=gpu)
        fuse = dict(module=T, s_off=entries[2]["s_off"], skip=g["skip"], out=img, store=True)
        bb = B.run(a, styles, entries[1]["s_off"], dm(entries[1]), g["nzB"], bufs, "B", rgb=fuse, prescaled=fold,
                   post_off=entries[3]["s_off"] if fold else None)
        assert fuse.get("done") and B.posted == fold
        c = C.run(bb, styles, entries[3]["s_off"], dm(entries[3]), g["nzC"], bufs, "C", prescaled=fold)
        assert not C.posted
        torch.cuda.synchronize()
        return a.clone(), bb.clone(), img.clone(), c.clone()

    a0, b0, img0, out0 = chain(False)
    a1, b1, img1, out1 = chain(True)
    for tns in (a0, b0, img0, out0, a1, b1, img1, out1):
        assert torch.isfinite(tns).all()  # (NaN-prefilled buffers: every element written)
    sB = styles[:, entries[1]["s_off"]: entries[1]["s_off"] + c1][:, :, None, None]
    sC = styles[:, entries[3]["s_off"]: entries[3]["s_off"] + c1][:, :, None, None]
    # the stored maps of the folded chain are the un-folded ones times the consumer's styles
    np.testing.assert_allclose(a1.cpu().numpy(), (a0 * sB).cpu().numpy(), atol=2e-6 * float((a0 * sB).abs().max()), rtol=1e-6)
    scale = float(out0.abs().max())
    assert float((b1 - b0 * sC).abs().max()) <= 2e-5 * float((b0 * sC).abs().max())
    assert float((img1 - img0).abs().max()) <= 2e-5 * float(img0.abs().max())
    assert float((out1 - out0).abs().max()) <= 2e-5 * scale, float((out1 - out0).abs().max()) / scale
    # ... and the whole chain against the oracle
    wa = so.styled_conv(sd, "A", inp["x"], inp["lat"][:, 0], inp["nzA"], True)
    wb = so.styled_conv(sd, "B", wa, inp["lat"][:, 1], inp["nzB"], False)
    wimg = so.to_rgb(sd, "T", wb, inp["lat"][:, 2], inp["skip"])
    wc = so.styled_conv(sd, "C", wb, inp["lat"][:, 3], inp["nzC"], True)
    np.testing.assert_allclose(img1.cpu().numpy(), wimg.numpy(), atol=2e-3, rtol=1e-4)
    np.testing.assert_allclose(out1.cpu().numpy(), wc.numpy(), atol=1e-3, rtol=1e-4)
    print(f"[style fold {c0}->{c1}->{c2} @{h}x{w}] folded vs un-folded {float((out1 - out0).abs().max()) / scale:.2e} of the output scale; "
          f"vs oracle {float((out1.cpu() - wc).abs().max()):.2e} abs at scale {scale:.1f}")


def test_style_fold_abi_contract(gpu):
    """s == NULL / post_s outside the kernels that implement them is MAUA_ENOSYS (include/maua_hip.h), never a silently un-scaled result."""
    lib = _lib.load()
    st = _lib.stream_ptr(gpu)
    b, cin, cout, h, w = 1, 32, 32, 16, 32
    x = torch.randn(b, cin, h, w, device=gpu)
    y = torch.empty(b, cout, 2 * h + 1, 2 * w + 1, device=gpu)
    wp = torch.randn(24 * cin * cout + 64, device=gpu)
    ws = torch.empty(1 << 16, device=gpu)
    for mode in (0, 1, 2, 3, 4, 7, 8):
        rc = lib.maua_modconv3x3_f32(x.data_ptr(), wp.data_ptr(), None, cin, None, y.data_ptr(), b, cin, cout, h, w, mode, 1.0, 0, None, 0,
                                     None, None, ws.data_ptr(), None, 0, st)
        assert rc == -38, (mode, rc)
    s = torch.randn(b, cin, device=gpu)
    rgb_w, rgb_b, img = torch.randn(3, cout, device=gpu), torch.zeros(3, device=gpu), torch.empty(b, 3, h, w, device=gpu)
    nw, bias = torch.zeros(1, device=gpu), torch.zeros(cout, device=gpu)
    for mode in (0, 2, 3):  # post_s: the 2-D Winograd kernels only
        rc = lib.maua_styledconv_torgb_f32(x.data_ptr(), wp.data_ptr(), s.data_ptr(), cin, None, y.data_ptr(), b, cin, cout, h, w, mode, 1.0,
                                           None, 0, nw.data_ptr(), bias.data_ptr(), rgb_w.data_ptr(), s.data_ptr(), 1.0, rgb_b.data_ptr(), None,
                                           None, img.data_ptr(), 1, None, None, 0, s.data_ptr(), st)
        assert rc == -38, (mode, rc)
    torch.cuda.synchronize()


@pytest.mark.parametrize("cin,cout,h,w,up", [(36, 32, 8, 32, True), (64, 64, 16, 32, True), (36, 64, 16, 32, False), (32, 32, 32, 64, False),
                                             (28, 32, 8, 32, False)])
def test_prescaled_instances_equal_the_style_multiplying_ones(gpu, cin, cout, h, w, up):
    """maua_modconv3x3_f32 with s == NULL on x * s against the same call with (x, s): modconv_up2d_kernel<4 | 8, 0, true> (+ its edge
    lines and exported column), modconv_w2d_kernel<4,2,2,true>, <2,2,3,true> and modconv_w2dw_kernel<true>."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    r = np.random.default_rng(cin + cout + h + w)
    m = ModulatedConv2d(cin, cout, 3, 512, upsample=up).to(gpu)
    mode = m.conv_mode(h, w)
    assert mode == (6 if up else 5)
    lib = _lib.load()
    b = 2
    x = torch.from_numpy(r.standard_normal((b, cin, h, w)).astype(np.float32)).to(gpu)
    s = torch.from_numpy((1 + 0.5 * r.standard_normal((b, cin))).astype(np.float32)).to(gpu)
    d = torch.from_numpy((0.5 + r.random((b, cout))).astype(np.float32)).to(gpu)
    shape = (b, cout, 2 * h + 1, 2 * w + 1) if up else (b, cout, h, w)
    n_ws = lib.maua_modconv_ws_floats(b, cin, cout, h, w, mode)
    ws = torch.empty(max(n_ws, 1), device=gpu)
    a, p = torch.full(shape, float("nan"), device=gpu), torch.full(shape, float("nan"), device=gpu)
    m.run(x, s, 0, d, a, ws)
    name_a = _lib.last_modconv_instance()
    m.run((x * s[:, :, None, None]).contiguous(), s, 0, d, p, ws, prescaled=True)
    name_p = _lib.last_modconv_instance()
    torch.cuda.synchronize()
    # the same template instance but for its PRE argument ("..., false>" / "..., true>" on the 2-D Winograd kernels, "..., false, 32>" / "..., true, 32>" on
    # the F(2,2)^2 kernel, whose last argument is the tile width)
    assert "false" in name_a and "true" in name_p and name_a.replace("false", "true") == name_p, (name_a, name_p)
    assert torch.isfinite(a).all() and torch.isfinite(p).all()
    assert float((a - p).abs().max()) <= 2e-5 * float(a.abs().max())
