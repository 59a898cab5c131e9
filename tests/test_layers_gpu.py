"""GPU parity: ModulatedConv2d / StyledConv / ToRGB on the HIP path vs golden vectors from the reference."""
import numpy as np
import pytest
import torch

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


@pytest.mark.parametrize("cin,cout,hw,up,batch", [
    (512, 512, 4, False, 3), (512, 512, 8, True, 2), (512, 512, 16, False, 1), (512, 256, 32, True, 1),
    (128, 128, 64, False, 2), (128, 64, 64, True, 1), (64, 64, 128, False, 1), (64, 32, 96, True, 1),
    (32, 32, 160, False, 2), (40, 24, 20, False, 1), (24, 72, 12, True, 2), (16, 8, 128, True, 2), (8, 40, 130, True, 1),
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
    m.conv.winograd43_min_cout = 1 << 30  # this test pins the F(2,3) mode
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
