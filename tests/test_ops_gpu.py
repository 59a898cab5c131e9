"""GPU parity: the HIP upfirdn2d / fused_bias_act (through the C ABI) vs the golden vectors and the oracle."""
import numpy as np
import pytest
import torch

from oracle import ops_oracle

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
TOL = 1e-5  # fp32 FIR sums of <= 16 O(1) terms; north_star budget is 1e-3


def t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_upfirdn2d_golden(gpu, golden):
    from maua_stylegan2_amd.op import upfirdn2d

    g = golden("ops_upfirdn2d.npz")
    for name in g["cases"]:
        up, down, p0, p1 = (int(v) for v in g[f"{name}.cfg"])
        y = upfirdn2d(t(g[f"{name}.x"], gpu), t(g[f"{name}.k"], gpu), up=up, down=down, pad=(p0, p1))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"{name}.y"], atol=TOL, err_msg=str(name))


@pytest.mark.parametrize("shape,k,pad", [
    ((1, 2, 65, 65), 4, (1, 1)), ((2, 3, 129, 129), 4, (1, 1)), ((1, 4, 257, 300), 4, (1, 1)),
    ((1, 2, 100, 37), 3, (1, 1)), ((1, 1, 70, 513), 2, (1, 0)), ((3, 1, 33, 33), 4, (2, 2)),
    ((1, 2, 40, 40), 4, (-1, 0)), ((1, 1, 1025, 1025), 4, (1, 1)),
    # 16-byte path (out_w >= 256, out_w % 4 == 0) with every row phase / partial tiles / odd plane sizes
    ((3, 3, 70, 517), 4, (1, 1)), ((2, 2, 45, 512), 3, (1, 1)), ((1, 5, 33, 260), 2, (1, 0)), ((2, 1, 259, 263), 4, (2, 2)),
    ((1, 3, 257, 257), 4, (1, 1)), ((1, 2, 64, 1028), 4, (0, 3)),
])
def test_upfirdn2d_tiled_path_vs_oracle(gpu, shape, k, pad):
    """Blur-style calls (up = down = 1) across tile-boundary sizes, incl. the 1025^2 -> 1024^2 headline shape."""
    from maua_stylegan2_amd.op import upfirdn2d

    r = np.random.default_rng(sum(shape) + k)
    x = r.standard_normal(shape).astype(np.float32)
    kern = r.standard_normal((k, k)).astype(np.float32)
    want = ops_oracle.upfirdn2d(torch.from_numpy(x), torch.from_numpy(kern), pad=pad).numpy()
    got = upfirdn2d(t(x, gpu), t(kern, gpu), pad=pad).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("up,down,k,pad", [(2, 1, 4, (2, 1)), (1, 2, 4, (1, 1)), (2, 2, 3, (1, 1)), (3, 1, 5, (2, 2))])
def test_upfirdn2d_generic_path_vs_oracle(gpu, up, down, k, pad):
    from maua_stylegan2_amd.op import upfirdn2d

    r = np.random.default_rng(7 + up + 10 * down)
    x = r.standard_normal((2, 3, 37, 41)).astype(np.float32)
    kern = r.standard_normal((k, k)).astype(np.float32)
    want = ops_oracle.upfirdn2d(torch.from_numpy(x), torch.from_numpy(kern), up=up, down=down, pad=pad).numpy()
    got = upfirdn2d(t(x, gpu), t(kern, gpu), up=up, down=down, pad=pad).cpu().numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("shape,k,pad", [
    ((1, 2, 4, 4), 4, (2, 1)), ((2, 3, 64, 64), 4, (2, 1)), ((1, 2, 37, 41), 4, (2, 1)), ((1, 1, 65, 129), 4, (2, 1)),
    ((1, 2, 128, 128), 4, (2, 1)), ((1, 1, 33, 300), 4, (2, 1)),               # one / two / four strips across, partial tiles
    ((1, 2, 20, 67), 4, (1, 2)), ((1, 1, 31, 65), 4, (3, 3)), ((2, 1, 19, 23), 4, (0, 0)), ((1, 1, 40, 40), 4, (-1, 1)),
    ((1, 1, 40, 70), 4, (5, -2)), ((1, 2, 29, 130), 3, (1, 1)), ((1, 2, 17, 66), 2, (1, 0)), ((1, 1, 50, 50), 3, (2, 2)),
    ((1, 1, 16, 129), 1, (0, 0)),
])
def test_upfirdn2d_up2_tiled_path_vs_oracle(gpu, shape, k, pad):
    """Upsample-style calls (up = 2, down = 1; reference models/stylegan2.py:51-67, op/upfirdn2d_kernel.cu:313-359 mode 3): every pad
    parity incl. negative pads (crops), odd output widths (a lone last column), 1 .. 4-tap kernels, sizes across the strip / tile edges."""
    from maua_stylegan2_amd.op import upfirdn2d

    r = np.random.default_rng(sum(shape) + 31 * k + pad[0])
    x = r.standard_normal(shape).astype(np.float32)
    kern = r.standard_normal((k, k)).astype(np.float32)
    want = ops_oracle.upfirdn2d(torch.from_numpy(x), torch.from_numpy(kern), up=2, pad=pad).numpy()
    got = upfirdn2d(t(x, gpu), t(kern, gpu), up=2, pad=pad).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-5)


@pytest.mark.parametrize("shape,k,pad", [
    ((1, 2, 8, 8), 4, (1, 1)), ((2, 3, 128, 128), 4, (1, 1)), ((1, 2, 37, 41), 4, (1, 1)), ((1, 1, 130, 258), 4, (1, 1)),
    ((1, 2, 129, 131), 4, (1, 1)), ((1, 1, 70, 601), 4, (1, 1)),                # one / two / four strips across, partial tiles, odd sizes
    ((1, 2, 40, 67), 4, (2, 1)), ((1, 1, 31, 65), 4, (3, 3)), ((2, 1, 19, 23), 4, (0, 0)), ((1, 1, 40, 40), 4, (-1, 1)),
    ((1, 1, 40, 70), 4, (5, -2)), ((1, 2, 29, 130), 3, (1, 1)), ((1, 2, 17, 66), 2, (0, 0)), ((1, 1, 50, 50), 3, (0, 1)),
    ((1, 1, 16, 129), 1, (0, 0)),
])
def test_upfirdn2d_down2_tiled_path_vs_oracle(gpu, shape, k, pad):
    """Downsample-style calls (up = 1, down = 2; reference models/stylegan2.py:70-84, op/upfirdn2d_kernel.cu:313-359 modes 5 / 6; also the
    backward of Upsample, op/upfirdn2d.py:20-60): pads of both parities incl. negative ones, odd sizes, 1 .. 4-tap kernels, strip / tile edges."""
    from maua_stylegan2_amd.op import upfirdn2d

    r = np.random.default_rng(sum(shape) + 17 * k + pad[0])
    x = r.standard_normal(shape).astype(np.float32)
    kern = r.standard_normal((k, k)).astype(np.float32)
    want = ops_oracle.upfirdn2d(torch.from_numpy(x), torch.from_numpy(kern), down=2, pad=pad).numpy()
    got = upfirdn2d(t(x, gpu), t(kern, gpu), down=2, pad=pad).cpu().numpy()
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-5)


def test_upfirdn2d_down2_inverts_the_shape_of_up2_at_full_size(gpu):
    """Size-independent properties at [32, 1024, 1024] -> [32, 512, 512] (Downsample: pad (1, 1), 4 taps): linearity; a constant map stays
    constant away from the border (taps sum to 1); an impulse lands as the decimated tap matrix."""
    from maua_stylegan2_amd.op import upfirdn2d
    from maua_stylegan2_amd.seeding import fir_kernel_2d

    k = torch.from_numpy(fir_kernel_2d((1, 3, 3, 1), 1.0)).to(gpu)
    gen = torch.Generator(device="cpu").manual_seed(6)
    a = torch.randn(1, 32, 1024, 1024, generator=gen).to(gpu)
    b = torch.randn(1, 32, 1024, 1024, generator=gen).to(gpu)
    ya, yb, yab = upfirdn2d(a, k, down=2, pad=(1, 1)), upfirdn2d(b, k, down=2, pad=(1, 1)), upfirdn2d(2 * a - 3 * b, k, down=2, pad=(1, 1))
    assert ya.shape == (1, 32, 512, 512)
    assert float((yab - (2 * ya - 3 * yb)).abs().max()) < 1e-4
    ones = upfirdn2d(torch.ones(1, 2, 1024, 1024, device=gpu), k, down=2, pad=(1, 1))
    assert float((ones[:, :, 1:-1, 1:-1] - 1.0).abs().max()) < 1e-6
    imp = torch.zeros(1, 1, 1024, 1024, device=gpu)
    imp[0, 0, 600, 822] = 1.0
    kr = torch.arange(16, dtype=torch.float32, device=gpu).reshape(4, 4)
    y = upfirdn2d(imp, kr, down=2, pad=(1, 1))
    # out[oy, ox] = sum kflip[i, j] x[2 oy + i - 1, 2 ox + j - 1]: x[600, 822] is reached with i = 601 - 2 oy, j = 823 - 2 ox in 0 .. 3:
    # oy in {299, 300} (i = 3, 1), ox in {410, 411} (j = 3, 1); kflip[i, j] = k[3 - i, 3 - j]
    want = torch.tensor([[kr[0, 0], kr[0, 2]], [kr[2, 0], kr[2, 2]]], device=gpu)
    assert torch.equal(y[0, 0, 299:301, 410:412], want)
    assert float(y.sum()) == float(want.sum())


def test_upfirdn2d_up2_rectangular_taps_and_axis_pads(gpu):
    """The native-op boundary takes per-axis pads and a kh x kw tap matrix (op/upfirdn2d.cpp:12-22)."""
    from maua_stylegan2_amd.op import upfirdn2d_native_op

    r = np.random.default_rng(77)
    x = r.standard_normal((3, 21, 45, 1)).astype(np.float32)
    kern = r.standard_normal((3, 4)).astype(np.float32)
    got = upfirdn2d_native_op(t(x, gpu), t(kern, gpu), 2, 2, 1, 1, 2, 1, 0, 3).cpu().numpy()[..., 0]
    # oracle by hand: zero-stuff, pad (y: 0 / 3, x: 2 / 1), true convolution
    canvas = np.zeros((3, 42 + 0 + 3, 90 + 2 + 1), dtype=np.float64)
    canvas[:, 0:42:2, 2:92:2] = x[..., 0]
    kf = kern[::-1, ::-1].astype(np.float64)
    want = np.zeros((3, canvas.shape[1] - 2, canvas.shape[2] - 3))
    for i in range(3):
        for j in range(4):
            want += kf[i, j] * canvas[:, i:i + want.shape[1], j:j + want.shape[2]]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, atol=2e-5)


def test_upfirdn2d_up2_full_size_properties(gpu):
    """Size-independent properties at the generator's largest Upsample, [32, 512, 512] -> [32, 1024, 1024] (pad (2, 1), taps x 4):
    linearity; an impulse lands as the tap matrix itself (true convolution of the zero-stuffed canvas); a constant map comes back
    constant away from the border (the four polyphase tap sums are 1 each)."""
    from maua_stylegan2_amd.op import upfirdn2d
    from maua_stylegan2_amd.seeding import fir_kernel_2d

    k = torch.from_numpy(fir_kernel_2d((1, 3, 3, 1), 4.0)).to(gpu)  # (gain 4 = up ** 2: reference models/stylegan2.py:56)
    gen = torch.Generator(device="cpu").manual_seed(5)
    a = torch.randn(1, 32, 512, 512, generator=gen).to(gpu)
    b = torch.randn(1, 32, 512, 512, generator=gen).to(gpu)
    ya, yb, yab = upfirdn2d(a, k, up=2, pad=(2, 1)), upfirdn2d(b, k, up=2, pad=(2, 1)), upfirdn2d(2 * a - 3 * b, k, up=2, pad=(2, 1))
    assert ya.shape == (1, 32, 1024, 1024)
    assert float((yab - (2 * ya - 3 * yb)).abs().max()) < 1e-4
    imp = torch.zeros(1, 1, 512, 512, device=gpu)
    imp[0, 0, 300, 411] = 1.0
    kr = torch.arange(16, dtype=torch.float32, device=gpu).reshape(4, 4)
    y = upfirdn2d(imp, kr, up=2, pad=(2, 1))
    # canvas (2 * 300, 2 * 411) reaches out[oy, ox] through flipped tap (i, j) = (600 + 2 - oy, 822 + 2 - ox): y[599 + a, 821 + b] = k[a, b]
    assert torch.equal(y[0, 0, 599:603, 821:825], kr)
    assert float(y.sum()) == float(kr.sum())
    ones = upfirdn2d(torch.ones(1, 2, 512, 512, device=gpu), k, up=2, pad=(2, 1))
    assert float((ones[:, :, 2:-2, 2:-2] - 1.0).abs().max()) < 1e-6


def test_upfirdn2d_native_boundary_minor_axis(gpu):
    """The pybind-level layout [major, h, w, minor] with minor > 1 (op/upfirdn2d.cpp:12-22)."""
    from maua_stylegan2_amd.op import upfirdn2d_native_op

    r = np.random.default_rng(3)
    x = r.standard_normal((3, 9, 10, 2)).astype(np.float32)
    kern = r.standard_normal((4, 4)).astype(np.float32)
    got = upfirdn2d_native_op(t(x, gpu), t(kern, gpu), 2, 2, 1, 1, 2, 1, 2, 1).cpu().numpy()
    xx = torch.from_numpy(x).permute(0, 3, 1, 2)  # [major, minor, h, w]
    want = ops_oracle.upfirdn2d(xx, torch.from_numpy(kern), up=2, pad=(2, 1)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(got, want, atol=2e-5)


def test_upfirdn2d_linearity_and_shift_full_size(gpu):
    """Size-independent properties at the BASELINE shape [32,1025,1025]: linearity, and an impulse reproduces the
    flipped... i.e. the true-convolution taps at the right place."""
    from maua_stylegan2_amd.op import upfirdn2d
    from maua_stylegan2_amd.seeding import fir_kernel_2d

    k = torch.from_numpy(fir_kernel_2d((1, 3, 3, 1), 4.0)).to(gpu)
    gen = torch.Generator(device="cpu").manual_seed(0)
    a = torch.randn(1, 32, 1025, 1025, generator=gen).to(gpu)
    b = torch.randn(1, 32, 1025, 1025, generator=gen).to(gpu)
    ya, yb, yab = upfirdn2d(a, k, pad=(1, 1)), upfirdn2d(b, k, pad=(1, 1)), upfirdn2d(2 * a - 3 * b, k, pad=(1, 1))
    assert ya.shape == (1, 32, 1024, 1024)
    assert float((yab - (2 * ya - 3 * yb)).abs().max()) < 1e-4
    imp = torch.zeros(1, 1, 1025, 1025, device=gpu)
    imp[0, 0, 500, 700] = 1.0
    kr = torch.arange(16, dtype=torch.float32, device=gpu).reshape(4, 4)
    y = upfirdn2d(imp, kr, pad=(1, 1))
    # out[oy,ox] = sum kflip[i,j] x[oy+i-1, ox+j-1]  ->  y[500+1-i, 700+1-j] = kflip[i,j] = k[3-i,3-j]
    patch = y[0, 0, 498:502, 698:702]
    assert torch.equal(patch, kr)
    assert float(y.sum()) == float(kr.sum())


def test_fused_leaky_relu_golden(gpu, golden):
    from maua_stylegan2_amd.op import fused_leaky_relu

    g = golden("ops_fused_leaky_relu.npz")
    for name in g["cases"]:
        y = fused_leaky_relu(t(g[f"{name}.x"], gpu), t(g[f"{name}.b"], gpu))
        np.testing.assert_allclose(y.cpu().numpy(), g[f"{name}.y"], atol=1e-6, err_msg=str(name))


@pytest.mark.parametrize("act,grad", [(3, 0), (3, 1), (3, 2), (1, 0), (1, 2)])
def test_fused_bias_act_switch(gpu, act, grad):
    from maua_stylegan2_amd.op import fused_bias_act

    r = np.random.default_rng(11)
    for shape in [(2, 8, 16, 16), (3, 5, 7)]:
        x = r.standard_normal(shape).astype(np.float32)
        b = r.standard_normal(shape[1]).astype(np.float32)
        ref = r.standard_normal(shape).astype(np.float32)
        want = ops_oracle.fused_bias_act_kernel_semantics(x, b, ref, act, grad, 0.2, 1.5)
        got = fused_bias_act(t(x, gpu), t(b, gpu), t(ref, gpu), act, grad, 0.2, 1.5).cpu().numpy()
        np.testing.assert_allclose(got, want, atol=1e-6)
        want_nb = ops_oracle.fused_bias_act_kernel_semantics(x, None, ref, act, grad, 0.2, 1.5)
        got_nb = fused_bias_act(t(x, gpu), torch.empty(0, device=gpu), t(ref, gpu), act, grad, 0.2, 1.5).cpu().numpy()
        np.testing.assert_allclose(got_nb, want_nb, atol=1e-6)


def test_fused_leaky_relu_full_size_properties(gpu):
    """[1,32,1024,1024]: positive homogeneity f(c*x, c*b) = c*f(x, b) for c > 0, and sign structure."""
    from maua_stylegan2_amd.op import fused_leaky_relu

    gen = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(1, 32, 1024, 1024, generator=gen).to(gpu)
    b = torch.randn(32, generator=gen).to(gpu)
    y = fused_leaky_relu(x, b)
    y2 = fused_leaky_relu(4 * x, 4 * b)
    assert torch.equal(y2, 4 * y)
    pre = x + b.view(1, -1, 1, 1)
    assert torch.equal(y > 0, pre > 0)
    ratio = (y / pre)[pre.abs() > 1e-3]
    assert float((ratio[ratio > 1] - 2 ** 0.5).abs().max()) < 1e-5


def test_frames_to_u8(gpu, golden):
    from maua_stylegan2_amd import _lib

    g = golden("postprocess.npz")
    x = t(g["x"], gpu)
    out = torch.empty((1, 4, 8, 3), dtype=torch.uint8, device=gpu)
    _lib.check(_lib.load().maua_frames_to_u8(x.data_ptr(), out.data_ptr(), 1, 4, 8, _lib.stream_ptr()), "u8")
    assert (out.cpu().numpy() == g["y"]).all()


def test_empty_and_degenerate_inputs(gpu):
    """Edge cases: empty batch, 1x1 planes, a single row / column, output smaller than one tile."""
    from maua_stylegan2_amd.op import fused_leaky_relu, upfirdn2d

    k = torch.ones(4, 4, device=gpu) / 16
    y = upfirdn2d(torch.empty(0, 3, 9, 9, device=gpu), k, pad=(1, 1))
    assert y.shape == (0, 3, 8, 8)
    assert fused_leaky_relu(torch.empty(0, 4, 2, 2, device=gpu), torch.zeros(4, device=gpu)).shape == (0, 4, 2, 2)
    for shape, kk, up, down, pad in [((1, 1, 1, 1), 2, 1, 1, (1, 0)), ((2, 1, 1, 37), 2, 2, 1, (1, 0)), ((1, 2, 41, 1), 3, 1, 1, (1, 1)),
                                     ((1, 1, 4, 4), 4, 1, 1, (0, 0)), ((1, 1, 3, 3), 4, 2, 2, (2, 1))]:
        r = np.random.default_rng(sum(shape))
        x = r.standard_normal(shape).astype(np.float32)
        kern = r.standard_normal((kk, kk)).astype(np.float32)
        want = ops_oracle.upfirdn2d(torch.from_numpy(x), torch.from_numpy(kern), up=up, down=down, pad=pad).numpy()
        got = upfirdn2d(t(x, gpu), t(kern, gpu), up=up, down=down, pad=pad).cpu().numpy()
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, atol=2e-5)
    with pytest.raises(RuntimeError):
        upfirdn2d(torch.zeros(1, 1, 2, 2, device=gpu), k)  # 2 + 0 - 4 -> empty output, as the reference would fail
    with pytest.raises(RuntimeError, match="float16, float32 or float64"):  # half and double are dispatched (next test)
        upfirdn2d(torch.zeros(1, 1, 8, 8, device=gpu, dtype=torch.bfloat16), k)


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 2e-2), (torch.float64, 1e-12)])
def test_native_ops_half_and_double(gpu, dtype, tol):
    """The reference's two native ops dispatch half / float / double (op/upfirdn2d_kernel.cu:313-359,
    op/fused_bias_act_kernel.cu:79): the f16 / f64 entries of the C ABI against the oracle evaluated in fp64."""
    from maua_stylegan2_amd.op import fused_leaky_relu, upfirdn2d
    from oracle import ops_oracle

    r = np.random.default_rng(77)
    for shape, kshape, up, down, pad in [((2, 3, 9, 11), (4, 4), 1, 1, (1, 1)), ((1, 2, 6, 5), (3, 3), 2, 1, (2, 1)),
                                         ((1, 2, 12, 12), (4, 4), 1, 2, (1, 1))]:
        x, k = r.standard_normal(shape), r.standard_normal(kshape)
        want = ops_oracle.upfirdn2d_loops(x, k, up, down, pad)  # the oracle's fp64 scalar-loop definition
        got = upfirdn2d(torch.from_numpy(x).to(gpu, dtype), torch.from_numpy(k).to(gpu, dtype), up=up, down=down, pad=pad)
        assert got.dtype == dtype and got.shape == want.shape
        np.testing.assert_allclose(got.double().cpu().numpy(), want, atol=tol * max(1.0, np.abs(want).max()))
    x, b = r.standard_normal((2, 7, 5, 5)), r.standard_normal(7)
    want = ops_oracle.fused_leaky_relu(torch.from_numpy(x), torch.from_numpy(b)).numpy()  # torch fp64
    assert want.dtype == np.float64
    got = fused_leaky_relu(torch.from_numpy(x).to(gpu, dtype), torch.from_numpy(b).to(gpu, dtype))
    assert got.dtype == dtype
    np.testing.assert_allclose(got.double().cpu().numpy(), want, atol=tol * 4)
    with pytest.raises(RuntimeError, match="float16, float32 or float64"):
        upfirdn2d(torch.zeros(1, 1, 4, 4, dtype=torch.int32, device=gpu), torch.ones(2, 2, device=gpu))
