"""GPU: red-zone (canary) checks of every matrix-core kernel family through the C ABI.

Device AddressSanitizer refuses exactly these kernels ("instrumented kernel exceeds a launch resource", profiles/r05_asan.txt leg 4): the
ones with hand-built buffer descriptors, LDS-DMA operand staging, 32-bit store offsets and caller-owned workspaces (split-K slabs `ws`,
the exported input column `xcol`, the seam rows `hbuf`, partial-RGB planes).  What replaces it:

  * EVERY operand, output and workspace buffer of a call is a window of exactly the size the header promises inside a larger allocation
    whose surroundings (4096 floats either side) hold a NaN with a recognisable payload;
  * outputs are pre-filled with the same canary;
  * after the launch the red zones must be bit-identical (no write outside any buffer), no output element may still be the canary (every
    element written) and every output must be finite (an operand read outside its buffer multiplies a NaN into the result — the padding
    of these kernels comes from out-of-range descriptor offsets, which return 0, never from reading a neighbour);
  * results are compared with the oracle where the call is a whole layer, with the un-guarded call otherwise.

Shapes: the minimum each kernel accepts, odd / non-power-of-two tile counts, and the generator's own large shapes (one image)."""
import numpy as np
import pytest
import torch

from maua_stylegan2_amd import _lib, seeding

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

RED = 4096
CANARY_BITS = 0x7FC0BEEF  # a quiet NaN with a payload nothing computes


class Guard:
    """Windows of exact size between red zones, all in one registry so that one call checks every buffer of a launch."""

    def __init__(self, dev):
        self.dev = dev
        self.items = []

    def _alloc(self, n, dtype):
        assert dtype in (torch.float32, torch.uint8)
        if dtype == torch.uint8:
            n_f = (n + 3) // 4
        else:
            n_f = n
        raw = torch.full((RED + n_f + RED,), CANARY_BITS, dtype=torch.int32, device=self.dev)
        return raw, n_f

    def inp(self, t, name):
        """A copy of ``t`` (any shape, fp32) in a guarded window."""
        t = t.to(self.dev, torch.float32).contiguous()
        raw, n_f = self._alloc(t.numel(), torch.float32)
        view = raw[RED: RED + n_f].view(torch.float32)
        view.copy_(t.reshape(-1))
        self.items.append((name, raw, n_f, False))
        return view.view(t.shape)

    def out(self, shape, name, dtype=torch.float32):
        """An output / workspace window, pre-filled with the canary."""
        n = int(np.prod(shape))
        raw, n_f = self._alloc(n, dtype)
        self.items.append((name, raw, n_f, True))
        if dtype == torch.uint8:
            return raw[RED: RED + n_f].view(torch.uint8)[:n].view(shape)
        return raw[RED: RED + n_f].view(torch.float32).view(shape)

    def check(self, written=()):
        """Red zones intact everywhere; the outputs named in ``written`` hold no canary and only finite values."""
        torch.cuda.synchronize(self.dev)
        for name, raw, n_f, is_out in self.items:
            lo, hi = raw[:RED], raw[RED + n_f:]
            assert bool((lo == CANARY_BITS).all()), f"{name}: write BELOW the buffer ({int((lo != CANARY_BITS).sum())} dwords)"
            assert bool((hi == CANARY_BITS).all()), f"{name}: write BEYOND the buffer ({int((hi != CANARY_BITS).sum())} dwords)"
            if is_out and name in written:
                body = raw[RED: RED + n_f]
                assert not bool((body == CANARY_BITS).any()), f"{name}: {int((body == CANARY_BITS).sum())} elements never written"
                assert bool(torch.isfinite(body.view(torch.float32)).all()), f"{name}: non-finite output (an operand was read outside its buffer?)"


def _layer(cin, cout, up, seed, dev):
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    r = np.random.default_rng(seed)
    m = ModulatedConv2d(cin, cout, 3, 512, upsample=up)
    m.weight.copy_(torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)))
    return m.to(dev), r


def _packed(m, mode, g, name="wp"):
    wp = m.packed_wino(mode) if mode >= 2 else m.packed()[0]
    torch.cuda.synchronize()
    return g.inp(wp.reshape(-1), name)


def _direct_conv(x, s, d, w, up):
    """fp64 reference of the shared-weight formulation: conv(x * s, W) * wscale * d (transposed, stride 2, for up)."""
    import torch.nn.functional as F

    xs = (x * s[:, :, None, None]).double().cpu()
    wd = w[0].double().cpu()
    if up:
        y = F.conv_transpose2d(xs, wd.transpose(0, 1), stride=2)
    else:
        y = F.conv2d(xs, wd, padding=1)
    scale = 1.0 / np.sqrt(w.shape[2] * 9)
    return (y * scale * d[:, :, None, None].double().cpu()).float()


PLAIN = [  # (mode, cin, cout, h, w, batch)
    (0, 512, 512, 4, 4, 2),      # direct, split-K (the 4^2 layer of every generator): `ws` slabs
    (0, 24, 40, 5, 7, 3),        # direct, odd everything: generic loads, ragged tiles
    (0, 512, 512, 8, 8, 8),      # the generator's 8^2 layer at the bench batch
    (2, 64, 64, 16, 34, 1),      # Winograd F(2,3): W even, not a multiple of the tile
    (2, 512, 512, 16, 16, 2),
    (2, 512, 512, 8, 8, 8),      # ... several images per tile (the generator's 8^2 layer at the bench batch, round 6)
    (3, 64, 96, 8, 36, 2),       # Winograd F(4,3): W % 4 == 0
    (3, 512, 512, 32, 32, 1),
    (5, 4, 32, 16, 32, 1),       # 2-D Winograd, minimum: one K step, wave-complete kernel
    (5, 36, 64, 8, 32, 3),       # <4,2,2>, 9 K steps
    (5, 32, 32, 24, 96, 2),      # <2,2,3> (h % 16 != 0)
    (5, 512, 512, 32, 32, 1),    # generator shape, 8 output-channel tiles
    (5, 32, 32, 1024, 1024, 1),  # the last layer's shape
]


@pytest.mark.parametrize("mode,cin,cout,h,w,batch", PLAIN)
@pytest.mark.parametrize("prescaled", [False, True])
def test_plain_conv_modes_stay_inside_their_buffers(gpu, mode, cin, cout, h, w, batch, prescaled):
    if prescaled and mode != 5:
        pytest.skip("s == NULL exists for modes 5 and 6")
    lib = _lib.load()
    m, r = _layer(cin, cout, False, cin + cout + h + w + mode, gpu)
    g = Guard(gpu)
    x_ = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s_ = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, cin))).astype(np.float32))
    d_ = torch.from_numpy((0.5 + r.random((batch, cout))).astype(np.float32))
    x = g.inp(x_ * s_[:, :, None, None] if prescaled else x_, "x")
    s, d = g.inp(s_, "s"), g.inp(d_, "d")
    wp = _packed(m, mode, g)
    y = g.out((batch, cout, h, w), "y")
    n_ws = lib.maua_modconv_ws_floats(batch, cin, cout, h, w, mode)
    ws = g.out((n_ws,), "ws") if n_ws else None
    rc = lib.maua_modconv3x3_f32(x.data_ptr(), wp.data_ptr(), None if prescaled else s.data_ptr(), cin, d.data_ptr(), y.data_ptr(), batch, cin,
                                 cout, h, w, mode, float(m.scale), 0, None, 0, None, None, _lib.ptr(ws), None, 0, _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",))
    want = _direct_conv(x_, s_, d_, m.weight.cpu(), False)
    tol = 3e-4 * float(want.abs().max()) if h * w <= 65536 else 1e-3 * float(want.abs().max())
    assert float((y.cpu() - want).abs().max()) <= tol


UP = [  # (mode, cin, cout, h, w, batch)
    (1, 512, 512, 4, 4, 2),      # polyphase, split-K
    (1, 24, 40, 5, 7, 3),        # generic loads, ragged
    (4, 64, 32, 48, 96, 1),      # F(2,2) on the even x-phase
    (4, 128, 64, 64, 64, 2),
    (6, 4, 32, 8, 32, 1),        # F(2,2)^2, minimum: CC = 4, one K step; xcol export + edge lines
    (6, 36, 32, 8, 32, 2),       # CC = 4, 9 K steps
    (6, 512, 512, 32, 32, 1),    # generator shape
    (6, 64, 32, 512, 512, 1),    # the largest transposed layer
]


@pytest.mark.parametrize("mode,cin,cout,h,w,batch", UP)
@pytest.mark.parametrize("prescaled", [False, True])
def test_transposed_conv_modes_stay_inside_their_buffers(gpu, mode, cin, cout, h, w, batch, prescaled):
    if prescaled and mode != 6:
        pytest.skip("s == NULL exists for modes 5 and 6")
    lib = _lib.load()
    m, r = _layer(cin, cout, True, cin + cout + h + w + mode, gpu)
    g = Guard(gpu)
    x_ = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s_ = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, cin))).astype(np.float32))
    d_ = torch.from_numpy((0.5 + r.random((batch, cout))).astype(np.float32))
    x = g.inp(x_ * s_[:, :, None, None] if prescaled else x_, "x")
    s, d = g.inp(s_, "s"), g.inp(d_, "d")
    wp = _packed(m, mode, g)
    y = g.out((batch, cout, 2 * h + 1, 2 * w + 1), "y")
    n_ws = lib.maua_modconv_ws_floats(batch, cin, cout, h, w, mode)
    ws = g.out((n_ws,), "ws") if n_ws else None
    rc = lib.maua_modconv3x3_f32(x.data_ptr(), wp.data_ptr(), None if prescaled else s.data_ptr(), cin, d.data_ptr(), y.data_ptr(), batch, cin,
                                 cout, h, w, mode, float(m.scale), 0, None, 0, None, None, _lib.ptr(ws), None, 0, _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",) + (("ws",) if mode == 6 else ()))  # (mode 6: the exported last input column fills its workspace exactly)
    want = _direct_conv(x_, s_, d_, m.weight.cpu(), True)
    assert float((y.cpu() - want).abs().max()) <= 3e-4 * float(want.abs().max())


FUSED = [  # (cin, cout, h, w, batch, noise_batch)
    (8, 32, 8, 32, 1, 1),        # minimum: one K step, two y tiles, no seam
    (64, 32, 40, 96, 2, 1),      # odd tile counts, shared noise map, several segments: seam rows through `hbuf`
    (256, 64, 24, 64, 1, 0),     # the longest K loop the kernel accepts, no noise
    (128, 64, 256, 256, 1, 1),   # convs.12
    (64, 32, 512, 512, 1, 1),    # convs.14
]


@pytest.mark.parametrize("cin,cout,h,w,batch,noise_batch", FUSED)
@pytest.mark.parametrize("fold", [False, True])
def test_fused_upsampling_layer_stays_inside_its_buffers(gpu, cin, cout, h, w, batch, noise_batch, fold):
    """maua_upconv_blur_f32 (transposed conv + blur + noise + bias + act in one kernel + the seam pass): x, packed weight, styles, demod,
    taps, noise, bias, the seam workspace and the output, all guarded; with the style fold on both sides (s == NULL, post_s)."""
    from oracle import ops_oracle

    lib = _lib.load()
    m, r = _layer(cin, cout, True, cin + cout + h + w, gpu)
    assert lib.maua_upconv_blur_ok(cin, cout, h, w)
    g = Guard(gpu)
    stride = max(cin, cout)  # s and post_s are rows of ONE styles buffer [B, s_stride] (include/maua_hip.h): this layer's slice / the consumer's
    x_ = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s_row = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, stride))).astype(np.float32))
    s_ = s_row[:, :cin]
    d_ = torch.from_numpy((0.5 + r.random((batch, cout))).astype(np.float32))
    post_row = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, stride))).astype(np.float32))
    post_ = post_row[:, :cout]
    bias_ = torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32))
    nz_ = torch.from_numpy(r.standard_normal((max(noise_batch, 1), 1, 2 * h, 2 * w)).astype(np.float32))
    x = g.inp(x_ * s_[:, :, None, None] if fold else x_, "x")
    s, d, post, bias = g.inp(s_row, "s"), g.inp(d_, "d"), g.inp(post_row, "post_s"), g.inp(bias_, "bias")
    k4 = g.inp(m.blur.kernel, "k4")
    nw = g.inp(torch.tensor([0.37]), "noise_w")
    nz = g.inp(nz_, "noise") if noise_batch else None
    wq = _packed(m, 6, g, "wq")
    y = g.out((batch, cout, 2 * h, 2 * w), "y")
    n_seam = lib.maua_upconv_blur_ws_floats(batch, cin, cout, h, w)
    seam = g.out((n_seam,), "hbuf") if n_seam else None
    nstride = 0 if noise_batch <= 1 else 4 * h * w
    rc = lib.maua_upconv_blur_f32(x.data_ptr(), wq.data_ptr(), None if fold else s.data_ptr(), stride, d.data_ptr(), y.data_ptr(), _lib.ptr(seam),
                                  k4.data_ptr(), _lib.ptr(nz), nstride, nw.data_ptr(), bias.data_ptr(), None, 0, batch, cin, cout, h, w,
                                  float(m.scale), post.data_ptr() if fold else None, _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",))
    # reference: raw transposed conv (fp64) -> blur -> noise -> bias -> leaky ReLU * sqrt2 (-> * post_s)
    raw = _direct_conv(x_, s_, d_, m.weight.cpu(), True)
    blurred = ops_oracle.upfirdn2d(raw, m.blur.kernel.cpu(), up=1, down=1, pad=(1, 1))
    t = blurred + (0.37 * nz_ if noise_batch else 0.0) + bias_[None, :, None, None]
    want = torch.where(t > 0, t, 0.2 * t) * 2 ** 0.5
    if fold:
        want = want * post_[:, :, None, None]
    assert float((y.cpu() - want).abs().max()) <= 3e-4 * float(want.abs().max())


TAIL = [(512, 9, 9, 2, 2), (40, 11, 15, 3, 1), (128, 257, 257, 1, 1), (32, 1025, 1025, 1, 1)]  # (channels, in_h, in_w, batch, noise_batch)


@pytest.mark.parametrize("channels,in_h,in_w,batch,noise_batch", TAIL)
def test_blur_tail_stays_inside_its_buffers(gpu, channels, in_h, in_w, batch, noise_batch):
    """maua_blur_noise_act_f32 on the raw (2H+1) x (2W+1) map (rows only 4-byte aligned: dword buffer accesses with per-plane
    descriptors) incl. the post scale of the style fold."""
    from oracle import ops_oracle

    lib = _lib.load()
    r = np.random.default_rng(channels + in_h)
    g = Guard(gpu)
    x_ = torch.from_numpy(r.standard_normal((batch, channels, in_h, in_w)).astype(np.float32))
    gain_ = torch.from_numpy((0.5 + r.random((batch, channels))).astype(np.float32))
    post_ = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, channels))).astype(np.float32))
    bias_ = torch.from_numpy((0.3 * r.standard_normal(channels)).astype(np.float32))
    k_ = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0))
    oh, ow = in_h - 1, in_w - 1
    nz_ = torch.from_numpy(r.standard_normal((noise_batch, 1, oh, ow)).astype(np.float32))
    x, gain, post, bias, k, nz = g.inp(x_, "x"), g.inp(gain_, "gain"), g.inp(post_, "post_s"), g.inp(bias_, "bias"), g.inp(k_, "k"), g.inp(nz_, "noise")
    nw = g.inp(torch.tensor([0.41]), "noise_w")
    y = g.out((batch, channels, oh, ow), "y")
    rc = lib.maua_blur_noise_act_f32(x.data_ptr(), k.data_ptr(), y.data_ptr(), batch, channels, in_h, in_w, 4, 4, 1, 1, gain.data_ptr(),
                                     nz.data_ptr(), 0 if noise_batch == 1 else oh * ow, nw.data_ptr(), bias.data_ptr(), None, 0,
                                     post.data_ptr(), channels, _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",))
    t = ops_oracle.upfirdn2d(x_, k_, up=1, down=1, pad=(1, 1)) * gain_[:, :, None, None] + 0.41 * nz_ + bias_[None, :, None, None]
    want = torch.where(t > 0, t, 0.2 * t) * 2 ** 0.5 * post_[:, :, None, None]
    assert float((y.cpu() - want).abs().max()) <= 1e-4 * float(want.abs().max())


RGB = [  # (mode, cin, cout, h, w, batch, skip, u8)
    (0, 32, 32, 256, 256, 4, True, False),   # direct kernel with the ToRGB epilogue (grids large enough for no split-K: the entry owns no workspace)
    (3, 64, 64, 512, 512, 1, True, False),
    (5, 32, 32, 16, 32, 2, True, True),      # wave-complete kernel: uint8 frames + fp32 planes
    (5, 64, 64, 24, 32, 1, False, False),    # <4,2,2>, one output-channel tile
    (5, 32, 32, 1024, 1024, 1, True, True),  # the last layer
]


@pytest.mark.parametrize("mode,cin,cout,h,w,batch,with_skip,u8", RGB)
def test_torgb_fused_layers_stay_inside_their_buffers(gpu, mode, cin, cout, h, w, batch, with_skip, u8):
    """maua_styledconv_torgb_f32: feature map, RGB planes, uint8 frames (3 bytes per pixel: the only non-dword-sized rows of the path),
    skip image, taps, all guarded."""
    lib = _lib.load()
    m, r = _layer(cin, cout, False, cin + cout + h + w, gpu)
    g = Guard(gpu)
    f = lambda *shape: torch.from_numpy(r.standard_normal(shape).astype(np.float32))  # noqa: E731
    x, s, d = g.inp(f(batch, cin, h, w), "x"), g.inp(1 + 0.3 * f(batch, cin), "s"), g.inp(0.5 + torch.rand(batch, cout), "d")
    nz, nw, bias = g.inp(f(batch, 1, h, w), "noise"), g.inp(torch.tensor([0.2]), "noise_w"), g.inp(0.3 * f(cout), "bias")
    rgb_w, rgb_s, rgb_b = g.inp(f(3, cout), "rgb_w"), g.inp(1 + 0.3 * f(batch, cin), "rgb_s"), g.inp(0.3 * f(3), "rgb_bias")
    skip = g.inp(f(batch, 3, h // 2, w // 2), "skip") if with_skip else None
    k4 = g.inp(torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)), "k4")
    wp = _packed(m, mode, g)
    y, img = g.out((batch, cout, h, w), "y"), g.out((batch, 3, h, w), "rgb")
    frames = g.out((batch, h, w, 3), "frames", dtype=torch.uint8) if u8 else None
    assert cin == cout  # (rgb_s shares the stride of s)
    rc = lib.maua_styledconv_torgb_f32(x.data_ptr(), wp.data_ptr(), s.data_ptr(), cin, d.data_ptr(), y.data_ptr(), batch, cin, cout, h, w, mode,
                                       float(m.scale), nz.data_ptr(), h * w, nw.data_ptr(), bias.data_ptr(), rgb_w.data_ptr(), rgb_s.data_ptr(),
                                       0.1, rgb_b.data_ptr(), _lib.ptr(skip), k4.data_ptr() if with_skip else None, img.data_ptr(), 1,
                                       _lib.ptr(frames), None, 0, None, _lib.stream_ptr(gpu))
    if mode != 5:
        assert lib.maua_modconv_ws_floats(batch, cin, cout, h, w, mode) == 0
    assert rc == 0, rc
    g.check(written=("y", "rgb"))
    if u8:
        want = ((img.clamp(-1, 1) + 1) * 127.5).to(torch.uint8).permute(0, 2, 3, 1)
        assert int((frames.int() - want.int()).abs().max()) <= 1  # (the kernel converts the same registers; a float tie may fall either way)


def test_partial_rgb_planes_stay_inside_their_buffers(gpu):
    """maua_styledconv_torgb_partial_f32 (128..512-channel layers): [B, 3 m_tiles, H, W] partial sums."""
    lib = _lib.load()
    batch, cin, cout, h, w = 2, 128, 256, 16, 32
    m, r = _layer(cin, cout, False, 99, gpu)
    g = Guard(gpu)
    f = lambda *shape: torch.from_numpy(r.standard_normal(shape).astype(np.float32))  # noqa: E731
    x, s, d = g.inp(f(batch, cin, h, w), "x"), g.inp(1 + 0.3 * f(batch, 256), "s"), g.inp(0.5 + torch.rand(batch, cout), "d")
    nz, nw, bias = g.inp(f(1, 1, h, w), "noise"), g.inp(torch.tensor([0.2]), "noise_w"), g.inp(0.3 * f(cout), "bias")
    rgb_w, rgb_s, post = g.inp(f(3, cout), "rgb_w"), g.inp(1 + 0.3 * f(batch, 256), "rgb_s"), g.inp(1 + 0.3 * f(batch, 256), "post_s")
    wp = _packed(m, 5, g)
    mt = lib.maua_modconv_w2d_mtiles(cin, cout, h, w)
    assert mt == 4
    y, part = g.out((batch, cout, h, w), "y"), g.out((batch, 3 * mt, h, w), "rgb_partial")
    rc = lib.maua_styledconv_torgb_partial_f32(x.data_ptr(), wp.data_ptr(), s.data_ptr(), 256, d.data_ptr(), y.data_ptr(), batch, cin, cout, h, w, 5,
                                               float(m.scale), nz.data_ptr(), 0, nw.data_ptr(), bias.data_ptr(), rgb_w.data_ptr(), rgb_s.data_ptr(),
                                               0.1, part.data_ptr(), None, 0, post.data_ptr(), _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y", "rgb_partial"))


LOWRES_UP = [  # (up, cin, cout, h, w, batch, noise_batch)
    (1, 512, 512, 4, 4, 8, 8),     # polyphase kernel, K split 32-fold: slabs, then reduce + blur + tail
    (1, 24, 40, 5, 7, 3, 1),       # generic loads, ragged, K not split (one slab), shared noise
    (1, 512, 512, 8, 8, 2, 2),     # two flat runs per image
    (6, 8, 32, 16, 16, 1, 1),      # F(2,2)^2 on 16 x 16-position tiles, minimum: one K step, one slab; exported column + edge lines
    (6, 72, 96, 16, 16, 3, 0),     # nine K steps, three output-channel tiles, no noise
    (6, 512, 512, 16, 16, 8, 8),   # the generator's 16^2 -> 32^2 layer at the bench batch: K split four-fold
    (6, 64, 32, 16, 16, 2, 2),     # one output-channel tile, two images, K split in two (the sanitizer driver's case)
    (6, 64, 64, 32, 16, 1, 1),     # two tiles per image along y? (H = 32: outside 2H * 2W <= 1024 -> rejected; kept as the boundary of the entry)
]


@pytest.mark.parametrize("up,cin,cout,h,w,batch,noise_batch", LOWRES_UP)
def test_lowres_upsampling_entry_stays_inside_its_buffers(gpu, up, cin, cout, h, w, batch, noise_batch):
    """maua_upconv_blur_lowres_f32 (transposed convolution -> split-K slabs [+ exported column], slab sum + demodulation + blur + noise + bias +
    act + post scale in one launch): every operand, the slab workspace at exactly maua_lowres_ws_floats and the output guarded."""
    from oracle import ops_oracle

    lib = _lib.load()
    m, r = _layer(cin, cout, True, cin + cout + h + w + up, gpu)
    if 4 * h * w > 1024:
        assert lib.maua_lowres_ok(cin, cout, h, w, up) == 0
        return
    assert lib.maua_lowres_ok(cin, cout, h, w, up) == 1
    g = Guard(gpu)
    stride = max(cin, cout)
    x_ = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s_row = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, stride))).astype(np.float32))
    d_ = torch.from_numpy((0.5 + r.random((batch, cout))).astype(np.float32))
    post_row = torch.from_numpy((1 + 0.3 * r.standard_normal((batch, stride))).astype(np.float32))
    bias_ = torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32))
    nz_ = torch.from_numpy(r.standard_normal((max(noise_batch, 1), 1, 2 * h, 2 * w)).astype(np.float32))
    x, s, d, post, bias = g.inp(x_, "x"), g.inp(s_row, "s"), g.inp(d_, "d"), g.inp(post_row, "post_s"), g.inp(bias_, "bias")
    k4, nw = g.inp(m.blur.kernel, "k4"), g.inp(torch.tensor([0.37]), "noise_w")
    nz = g.inp(nz_, "noise") if noise_batch else None
    wp = _packed(m, 6 if up == 6 else 0, g)
    y = g.out((batch, cout, 2 * h, 2 * w), "y")
    n_ws = lib.maua_lowres_ws_floats(batch, cin, cout, h, w, up)
    assert n_ws >= batch * cout * (2 * h + 1) * (2 * w + 1)
    ws = g.out((n_ws,), "ws")
    rc = lib.maua_upconv_blur_lowres_f32(x.data_ptr(), wp.data_ptr(), s.data_ptr(), stride, d.data_ptr(), y.data_ptr(), ws.data_ptr(), k4.data_ptr(),
                                         _lib.ptr(nz), 0 if noise_batch <= 1 else 4 * h * w, nw.data_ptr(), bias.data_ptr(), None, 0, batch, cin,
                                         cout, h, w, up, float(m.scale), post.data_ptr(), _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",))
    raw = _direct_conv(x_, s_row[:, :cin], d_, m.weight.cpu(), True)
    blurred = ops_oracle.upfirdn2d(raw, m.blur.kernel.cpu(), up=1, down=1, pad=(1, 1))
    t = blurred + (0.37 * nz_ if noise_batch else 0.0) + bias_[None, :, None, None]
    want = torch.where(t > 0, t, 0.2 * t) * 2 ** 0.5 * post_row[:, :cout, None, None]
    assert float((y.cpu() - want).abs().max()) <= 3e-4 * float(want.abs().max())
    # rejected: the pre-scaled form (no such instances here), a shape outside the entry's range
    assert lib.maua_upconv_blur_lowres_f32(x.data_ptr(), wp.data_ptr(), None, stride, d.data_ptr(), y.data_ptr(), ws.data_ptr(), k4.data_ptr(), None, 0,
                                           None, None, None, 0, batch, cin, cout, h, w, up, float(m.scale), None, None) == -22
    assert lib.maua_lowres_ok(cin, cout, 32, 32, up) == 0 and lib.maua_lowres_ok(cin, cout, 16, 32, 6) == 0


LOWRES_PLAIN = [(0, 512, 512, 4, 4, 8, True), (0, 40, 96, 4, 8, 3, True), (0, 512, 512, 16, 16, 2, False), (0, 64, 32, 8, 8, 1, True),
                (2, 512, 512, 8, 8, 8, True), (2, 512, 512, 16, 16, 2, True), (3, 128, 128, 16, 16, 1, False)]  # (mode, cin, cout, h, w, batch, noise)


@pytest.mark.parametrize("mode,cin,cout,h,w,batch,with_noise", LOWRES_PLAIN)
def test_lowres_plain_entry_and_plane_sum_stay_inside_their_buffers(gpu, mode, cin, cout, h, w, batch, with_noise):
    """maua_styledconv_rgbpart_lowres_f32 (direct convolution -> slabs; slab sum + tail + per-group partial ToRGB sums) and maua_torgb_f32's
    plane-sum form over its planes (+ bias + up-sampled skip)."""
    lib = _lib.load()
    m, r = _layer(cin, cout, False, cin + cout + h + w, gpu)
    assert lib.maua_lowres_ok(cin, cout, h, w, mode) == 1
    g = Guard(gpu)
    f = lambda *shape: torch.from_numpy(r.standard_normal(shape).astype(np.float32))  # noqa: E731
    stride = max(cin, cout)
    x_, s_, d_ = f(batch, cin, h, w), 1 + 0.3 * f(batch, stride), 0.5 + torch.rand(batch, cout)
    bias_, nz_ = 0.3 * f(cout), f(batch, 1, h, w)
    x, s, d = g.inp(x_, "x"), g.inp(s_, "s"), g.inp(d_, "d")
    nz, nw, bias = (g.inp(nz_, "noise") if with_noise else None), g.inp(torch.tensor([0.2]), "noise_w"), g.inp(bias_, "bias")
    rgb_w_, rgb_s_, rgb_b_ = f(3, cout), 1 + 0.3 * f(batch, stride), 0.3 * f(3)
    rgb_w, rgb_s, rgb_b = g.inp(rgb_w_, "rgb_w"), g.inp(rgb_s_, "rgb_s"), g.inp(rgb_b_, "rgb_bias")
    skip_ = f(batch, 3, h // 2, w // 2)
    skip, k4 = g.inp(skip_, "skip"), g.inp(torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)), "k4")
    wp = _packed(m, mode, g)
    groups = cout // 32
    y, part, img = g.out((batch, cout, h, w), "y"), g.out((batch, 3 * groups, h, w), "rgb_partial"), g.out((batch, 3, h, w), "rgb")
    ws = g.out((lib.maua_lowres_ws_floats(batch, cin, cout, h, w, mode),), "ws")
    rc = lib.maua_styledconv_rgbpart_lowres_f32(x.data_ptr(), wp.data_ptr(), s.data_ptr(), stride, d.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                                _lib.ptr(nz), h * w, nw.data_ptr(), bias.data_ptr(), rgb_w.data_ptr(), rgb_s.data_ptr(), 0.1,
                                                part.data_ptr(), None, 0, batch, cin, cout, h, w, mode, float(m.scale), _lib.stream_ptr(gpu))
    assert rc == 0, rc
    rc = lib.maua_torgb_f32(part.data_ptr(), None, None, 0, rgb_b.data_ptr(), skip.data_ptr(), k4.data_ptr(), img.data_ptr(), batch, 3 * groups, h, w,
                            1.0, _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y", "rgb_partial", "rgb"))
    t = _direct_conv(x_, s_[:, :cin], d_, m.weight.cpu(), False) + (0.2 * nz_ if with_noise else 0.0) + bias_[None, :, None, None]
    feat = torch.where(t > 0, t, 0.2 * t) * 2 ** 0.5
    assert float((y.cpu() - feat).abs().max()) <= 3e-4 * float(feat.abs().max())
    from oracle import ops_oracle
    from maua_stylegan2_amd.models.stylegan2 import Upsample

    up = Upsample([1, 3, 3, 1])
    want = torch.einsum("co,bo,bohw->bchw", 0.1 * rgb_w_, rgb_s_[:, :cout], feat) + rgb_b_[None, :, None, None] \
        + ops_oracle.upfirdn2d(skip_, up.kernel, up=2, down=1, pad=up.pad)
    assert float((img.cpu() - want).abs().max()) <= 3e-4 * max(1.0, float(want.abs().max()))


@pytest.mark.parametrize("batch,with_rgb", [(1, True), (8, True), (11, False)])
def test_const_conv_stays_inside_its_buffers(gpu, batch, with_rgb):
    """maua_pack_const_conv_f32 + maua_const_styledconv_f32 (conv1 on the constant input as y = T s): T at exactly Cout * 16 * Cin floats."""
    lib = _lib.load()
    cin, cout = 64, 96
    m, r = _layer(cin, cout, False, 1234 + batch, gpu)
    g = Guard(gpu)
    f = lambda *shape: torch.from_numpy(r.standard_normal(shape).astype(np.float32))  # noqa: E731
    c_, s_, d_ = f(cin, 4, 4), 1 + 0.3 * f(batch, 128), 0.5 + torch.rand(batch, cout)
    bias_, nz_, rgb_w_, rgb_s_ = 0.3 * f(cout), f(batch, 1, 4, 4), f(3, cout), 1 + 0.3 * f(batch, 128)
    w_in, c_in = g.inp(m.weight.reshape(cout, cin, 3, 3), "w"), g.inp(c_, "const")
    T = g.out((cout * 16 * cin,), "T")
    assert lib.maua_pack_const_conv_f32(w_in.data_ptr(), c_in.data_ptr(), T.data_ptr(), cout, cin, 4, 4, _lib.stream_ptr(gpu)) == 0
    g.check(written=("T",))
    s, d, bias, nz, nw = g.inp(s_, "s"), g.inp(d_, "d"), g.inp(bias_, "bias"), g.inp(nz_, "noise"), g.inp(torch.tensor([0.2]), "noise_w")
    rgb_w, rgb_s = g.inp(rgb_w_, "rgb_w"), g.inp(rgb_s_, "rgb_s")
    y = g.out((batch, cout, 4, 4), "y")
    part = g.out((batch, 3 * (cout // 32), 4, 4), "rgb_partial") if with_rgb else None
    rc = lib.maua_const_styledconv_f32(T.data_ptr(), s.data_ptr(), 128, d.data_ptr(), y.data_ptr(), nz.data_ptr(), 16, nw.data_ptr(), bias.data_ptr(),
                                       rgb_w.data_ptr() if with_rgb else None, rgb_s.data_ptr() if with_rgb else None, 0.1, _lib.ptr(part), None, 0,
                                       batch, cin, cout, 4, 4, float(m.scale), _lib.stream_ptr(gpu))
    assert rc == 0, rc
    g.check(written=("y",) + (("rgb_partial",) if with_rgb else ()))
    x_ = c_[None].expand(batch, -1, -1, -1)
    t = _direct_conv(x_, s_[:, :cin], d_, m.weight.cpu(), False) + 0.2 * nz_ + bias_[None, :, None, None]
    feat = torch.where(t > 0, t, 0.2 * t) * 2 ** 0.5
    assert float((y.cpu() - feat).abs().max()) <= 3e-4 * float(feat.abs().max())
    if with_rgb:
        want = torch.einsum("co,bo,bohw->bchw", 0.1 * rgb_w_, rgb_s_[:, :cout], feat)
        got = part.cpu().view(batch, cout // 32, 3, 4, 4).sum(1)
        assert float((got - want).abs().max()) <= 3e-4 * max(1.0, float(want.abs().max()))
    assert lib.maua_const_conv_ok(cin, cout, 4, 8) == 0 and lib.maua_const_conv_ok(60, cout, 4, 4) == 0
