#!/usr/bin/env python3
"""Kernel micro-benchmarks for profiling runs (rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE).

    python tools/microbench.py [fir] [copy] [conv] [--batch 8] [--iters 5]

`copy` launches the fused_bias_act kernel on buffers of KNOWN size through its 16-byte/lane and its dword/lane paths:
the PMC byte counters of these two launches calibrate FETCH_SIZE / WRITE_SIZE for the access widths used by the other
kernels (MI355X_MICROARCH.md §HBM: FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import _lib, seeding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="*", default=["fir", "copy"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--conv-debug", type=int, default=0)
    ap.add_argument("--conv-cfg", type=int, default=0)
    ap.add_argument("--lib", default=None, help="experimental build of libmaua_hip.so to load instead (tools/bin/...)")
    ap.add_argument("--wino-min-cout", type=int, default=None, help="override ModulatedConv2d.winograd_min_cout")
    ap.add_argument("--no-up-wino", action="store_true", help="transposed layers use the plain polyphase kernel (mode 1)")
    ap.add_argument("--wino43-min-cout", type=int, default=None, help="override ModulatedConv2d.winograd43_min_cout")
    ap.add_argument("--wino2d-min-cout", type=int, default=None, help="override ModulatedConv2d.winograd2d_min_cout (huge = mode 3)")
    ap.add_argument("--split-bf16-min-cout", type=int, default=None, help="side measurement: plain layers from this many channels run mode 7 (split-bf16 products)")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    if args.conv_debug or args.conv_cfg:  # ablation switches: experiments builds only (tools/build_exp.sh + --lib)
        lib.maua_tuning_set(1, args.conv_debug)
        lib.maua_tuning_set(2, args.conv_cfg)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    sp = stream.cuda_stream
    out = {}
    with torch.cuda.stream(stream):
        if "copy" in args.what:
            n = 1 << 28  # 1 GiB read + 1 GiB write per launch
            x = torch.randn(n, device=dev)
            y = torch.empty_like(x)
            b = torch.zeros(16, device=dev)
            for name, shape_step in [("copy_vec16B", 4096), ("copy_dword", 4097)]:
                nn = (n // shape_step) * shape_step
                e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                lib.maua_fused_bias_act_f32(x.data_ptr(), b.data_ptr(), None, y.data_ptr(), nn, 16, shape_step, 3, 0, 0.2, 1.0, sp)
                e0.record(sp)
                for _ in range(args.iters):
                    lib.maua_fused_bias_act_f32(x.data_ptr(), b.data_ptr(), None, y.data_ptr(), nn, 16, shape_step, 3, 0, 0.2, 1.0, sp)
                e1.record(sp)
                ms = e0.elapsed_ms(e1) / args.iters
                out[name] = {"ms": ms, "bytes_read": 4 * nn, "bytes_written": 4 * nn, "gbs": 8 * nn / ms / 1e6}
        if "fir" in args.what:
            B, C, r = args.batch, 32, 1024
            x = torch.randn(B, C, r + 1, r + 1, device=dev)
            k = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)).to(dev)
            y = torch.empty(B * C, r, r, 1, device=dev)
            call = lambda: lib.maua_upfirdn2d_f32(x.data_ptr(), k.data_ptr(), y.data_ptr(), B * C, r + 1, r + 1, 1, 4, 4, 1, 1, 1, 1, 1, 1, 1, 1, sp)  # noqa: E731
            call()
            e0, e1 = _lib.HipEvent(), _lib.HipEvent()
            e0.record(sp)
            for _ in range(args.iters):
                call()
            e1.record(sp)
            ms = e0.elapsed_ms(e1) / args.iters
            byts = 4 * B * C * ((r + 1) ** 2 + r ** 2)
            out["fir_1024"] = {"ms": ms, "bytes_read": 4 * B * C * (r + 1) ** 2, "bytes_written": 4 * B * C * r * r,
                               "gbs": byts / ms / 1e6}
            # the Upsample module as a standalone op (up = 2, pad (2, 1), taps x up ** 2): [B, C, 512, 512] -> [B, C, 1024, 1024]
            xu = torch.randn(B, C, r // 2, r // 2, device=dev)
            ku = k  # (fir_kernel_2d's gain 4 is the Upsample module's up ** 2)
            call_up = lambda: lib.maua_upfirdn2d_f32(xu.data_ptr(), ku.data_ptr(), y.data_ptr(), B * C, r // 2, r // 2, 1, 4, 4, 2, 2, 1, 1, 2, 1, 2, 1, sp)  # noqa: E731
            call_up()
            e0.record(sp)
            for _ in range(args.iters):
                call_up()
            e1.record(sp)
            ms = e0.elapsed_ms(e1) / args.iters
            byts = 4 * B * C * ((r // 2) ** 2 + r ** 2)
            out["fir_up2_512_to_1024"] = {"ms": ms, "bytes_read": 4 * B * C * (r // 2) ** 2, "bytes_written": 4 * B * C * r * r, "gbs": byts / ms / 1e6}
            # the Downsample module (down = 2, pad (1, 1)): [B, C, 1024, 1024] -> [B, C, 512, 512]
            xd = torch.randn(B, C, r, r, device=dev)
            yd = torch.empty(B * C, r // 2, r // 2, 1, device=dev)
            call_dn = lambda: lib.maua_upfirdn2d_f32(xd.data_ptr(), k.data_ptr(), yd.data_ptr(), B * C, r, r, 1, 4, 4, 1, 1, 2, 2, 1, 1, 1, 1, sp)  # noqa: E731
            call_dn()
            e0.record(sp)
            for _ in range(args.iters):
                call_dn()
            e1.record(sp)
            ms = e0.elapsed_ms(e1) / args.iters
            byts = 4 * B * C * (r ** 2 + (r // 2) ** 2)
            out["fir_down2_1024_to_512"] = {"ms": ms, "bytes_read": 4 * B * C * r ** 2, "bytes_written": 4 * B * C * (r // 2) ** 2, "gbs": byts / ms / 1e6}
            nz = torch.randn(1, 1, r, r, device=dev)
            nw = torch.full((1,), 0.1, device=dev)
            bias = torch.randn(C, device=dev)
            y2 = torch.empty(B, C, r, r, device=dev)
            call2 = lambda: lib.maua_blur_noise_act_f32(x.data_ptr(), k.data_ptr(), y2.data_ptr(), B, C, r + 1, r + 1, 4, 4, 1, 1, None, nz.data_ptr(), 0, nw.data_ptr(), bias.data_ptr(), None, 0, None, 0, sp)  # noqa: E731
            call2()
            e0.record(sp)
            for _ in range(args.iters):
                call2()
            e1.record(sp)
            ms = e0.elapsed_ms(e1) / args.iters
            out["fir_tail_1024"] = {"ms": ms, "gbs": byts / ms / 1e6}
        if "conv" in args.what:
            from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

            B = args.batch
            if args.wino_min_cout is not None:
                ModulatedConv2d.winograd_min_cout = args.wino_min_cout
            if args.no_up_wino:
                ModulatedConv2d.upconv_winograd = False
            if args.wino43_min_cout is not None:
                ModulatedConv2d.winograd43_min_cout = args.wino43_min_cout
            if args.wino2d_min_cout is not None:
                ModulatedConv2d.winograd2d_min_cout = args.wino2d_min_cout
            if args.split_bf16_min_cout is not None:
                ModulatedConv2d.split_bf16_min_cout = args.split_bf16_min_cout
                ModulatedConv2d.split_bf16_up_min_cout = 32  # (the transposed kernel works on 32-channel m-tiles)
            for name, cin, cout, h, up in [("plain32@1024", 32, 32, 1024, 0), ("plain64@512", 64, 64, 512, 0),
                                           ("plain128@256", 128, 128, 256, 0), ("plain256@128", 256, 256, 128, 0),
                                           ("plain512@64", 512, 512, 64, 0), ("up64-32@512", 64, 32, 512, 1),
                                           ("up128-64@256", 128, 64, 256, 1), ("up256-128@128", 256, 128, 128, 1),
                                           ("up512-256@64", 512, 256, 64, 1), ("up512-512@32", 512, 512, 32, 1)]:
                m = ModulatedConv2d(cin, cout, 3, 512, upsample=bool(up)).to(dev)
                x = torch.randn(B, cin, h, h, device=dev)
                s = torch.randn(B, cin, device=dev)
                d = torch.rand(B, cout, device=dev)
                oh = 2 * h + 1 if up else h
                yv = torch.empty(B, cout, oh, oh, device=dev)
                nz = torch.randn(B, 1, oh, oh, device=dev)
                nw = torch.full((1,), 0.1, device=dev)
                bias = torch.randn(cout, device=dev)
                nws = lib.maua_modconv_ws_floats(B, cin, cout, h, h, m.conv_mode(h, h))
                ws = torch.empty(max(nws, 1), device=dev)
                run = lambda: m.run(x, s, 0, d, yv, ws if nws else None, fuse_act=not up, noise=None if up else nz, noise_w=nw, bias=bias)  # noqa: E731
                run()
                e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                e0.record(sp)
                for _ in range(args.iters):
                    run()
                e1.record(sp)
                ms = e0.elapsed_ms(e1) / args.iters
                fl = 2.0 * cin * cout * 9 * h * h * B
                out[name] = {"ms": ms, "tflops": fl / ms / 1e9, "mode": m.conv_mode(h, h), "kernel": _lib.last_modconv_instance()}
        if "fused" in args.what:
            # the two fused layers of the 1024^2 generator: StyledConv + ToRGB (+ skip) in one launch, feature map stored
            # (convs.13) / not stored and frames written as uint8 (convs.15)
            from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d, StyledConv, ToRGB, _style_table

            B = args.batch
            if args.wino2d_min_cout is not None:
                ModulatedConv2d.winograd2d_min_cout = args.wino2d_min_cout
            for name, c, h, last in [("fused64@512", 64, 512, False), ("fused32@1024", 32, 1024, True)]:
                conv, rgb = StyledConv(c, c, 3, 512).to(dev), ToRGB(c, 512).to(dev)
                x = torch.randn(B, c, h, h, device=dev)
                lat = torch.randn(B, 2, 512, device=dev)
                styles = torch.empty(B, 2 * c, device=dev)
                demod = torch.empty(B * c, device=dev)
                table = _style_table([conv.conv.table_entry(0, 0, 0), rgb.conv.table_entry(1, c, B * c)], dev)
                lib.maua_style_affine_f32(lat.data_ptr(), B, 2, 512, None, None, table.data_ptr(), 2, c, styles.data_ptr(), 2 * c, None, sp)
                lib.maua_demod_f32(table.data_ptr(), 2, c, styles.data_ptr(), 2 * c, demod.data_ptr(), B, sp)
                nz = torch.randn(1, 1, h, h, device=dev)
                skip = torch.randn(B, 3, h // 2, h // 2, device=dev)
                feat = torch.empty(B, c, h, h, device=dev)
                img = torch.empty(B, 3, h, h, device=dev)
                u8 = torch.empty(B, h, h, 3, dtype=torch.uint8, device=dev)
                bufs = lambda nm, shape: feat  # noqa: E731

                def run():
                    fuse = dict(module=rgb, s_off=c, skip=skip, out=img, store=not last, u8=u8 if last else None)
                    conv.run(x, styles, 0, demod.view(B, c), nz, bufs, "f", rgb=fuse)
                    assert fuse.get("done")

                run()
                e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                e0.record(sp)
                for _ in range(args.iters):
                    run()
                e1.record(sp)
                ms = e0.elapsed_ms(e1) / args.iters
                out[name] = {"ms": ms, "tflops": 2.0 * c * c * 9 * h * h * B / ms / 1e9, "mode": conv.conv.conv_mode(h, h),
                             "kernel": _lib.last_modconv_instance()}
        stream.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
