#!/usr/bin/env python3
"""Fused transposed-convolution + blur: measurement harness for the experiment build.

    tools/build_exp.sh fuse && python tools/fuse_probe.py --lib tools/bin/libmaua_fuse.so

For the up-sampling layers of the 1024^2 generator it runs, on the same inputs,
  (a) the product pair: maua_modconv3x3_f32 (mode 6: raw (2H+1) x (2W+1) map + edge lines) -> maua_blur_noise_act_f32, and
  (b) the experiment kernel maua_exp_upconv_blur_fused_f32 (blur + noise + bias + leaky ReLU in the up-conv's epilogue, tile halos = 0),
checks (b) against (a) on every pixel whose 4 x 4 blur footprint lies inside one workgroup tile (the seam rows / columns are wrong by
construction in the experiment), and times (a)'s two launches, (b), and (b) with 12 / 20 % redundant tiles — the work an exact
overlapped tiling (x: 60 of 64 columns per tile kept, y: a warm-up tile per vertical segment) adds.  Prints one JSON object."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import _lib, seeding  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", required=True)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--exact", action="store_true", help="the product's exact fused kernel (maua_upconv_blur_f32) against the two-launch pair: whole maps")
    ap.add_argument("--sweep", default="", help="with --exact: comma-separated segment lengths (MAUA_FUSE_SEG) timed in alternation in this process")
    ap.add_argument("--diag", action="store_true", help="error maps of the experiment for single-tap (shift) kernels: which rows / columns are wrong")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    fused = None
    if not args.exact:  # (the halo-free experiment entry only exists in experiments builds)
        raw = ctypes.CDLL(_lib.LIB_PATH)
        fused = raw.maua_exp_upconv_blur_fused_f32
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        fused.argtypes = [vp, vp, vp, ctypes.c_int, vp, vp, vp, vp, i64, vp, vp] + [ctypes.c_int] * 5 + [ctypes.c_float, ctypes.c_int, vp]
        fused.restype = ctypes.c_int
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d

    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    sp = stream.cuda_stream
    B = args.batch
    out = {}
    k4 = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)).to(dev)
    if args.exact:
        with torch.cuda.stream(stream):
            for name, cin, cout, h, Bx in [("small 64->32 @32 B=3", 64, 32, 32, 3), ("small 128->64 @64x64 B=2", 128, 64, 64, 2),
                                           ("convs.14 64->32 @512", 64, 32, 512, B), ("convs.12 128->64 @256", 128, 64, 256, B),
                                           ("convs.10 256->128 @128", 256, 128, 128, B)]:
                m = ModulatedConv2d(cin, cout, 3, 512, upsample=True).to(dev)
                x = torch.randn(Bx, cin, h, h, device=dev)
                s = torch.randn(Bx, cin, device=dev)
                d = torch.rand(Bx, cout, device=dev) + 0.5
                raw_map = torch.empty(Bx, cout, 2 * h + 1, 2 * h + 1, device=dev)
                ref = torch.empty(Bx, cout, 2 * h, 2 * h, device=dev)
                got = torch.full((Bx, cout, 2 * h, 2 * h), float("nan"), device=dev)
                nz = torch.randn(Bx, 1, 2 * h, 2 * h, device=dev)
                nw = torch.full((1,), 0.3, device=dev)
                bias = torch.randn(cout, device=dev)
                ws = torch.empty(max(lib.maua_modconv_ws_floats(Bx, cin, cout, h, h, 6), 1), device=dev)
                nseam = lib.maua_upconv_blur_ws_floats(Bx, cin, cout, h, h)
                sweep = [int(v) for v in args.sweep.split(",") if v] if h >= 128 else []
                for sg in sweep:  # (the seam rows' buffer grows with the number of segments)
                    os.environ["MAUA_FUSE_SEG"] = str(sg)
                    nseam = max(nseam, lib.maua_upconv_blur_ws_floats(Bx, cin, cout, h, h))
                os.environ.pop("MAUA_FUSE_SEG", None)
                seam = torch.full((max(nseam, 1),), float("nan"), device=dev)
                wq = m.packed_wino(6)
                assert lib.maua_upconv_blur_ok(cin, cout, h, h)

                def pair():
                    m.run(x, s, 0, d, raw_map, ws)
                    _lib.check(lib.maua_blur_noise_act_f32(raw_map.data_ptr(), k4.data_ptr(), ref.data_ptr(), Bx, cout, 2 * h + 1, 2 * h + 1, 4, 4,
                                                           1, 1, None, nz.data_ptr(), 4 * h * h, nw.data_ptr(), bias.data_ptr(), None, 0, None, 0, sp), "tail")

                def exact():
                    _lib.check(lib.maua_upconv_blur_f32(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, d.data_ptr(), got.data_ptr(), seam.data_ptr(),
                                                        k4.data_ptr(), nz.data_ptr(), 4 * h * h, nw.data_ptr(), bias.data_ptr(), None, 0, Bx, cin, cout, h,
                                                        h, float(m.scale), None, sp), "maua_upconv_blur_f32")

                def exact_plain():  # (no noise map, no bias: what the tail's global loads cost)
                    _lib.check(lib.maua_upconv_blur_f32(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, d.data_ptr(), got.data_ptr(), seam.data_ptr(),
                                                        k4.data_ptr(), None, 0, nw.data_ptr(), None, None, 0, Bx, cin, cout, h,
                                                        h, float(m.scale), None, sp), "maua_upconv_blur_f32")

                pair(), exact()
                stream.synchronize()
                diff = torch.nan_to_num((got - ref).abs(), nan=99.0)
                rec = {"max_abs_err": float(diff.max()), "interior_max_abs_err": float(diff[:, :, :, 64:-128].max()) if h >= 128 else None,
                       "unwritten": int(torch.isnan(got).sum()), "ref_abs_mean": float(ref.abs().mean()),
                       "by_row_mod16": [round(float(v), 6) for v in diff.amax(dim=(0, 1, 3)).reshape(-1, 16).amax(0)],
                       "by_col_mod56_first8_last8": [round(float(v), 6) for v in torch.cat([diff.amax(dim=(0, 1, 2))[:8], diff.amax(dim=(0, 1, 2))[-8:]])],
                       "worst_rows": [int(v) for v in torch.topk(diff.amax(dim=(0, 1, 3)), 6).indices], "worst_cols": [int(v) for v in torch.topk(diff.amax(dim=(0, 1, 2)), 6).indices]}

                def timed(fn):
                    e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                    fn()
                    e0.record(sp)
                    for _ in range(args.iters):
                        fn()
                    e1.record(sp)
                    return e0.elapsed_ms(e1) / args.iters

                tp, te, tn = [], [], []
                for _ in range(args.rounds):
                    tp.append(timed(pair)), te.append(timed(exact)), tn.append(timed(exact_plain))
                rec["pair_ms"], rec["exact_ms"], rec["exact_no_noise_no_bias_ms"] = float(np.median(tp)), float(np.median(te)), float(np.median(tn))
                if sweep:
                    rows = {sg: [] for sg in sweep}
                    for _ in range(args.rounds):
                        for sg in sweep:
                            os.environ["MAUA_FUSE_SEG"] = str(sg)
                            rows[sg].append(timed(exact))
                    os.environ.pop("MAUA_FUSE_SEG", None)
                    rec["exact_ms_by_seg_tiles"] = {sg: round(float(np.median(v)), 4) for sg, v in rows.items()}
                out[name] = rec
        print(json.dumps(out))
        return
    if args.diag:
        with torch.cuda.stream(stream):
            cin, cout, h, Bd = 128, 64, 256, 2
            m = ModulatedConv2d(cin, cout, 3, 512, upsample=True).to(dev)
            x = torch.randn(Bd, cin, h, h, device=dev)
            s = torch.randn(Bd, cin, device=dev)
            d = torch.rand(Bd, cout, device=dev) + 0.5
            raw_map = torch.empty(Bd, cout, 2 * h + 1, 2 * h + 1, device=dev)
            ref = torch.empty(Bd, cout, 2 * h, 2 * h, device=dev)
            nz = torch.randn(Bd, 1, 2 * h, 2 * h, device=dev)
            nw = torch.full((1,), 0.3, device=dev)
            bias = torch.randn(cout, device=dev)
            ws = torch.empty(max(lib.maua_modconv_ws_floats(Bd, cin, cout, h, h, 6), 1), device=dev)
            wq = m.packed_wino(6)
            m.run(x, s, 0, d, raw_map, ws)
            cases = [("delta(1,1)", (1, 1), False), ("delta(1,0)", (1, 0), False), ("delta(1,2)", (1, 2), False), ("delta(1,3)", (1, 3), False), ("delta(0,1)", (0, 1), False),
                     ("delta(3,1)", (3, 1), False), ("full", None, False), ("full+noise+bias", None, True)]
            for cname, tap, tail in cases:
                if tap is None:
                    k = k4.clone()
                else:
                    k = torch.zeros(4, 4, device=dev)
                    k[3 - tap[0], 3 - tap[1]] = 1.0  # (the kernels flip the tap matrix: this is K[tap] = 1 in out = sum K[i][j] raw[Y-1+i][X-1+j])
                got = torch.full((Bd, cout, 2 * h, 2 * h), float("nan"), device=dev)
                _lib.check(lib.maua_blur_noise_act_f32(raw_map.data_ptr(), k.data_ptr(), ref.data_ptr(), Bd, cout, 2 * h + 1, 2 * h + 1, 4, 4, 1, 1,
                                                       None, nz.data_ptr() if tail else None, 4 * h * h, nw.data_ptr(), bias.data_ptr() if tail else None,
                                                       None, 0, None, 0, sp), "tail")
                _lib.check(fused(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, d.data_ptr(), got.data_ptr(), k.data_ptr(),
                                 nz.data_ptr() if tail else None, 4 * h * h, nw.data_ptr(), bias.data_ptr() if tail else None, Bd, cin, cout, h, h,
                                 float(m.scale), 0, sp), "fused")
                stream.synchronize()
                diff = torch.nan_to_num((got - ref).abs(), nan=99.0)[:, :, 16:-16]  # (the first / last tile rows: image borders, rows never written)
                per_row = diff.amax(dim=(0, 1, 3)).reshape(-1, 16).amax(0)
                inner_rows = diff.reshape(Bd, cout, -1, 16, 2 * h)[:, :, :, 1:14].reshape(Bd, cout, -1, 2 * h)
                per_col = inner_rows.amax(dim=(0, 1, 2)).reshape(-1, 64).amax(0)
                per_ch = inner_rows.amax(dim=(0, 2, 3))
                out[cname] = {"by_row_mod16": [round(float(v), 3) for v in per_row], "by_col_mod64": [round(float(v), 3) for v in per_col],
                              "by_channel_first8": [round(float(v), 3) for v in per_ch[:8]],
                              "got_row37_cols0_11": [round(float(v), 3) for v in got[0, 0, 37, :12]], "ref_row37_cols0_11": [round(float(v), 3) for v in ref[0, 0, 37, :12]]}
        print(json.dumps(out))
        return
    with torch.cuda.stream(stream):
        for name, cin, cout, h in [("convs.14 64->32 @512", 64, 32, 512), ("convs.12 128->64 @256", 128, 64, 256),
                                   ("convs.10 256->128 @128", 256, 128, 128), ("convs.8 512->256 @64", 512, 256, 64)]:
            m = ModulatedConv2d(cin, cout, 3, 512, upsample=True).to(dev)
            assert m.conv_mode(h, h) == 6
            x = torch.randn(B, cin, h, h, device=dev)
            s = torch.randn(B, cin, device=dev)
            d = torch.rand(B, cout, device=dev) + 0.5
            raw_map = torch.empty(B, cout, 2 * h + 1, 2 * h + 1, device=dev)
            ref = torch.empty(B, cout, 2 * h, 2 * h, device=dev)
            got = torch.full((B, cout, 2 * h, 2 * h), float("nan"), device=dev)
            nz = torch.randn(B, 1, 2 * h, 2 * h, device=dev)
            nw = torch.full((1,), 0.3, device=dev)
            bias = torch.randn(cout, device=dev)
            nws = lib.maua_modconv_ws_floats(B, cin, cout, h, h, 6)
            ws = torch.empty(max(nws, 1), device=dev)
            wq = m.packed_wino(6)

            def pair_up():
                m.run(x, s, 0, d, raw_map, ws)

            def pair_tail():
                _lib.check(lib.maua_blur_noise_act_f32(raw_map.data_ptr(), k4.data_ptr(), ref.data_ptr(), B, cout, 2 * h + 1, 2 * h + 1, 4, 4,
                                                       1, 1, None, nz.data_ptr(), 4 * h * h, nw.data_ptr(), bias.data_ptr(), None, 0, None, 0, sp),
                           "maua_blur_noise_act_f32")

            def run_fused(extra):
                rc = fused(x.data_ptr(), wq.data_ptr(), s.data_ptr(), cin, d.data_ptr(), got.data_ptr(), k4.data_ptr(), nz.data_ptr(),
                           4 * h * h, nw.data_ptr(), bias.data_ptr(), B, cin, cout, h, h, float(m.scale), extra, sp)
                _lib.check(rc, "maua_exp_upconv_blur_fused_f32")

            pair_up(), pair_tail(), run_fused(0)
            stream.synchronize()
            # pixels the experiment computes exactly: rows 16 t + 1 .. 16 t + 13, columns 64 u + 1 .. 64 u + 61
            yy = torch.arange(2 * h, device=dev) % 16
            xx = torch.arange(2 * h, device=dev) % 64
            mask = ((yy >= 1) & (yy <= 13))[:, None] & ((xx >= 1) & (xx <= 61))[None, :]
            diff = (got - ref).abs()
            inner = diff[:, :, mask]
            written = ~torch.isnan(got)
            rec = {"interior_max_abs_err": float(inner.max()), "interior_share": float(mask.float().mean()),
                   "ref_abs_mean": float(ref.abs().mean()), "rows_never_written": int((~written[0, 0].any(1)).sum()),
                   "seam_max_abs_err": float(torch.nan_to_num(diff[:, :, ~mask], nan=0.0).max())}

            def timed(fn):
                e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                fn()
                e0.record(sp)
                for _ in range(args.iters):
                    fn()
                e1.record(sp)
                return e0.elapsed_ms(e1) / args.iters

            rows = {"pair_upconv_ms": [], "pair_tail_ms": [], "fused_ms": [], "fused_plus12_ms": [], "fused_plus20_ms": []}
            for _ in range(args.rounds):  # alternating rounds: the chip's clock drifts with load
                rows["pair_upconv_ms"].append(timed(pair_up))
                rows["pair_tail_ms"].append(timed(pair_tail))
                rows["fused_ms"].append(timed(lambda: run_fused(0)))
                rows["fused_plus12_ms"].append(timed(lambda: run_fused(12)))
                rows["fused_plus20_ms"].append(timed(lambda: run_fused(20)))
            for key, vals in rows.items():
                rec[key] = float(np.median(vals))
            rec["pair_ms"] = rec["pair_upconv_ms"] + rec["pair_tail_ms"]
            out[name] = rec
        stream.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
