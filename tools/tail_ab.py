#!/usr/bin/env python3
"""A/B of the fused blur+noise+bias+act tail (maua_blur_noise_act_f32) across maua_tuning_set(0, path) values: timing at the
generator's up-sampling shapes and equality with path 0 (incl. ragged shapes that exercise the edge tiles).

    python tools/tail_ab.py [path ...]      # default: 0 1 5 (auto, per-plane tiles, tiles + non-temporal)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from maua_stylegan2_amd import _lib, seeding


def main():
    paths = [int(a) for a in sys.argv[1:]] or [0, 1, 5]
    lib = _lib.load()
    dev = torch.device("cuda:0")
    sp = _lib.stream_ptr(dev)
    k = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)).to(dev)

    def run(path, x, nz, nstride, nw, bias, gain, y):
        b, c, ih, iw = x.shape
        lib.maua_tuning_set(0, path)
        rc = lib.maua_blur_noise_act_f32(x.data_ptr(), k.data_ptr(), y.data_ptr(), b, c, ih, iw, 4, 4, 1, 1, _lib.ptr(gain),
                                         _lib.ptr(nz), nstride, nw.data_ptr(), bias.data_ptr(), sp)
        lib.maua_tuning_set(0, 0)
        assert rc == 0, rc

    out = {}
    # equality on ragged shapes (edge tiles in x and y, channel count not a multiple of the strip, batch-shared / per-frame noise)
    for (b, c, r, per_frame) in [(2, 5, 37, True), (1, 7, 130, False), (3, 4, 300, True)]:
        x = torch.randn(b, c, r + 1, r + 1, device=dev)
        nz = torch.randn(b if per_frame else 1, 1, r, r, device=dev)
        nw = torch.full((1,), 0.3, device=dev)
        bias = torch.randn(c, device=dev)
        gain = torch.rand(b * c, device=dev) + 0.5
        ref = torch.empty(b, c, r, r, device=dev)
        run(0, x, nz, r * r if per_frame else 0, nw, bias, gain, ref)
        for p in paths[1:]:
            y = torch.full_like(ref, float("nan"))
            run(p, x, nz, r * r if per_frame else 0, nw, bias, gain, y)
            out[f"maxdiff path{p} [{b},{c},{r}]"] = float((y - ref).abs().max())
    for (c, r) in [(32, 1024), (64, 512), (128, 256), (256, 128)]:
        b = 8
        x = torch.randn(b, c, r + 1, r + 1, device=dev)
        nz = torch.randn(b, 1, r, r, device=dev)
        nw = torch.full((1,), 0.1, device=dev)
        bias = torch.randn(c, device=dev)
        y = torch.empty(b, c, r, r, device=dev)
        byts = 4 * b * c * ((r + 1) ** 2 + r ** 2)
        for p in paths:
            run(p, x, nz, r * r, nw, bias, None, y)
            e0, e1 = _lib.HipEvent(), _lib.HipEvent()
            e0.record(sp)
            for _ in range(20):
                run(p, x, nz, r * r, nw, bias, None, y)
            e1.record(sp)
            ms = e0.elapsed_ms(e1) / 20
            out[f"tail {c}@{r} path{p}"] = (round(ms, 4), round(byts / ms / 1e6))
    for k_, v_ in out.items():
        print(k_, v_)


if __name__ == "__main__":
    main()
