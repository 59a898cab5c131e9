#!/usr/bin/env python3
"""Device time of maua_temporal_fir_f32 (gaussian_filter) on the default plugin's shapes: python tools/tfir_probe.py [--lib build.so]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    args = ap.parse_args()
    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    dev = torch.device("cuda:0")
    out = {}
    for name, shape in [("noise 256^2", (900, 256 * 256)), ("noise 64^2", (900, 64 * 64)), ("latents 18x512", (900, 18 * 512))]:
        x = torch.randn(shape, device=dev)
        y = torch.empty_like(x)
        for sigma in (2, 5, 20):
            radius = 4 * sigma
            taps = torch.exp(-0.5 / sigma ** 2 * torch.arange(-radius, radius + 1, dtype=torch.float32) ** 2)
            taps = (taps / taps.sum()).to(dev)
            call = lambda: _lib.check(lib.maua_temporal_fir_f32(x.data_ptr(), taps.data_ptr(), y.data_ptr(), shape[0], shape[1], radius, _lib.stream_ptr(dev)), "fir")  # noqa: E731
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                call()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            out[f"{name} sigma {sigma} ({2 * radius + 1} taps)"] = {"ms": round(ms, 4), "gflops": round(2 * shape[0] * shape[1] * (2 * radius + 1) / ms / 1e6, 1),
                                                                  "gbs_algorithmic": round(8 * shape[0] * shape[1] / ms / 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
