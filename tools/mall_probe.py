"""Does a transposed layer's raw map survive in the 256 MiB Infinity Cache between the convolution that writes it and the blur
that reads it?  Times conv + blur tail of the five transposed layers of the 1024^2 generator at batch 8 in ONE pass over the batch
(raw map of the whole batch: 136 MB .. 1.07 GB) against groups of g frames that re-use ONE raw buffer of g frames.
Run on the GPU box: python tools/mall_probe.py [--iters 10]."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.models.stylegan2 import StyledConv

    lib = _lib.load()
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(dev)
    sp = stream.cuda_stream
    B = args.batch
    out = {}
    with torch.cuda.stream(stream):
        for name, cin, cout, h in [("up64-32@512", 64, 32, 512), ("up128-64@256", 128, 64, 256), ("up256-128@128", 256, 128, 128),
                                   ("up512-256@64", 512, 256, 64), ("up512-512@32", 512, 512, 32)]:
            m = StyledConv(cin, cout, 3, 512, upsample=True).to(dev)
            x = torch.randn(B, cin, h, h, device=dev)
            s = torch.randn(B, cin, device=dev)
            d = torch.rand(B, cout, device=dev)
            nz = torch.randn(1, 1, 2 * h, 2 * h, device=dev)
            raw = torch.empty(B, cout, 2 * h + 1, 2 * h + 1, device=dev)
            y = torch.empty(B, cout, 2 * h, 2 * h, device=dev)
            res = {}
            ref = None
            for g in (B, 4, 2, 1):
                def bufs(nm, shape, g=g, i=[0]):
                    if nm.endswith(".raw"):
                        return raw[:g]
                    if nm.endswith(".ws"):
                        return torch.empty(shape, device=dev)
                    return y[bufs.i0:bufs.i0 + g]

                def run(g=g, bufs=bufs):
                    for i0 in range(0, B, g):
                        bufs.i0 = i0
                        m.run(x[i0:i0 + g], s[i0:i0 + g], 0, d[i0:i0 + g], nz, bufs, "l")

                run()
                stream.synchronize()
                if ref is None:
                    ref = y.clone()
                else:
                    assert torch.equal(ref, y), (name, g)
                e0, e1 = _lib.HipEvent(), _lib.HipEvent()
                e0.record(sp)
                for _ in range(args.iters):
                    run()
                e1.record(sp)
                res["g%d" % g] = round(e0.elapsed_ms(e1) / args.iters, 4)
            res["raw_MB_per_frame"] = round(cout * (2 * h + 1) ** 2 * 4 / 1e6, 1)
            out[name] = res
            del x, raw, y
        stream.synchronize()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
