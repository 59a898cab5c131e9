"""Which plain-convolution kernel for the low-resolution layers?  Times ModulatedConv2d.run (convolution + split-K reduce + tail) of a
512 -> 512 layer at 8^2 / 16^2 / 32^2, batch 8, in mode 0 (direct), 2 (Winograd F(2,3) along x), 3 (F(4,3)), 5 (2-D Winograd where accepted).

    python tools/plain_mode_probe.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from maua_stylegan2_amd import _lib  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    torch.set_grad_enabled(False)
    lib = _lib.load()
    stream = torch.cuda.Stream(dev)
    b = 8
    for cin, cout, h in [(512, 512, 8), (512, 512, 16), (512, 512, 32)]:
        m = ModulatedConv2d(cin, cout, 3, 512).to(dev)
        x = torch.randn(b, cin, h, h, device=dev)
        s = torch.randn(b, cin, device=dev)
        d = torch.rand(b, cout, device=dev) + 0.5
        y = torch.empty(b, cout, h, h, device=dev)
        nz = torch.randn(b, 1, h, h, device=dev)
        nw, bias = torch.full((1,), 0.1, device=dev), torch.zeros(cout, device=dev)
        ref = None
        for mode in (0, 2, 3, 5):
            m.conv_mode = lambda hh, ww, mode=mode: mode
            n_ws = lib.maua_modconv_ws_floats(b, cin, cout, h, h, mode)
            ws = torch.empty(max(n_ws, 1), device=dev)
            with torch.cuda.stream(stream):
                try:
                    fn = lambda: m.run(x, s, 0, d, y, ws if n_ws else None, fuse_act=True, noise=nz, noise_w=nw, bias=bias)  # noqa: E731
                    t = bench.time_calls(fn, 20, stream.cuda_stream)
                    stream.synchronize()
                    if ref is None:
                        ref = y.clone()
                    err = float((y - ref).abs().max() / ref.abs().max())
                    print(f"{cin}->{cout} @{h}^2 mode {mode}: {t * 1e3:.1f} us  ({_lib.last_modconv_instance()}, ws {n_ws}, rel err vs mode 0 {err:.1e})", flush=True)
                except Exception as e:  # noqa: BLE001
                    print(f"{cin}->{cout} @{h}^2 mode {mode}: {e}", flush=True)


if __name__ == "__main__":
    main()
