"""Which transposed-convolution kernel for the low-resolution up-sampling layers?  Times ModulatedConv2d.run (convolution + split-K reduce,
raw output) of a 512 -> 512 layer at 4^2 / 8^2 / 16^2 / 32^2 inputs, batch 8, in mode 1 (polyphase) and mode 4 (F(2,2) on the even x-phase).

    python tools/up_mode_probe.py"""
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
    for cin, cout, h in [(512, 512, 4), (512, 512, 8), (512, 512, 16), (512, 512, 32), (512, 256, 64)]:
        m = ModulatedConv2d(cin, cout, 3, 512, upsample=True).to(dev)
        x = torch.randn(b, cin, h, h, device=dev)
        s = torch.randn(b, cin, device=dev)
        d = torch.rand(b, cout, device=dev) + 0.5
        raw = torch.empty(b, cout, 2 * h + 1, 2 * h + 1, device=dev)
        for mode in (1, 4):
            m.conv_mode = lambda hh, ww, mode=mode: mode
            n_ws = lib.maua_modconv_ws_floats(b, cin, cout, h, h, mode)
            ws = torch.empty(max(n_ws, 1), device=dev)
            with torch.cuda.stream(stream):
                try:
                    t = bench.time_calls(lambda: m.run(x, s, 0, d, raw, ws if n_ws else None), 20, stream.cuda_stream)
                    print(f"{cin}->{cout} @{h}^2 mode {mode}: {t * 1e3:.1f} us  ({_lib.last_modconv_instance()}, ws {n_ws})", flush=True)
                except Exception as e:  # noqa: BLE001
                    print(f"{cin}->{cout} @{h}^2 mode {mode}: {e}", flush=True)


if __name__ == "__main__":
    main()
