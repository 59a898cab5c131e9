"""Probe for the transposed-conv kernels of mode 6 on the generator's five big up-sampling layer shapes (batch 8): run under
`rocprofv3 --kernel-trace` to get main / edge kernel durations per shape (tools/rocpd_summary.py prints min / max per kernel)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import _lib  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d  # noqa: E402

torch.set_grad_enabled(False)
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
stream = torch.cuda.Stream(dev)
B = 8
with torch.cuda.stream(stream):
    for cin, cout, h in [(512, 512, 32), (512, 256, 64), (256, 128, 128), (128, 64, 256), (64, 32, 512)]:
        m = ModulatedConv2d(cin, cout, 3, 512, upsample=True).to(dev)
        x = torch.randn(B, cin, h, h, device=dev)
        s = torch.randn(B, cin, device=dev)
        d = torch.rand(B, cout, device=dev) + 0.5
        raw = torch.empty(B, cout, 2 * h + 1, 2 * h + 1, device=dev)
        ws = torch.empty(_lib.load().maua_modconv_ws_floats(B, cin, cout, h, h, m.conv_mode(h, h)), device=dev)
        e0, e1 = _lib.HipEvent(), _lib.HipEvent()
        m.run(x, s, 0, d, raw, ws)
        e0.record(stream.cuda_stream)
        for _ in range(10):
            m.run(x, s, 0, d, raw, ws)
        e1.record(stream.cuda_stream)
        print(f"{cin}->{cout} @{h}: {e0.elapsed_ms(e1) / 10 * 1e3:.1f} us per main+edge")
