"""Round 6 (profiles/r06_store_hazard.md): the two wave-complete 2-D Winograd shapes whose 16-byte row stores lost their last dword when the
compiler re-used the data registers right behind a buffer store with an SGPR soffset.  python tests/store_hazard_check.py [libmaua_hip.so [abi]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maua_stylegan2_amd import _lib
from maua_stylegan2_amd.models.stylegan2 import StyledConv
from oracle import stylegan2_oracle as so
torch.set_grad_enabled(False)
gpu = torch.device("cuda:0")
if len(sys.argv) > 1:  # (an alternative build of the library)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
if len(sys.argv) > 2:
    _lib.ABI_VERSION = int(sys.argv[2])
for (cin, cout, h, w, batch) in [(32, 32, 32, 64, 2), (16, 32, 16, 32, 3)]:
    r = np.random.default_rng(5 * cin + cout + h + w)
    m = StyledConv(cin, cout, 3, 512, upsample=False)
    sd = {
        "L.conv.weight": torch.from_numpy(r.standard_normal((1, cout, cin, 3, 3)).astype(np.float32)),
        "L.conv.modulation.weight": torch.from_numpy(r.standard_normal((cin, 512)).astype(np.float32)),
        "L.conv.modulation.bias": torch.from_numpy((1 + 0.1 * r.standard_normal(cin)).astype(np.float32)),
        "L.noise.weight": torch.tensor([0.31]),
        "L.activate.bias": torch.from_numpy((0.3 * r.standard_normal(cout)).astype(np.float32)),
    }
    m.load_state_dict({k[2:]: v for k, v in sd.items()}, strict=True)
    m = m.to(gpu)
    x = torch.from_numpy(r.standard_normal((batch, cin, h, w)).astype(np.float32))
    s = torch.from_numpy(r.standard_normal((batch, 512)).astype(np.float32))
    nz = torch.from_numpy(r.standard_normal((batch, 1, h, w)).astype(np.float32))
    want = so.styled_conv(sd, "L", x, s, nz, False).numpy()
    for rep in range(3):
        got = m(x.to(gpu), s.to(gpu), noise=nz.to(gpu)).cpu().numpy()
        bad = np.abs(got - want) > 5e-4 + 1e-4 * np.abs(want)
        print(cin, cout, h, w, "instance", _lib.last_modconv_instance(), "bad", bad.sum(), "of", bad.size)
        if bad.any() and rep == 0 and len(sys.argv) > 3:
            idx = np.argwhere(bad)
            print(" bad per batch", np.bincount(idx[:, 0], minlength=batch))
            print(" bad per channel", np.bincount(idx[:, 1], minlength=cout))
            print(" bad per row", np.bincount(idx[:, 2], minlength=h))
            print(" bad per col", np.bincount(idx[:, 3], minlength=w))
            i = idx[0]
            print(" first", i, got[tuple(i)], want[tuple(i)], "ratio", got[tuple(i)] / want[tuple(i)])
            ratio = got[bad] / want[bad]
            print(" ratio stats", np.percentile(ratio, [0, 25, 50, 75, 100]))
