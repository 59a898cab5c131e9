"""Determinism soak for modconv_mfma_kernel: every kernel mode / tile config, 30 launches each on fresh output buffers
(poisoned with NaN first), outputs must be bit-identical to the first launch and free of NaN.  A missing barrier or an
unwritten output element shows up here long before it shows up in a parity test."""
import sys

import torch

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd.models.stylegan2 import StyledConv  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cases = [  # cin, cout, h, w, up, batch
    (512, 512, 4, 4, False, 8), (512, 512, 8, 8, True, 8), (512, 512, 16, 16, False, 8), (512, 512, 16, 16, True, 8),
    (512, 512, 32, 32, False, 8), (512, 512, 32, 32, True, 4), (512, 512, 64, 64, False, 4), (512, 256, 64, 64, True, 4),
    (256, 256, 128, 128, False, 4), (256, 128, 128, 128, True, 4), (128, 128, 256, 256, False, 2), (128, 64, 256, 256, True, 2),
    (64, 64, 512, 512, False, 2), (64, 32, 512, 512, True, 2), (32, 32, 1024, 1024, False, 2), (32, 32, 96, 68, False, 3),
    (64, 64, 40, 34, False, 3), (24, 40, 20, 38, False, 2), (72, 24, 33, 20, True, 3), (16, 16, 128, 128, False, 2),
    # mode 5 (2-D Winograd) incl. its 32-channel tile config, forced below the generator's 128-channel threshold
    ("w2d", 64, 64, 64, 96, 3), ("w2d", 32, 32, 64, 64, 3), ("w2d", 128, 192, 24, 32, 2), ("w2d", 32, 32, 40, 64, 2),
]
_empty = torch.empty


def poisoned_empty(*args, **kwargs):  # every buffer the layer allocates starts as NaN: an unwritten element cannot hide
    t = _empty(*args, **kwargs)
    return t.fill_(float("nan")) if t.is_floating_point() else t


bad = 0
for cin, cout, h, w, up, b in cases:
    if cin == "w2d":
        cin, cout, h, w, b, up = cout, h, w, up, b, False
        m = StyledConv(cin, cout, 3, 512).to(dev)
        m.conv.winograd2d_min_cout = 32
    else:
        m = StyledConv(cin, cout, 3, 512, upsample=up).to(dev)
    m.conv.weight.normal_(), m.noise.weight.fill_(0.3), m.activate.bias.normal_(0, 0.2)
    x = torch.randn(b, cin, h, w, device=dev)
    s = torch.randn(b, 512, device=dev)
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    nz = torch.randn(b, 1, oh, ow, device=dev)
    torch.empty = poisoned_empty
    ref = m(x, s, noise=nz).clone()
    ok = bool(torch.isfinite(ref).all())
    for _ in range(30):
        y = m(x, s, noise=nz)
        if not torch.equal(y, ref):
            ok = False
            break
    torch.empty = _empty
    mode = m.conv.conv_mode(h, w)
    print(f"cin {cin:4d} cout {cout:4d} {h}x{w} up={int(up)} batch {b} mode {mode}: {'ok' if ok else 'MISMATCH'}")
    bad += not ok
# round 6: the whole generator with the style fold (every producer stores its map pre-multiplied, the consumers run the PRE instances; the
# 16-byte stores with SGPR channel offsets carry their own wait states, profiles/r06_store_hazard.md): 20 forwards of 4 frames, bit-identical
from maua_stylegan2_amd import seeding  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import Generator  # noqa: E402

for size in (1024, 256):
    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(size, seed=0))
    g = g.to(dev).eval()
    lat = seeding.seeded_latents(4, g.n_latent, seed=1).to(dev)
    torch.empty = poisoned_empty
    ref, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
    ok = bool(torch.isfinite(ref).all()) and any(c.posted for c in g.convs)
    for _ in range(20):
        y, _ = g(styles=lat, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
        if not torch.equal(y, ref):
            ok = False
            break
    torch.empty = _empty
    print(f"generator {size}^2 batch 4, style fold on ({sum(c.posted for c in g.convs)} folded maps): {'ok' if ok else 'MISMATCH'}")
    bad += not ok
    del g
print("soak:", "all deterministic" if not bad else f"{bad} cases nondeterministic")
sys.exit(1 if bad else 0)
