"""Bisect helper: render an 11-frame 512^2 sequence (batch 2) repeatedly through render.synthesize with / without mode 6 and
report which frames differ between (a) full range, (b) frame_range shards, (c) repeated runs."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import render, seeding  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import Generator, ModulatedConv2d  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")


def frames_of(g, lat, noise, rng, use_graph=True):
    lo, hi = rng
    out = np.zeros((hi - lo, 512, 512, 3), np.uint8)
    for first, u8 in render.synthesize(g, lat, noise, 2, frame_range=rng, use_graph=use_graph):
        out[first - lo: first - lo + u8.shape[0]] = u8.cpu().numpy()
    return out


for up2d in (32, 10 ** 9):
    ModulatedConv2d.upwino2d_min_cout = up2d
    g = Generator(512, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(512, seed=1), strict=True)
    g = g.to(dev).eval()
    lat = seeding.seeded_latents(11, g.n_latent, seed=2)
    noise = [None] * g.num_layers
    full = frames_of(g, lat, noise, (0, 11))
    again = frames_of(g, lat, noise, (0, 11))
    eager = frames_of(g, lat, noise, (0, 11), use_graph=False)
    a, b = frames_of(g, lat, noise, (0, 6)), frames_of(g, lat, noise, (6, 11))
    sh = np.concatenate([a, b])
    print("up2d_min_cout", up2d)
    for name, other in (("again", again), ("eager", eager), ("shards", sh)):
        diff = [int(np.abs(full[i].astype(int) - other[i].astype(int)).max()) for i in range(11)]
        print("  full vs", name, diff)
