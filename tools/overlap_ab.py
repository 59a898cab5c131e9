"""Experiment: one hipGraph of 8 frames vs two concurrently replayed hipGraphs of 4 frames on two streams
(the low-occupancy small layers and block-tail gaps of one overlap with the other's big layers)."""
import copy
import sys

import torch

sys.path.insert(0, '/root/repo')
from maua_stylegan2_amd import seeding, _lib  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import Generator, ModulatedConv2d  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
g1 = Generator(1024, 512, 8, channel_multiplier=2, constant_input=True)
g1.load_state_dict(seeding.seeded_state_dict(1024, seed=0))
g1 = g1.to(dev).eval()
sizes = seeding.noise_sizes(1024)
shapes = [(r, r) if r <= 256 else None for r in sizes]


def clone_sharing_weights(g):
    g2 = copy.deepcopy(g)
    for (n1, p1), (n2, p2) in zip(g.named_parameters(), g2.named_parameters()):
        p2.data = p1.data
    for (n1, b1), (n2, b2) in zip(g.named_buffers(), g2.named_buffers()):
        b2.data = b1.data
    m1 = dict(g.named_modules())
    for n, m in g2.named_modules():
        if isinstance(m, ModulatedConv2d):
            m._packed, m._packed_wino = m1[n]._packed, m1[n]._packed_wino
    g2._bufs = {}
    return g2


def time_graphs(pairs, iters=10):
    for gr, st in pairs:
        with torch.cuda.stream(st):
            for _ in range(3):
                gr.replay()
    torch.cuda.synchronize()
    e0, e1 = _lib.HipEvent(), _lib.HipEvent()
    main = pairs[0][1]
    e0.record(main.cuda_stream)
    for gr, st in pairs[1:]:
        st.wait_stream(main)
    for _ in range(iters):
        for gr, st in pairs:
            with torch.cuda.stream(st):
                gr.replay()
    for gr, st in pairs[1:]:
        main.wait_stream(st)
    e1.record(main.cuda_stream)
    return e0.elapsed_ms(e1) / iters



def setup(n_graphs, batch):
    gens = [g1] + [clone_sharing_weights(g1) for _ in range(n_graphs - 1)]
    pairs = []
    for g in gens:
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            gr, _ = g.capture_graph(batch, shapes)
        st.synchronize()
        pairs.append((gr, st))
    return pairs, gens


with torch.cuda.stream(torch.cuda.Stream(dev)):
    g1.capture_graph(8, shapes)  # packs the weights once so that clones share them
torch.cuda.synchronize()
for n_graphs, batch in [(1, 8), (2, 4), (2, 8), (4, 4), (3, 8), (2, 16), (1, 8)]:
    pairs, keep = setup(n_graphs, batch)
    ms = time_graphs(pairs)
    print(f"{n_graphs} graphs x {batch} frames: {ms:.3f} ms -> {n_graphs * batch / ms * 1e3:.1f} frames/s")
    del pairs, keep
    torch.cuda.empty_cache()
