import sys, torch, time
sys.path.insert(0, '/root/repo')
from maua_stylegan2_amd import seeding, _lib
from maua_stylegan2_amd.models.stylegan2 import Generator
torch.set_grad_enabled(False)
dev = torch.device('cuda:0')
g = Generator(1024, 512, 8, channel_multiplier=2, constant_input=True)
g.load_state_dict(seeding.seeded_state_dict(1024, seed=0)); g = g.to(dev).eval()
B = 8
sizes = seeding.noise_sizes(1024)
shapes = [(r, r) if r <= 256 else None for r in sizes]
for flag in [False, True, False, True]:
    g.disable_rgb_fusion = flag
    stream = torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        graph, static = g.capture_graph(B, shapes)
        for _ in range(3): graph.replay()
        stream.synchronize()
        e0, e1 = _lib.HipEvent(), _lib.HipEvent()
        e0.record(stream.cuda_stream)
        for _ in range(10): graph.replay()
        e1.record(stream.cuda_stream)
        print("fusion disabled" if flag else "fusion enabled ", e0.elapsed_ms(e1) / 10, "ms/step")
