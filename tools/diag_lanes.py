import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from maua_stylegan2_amd import seeding, render
from maua_stylegan2_amd.models.stylegan2 import Generator
torch.set_grad_enabled(False)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 256
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True); g.load_state_dict(seeding.seeded_state_dict(size, seed=5)); g = g.cuda().eval()
n = 6 * bs
lat = seeding.seeded_latents(n, g.n_latent, seed=6)
noise = [torch.from_numpy(seeding.seeded_array(7, f"n{i}", (n, 1, r, r))) if r <= 64 else None for i, r in enumerate(seeding.noise_sizes(size))]
# independent truth: one forward per batch, fully synchronised
truth = []
for k in range(n // bs):
    img, _ = g(styles=lat[k*bs:(k+1)*bs].cuda(), noise=[None if z is None else z[k*bs:(k+1)*bs].cuda() for z in noise],
               truncation=1.0, randomize_noise=False, input_is_latent=True)
    torch.cuda.synchronize()
    u = torch.empty((bs, size, size, 3), dtype=torch.uint8, device="cuda")
    render.frames_to_uint8(img, u); torch.cuda.synchronize()
    truth.append(u.cpu().numpy().astype(np.int16))
truth = np.concatenate(truth)
def run(lanes, use_graph=True, sync_each=False):
    frames = []
    for first, u8 in render.synthesize(g, lat, noise, bs, lanes=lanes, use_graph=use_graph):
        if sync_each:
            torch.cuda.synchronize()
        frames.append(u8.clone())
    torch.cuda.synchronize()
    return torch.cat(frames).cpu().numpy().astype(np.int16)
def cmp(name, x):
    d = np.abs(x - truth)
    print(f"{name}: per-batch max diff vs truth {[int(d[i*bs:(i+1)*bs].max()) for i in range(n // bs)]}")
cmp("eager           ", run(1, use_graph=False))
cmp("eager sync      ", run(1, use_graph=False, sync_each=True))
cmp("graph lanes1    ", run(1))
cmp("graph lanes1 syn", run(1, sync_each=True))
cmp("graph lanes3    ", run(3))
x = run(1)
for k in range(n // bs):
    errs = [float(np.abs(x[k*bs:(k+1)*bs] - truth[j*bs:(j+1)*bs]).mean()) for j in range(n // bs)]
    print("graph batch", k, "mean abs diff to truth batches:", [round(e, 2) for e in errs])
# frame-level: does frame i of graph batch k match frame i of truth?
k = 0
print("batch 0 per-frame max diff:", [int(np.abs(x[i] - truth[i]).max()) for i in range(bs)])
