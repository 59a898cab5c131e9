"""Where the render stage of a 900-frame job spends its time beyond the replays: times render.synthesize alone (frames left in
HBM) for 896 / 900 frames and render.render with a counting host sink, warm, on one MI355X.
    python tools/render_probe.py [--frames 900] [--repeat 3]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import render, seeding  # noqa: E402
from maua_stylegan2_amd.models.stylegan2 import Generator  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=900)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = Generator(args.size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(args.size, seed=0))
    g = g.to(dev).eval()
    n = args.frames
    lat = torch.randn(n, g.n_latent, 512, device=dev)
    noise = [torch.randn(n, 1, r, r, device=dev) if r <= 256 else None for r in seeding.noise_sizes(args.size)]

    def synth(count):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in render.synthesize(g, lat[:count], [None if z is None else z[:count] for z in noise], args.batch):
            pass
        torch.cuda.synchronize()
        return time.perf_counter() - t

    class Sink(render.FrameSink):
        def __init__(self, *a, **k):
            self.count = 0

        def write(self, frame):
            self.count += 1

        def close(self):
            pass

    render.FrameSink = Sink

    def full(count):
        torch.cuda.synchronize()
        t = time.perf_counter()
        render.render(g, lat[:count], [None if z is None else z[:count] for z in noise], 0, count / 30.0, args.batch, args.size, "x.mp4")
        torch.cuda.synchronize()
        return time.perf_counter() - t

    whole = (n // args.batch) * args.batch
    synth(whole)  # captures the lanes
    if os.environ.get("MAUA_PROBE_CPROFILE"):
        import cProfile
        import pstats

        full(n)
        pr = cProfile.Profile()
        pr.enable()
        full(n)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(18)
        pr = cProfile.Profile()
        pr.enable()
        synth(n)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(10)
        return
    for r in range(args.repeat):
        a, b = synth(whole), synth(n)
        c, d = full(whole), full(n)
        print(f"run {r}: synthesize {whole} frames {a:.3f} s ({whole / a:.0f}/s) | {n} frames {b:.3f} s ({n / b:.0f}/s) | "
              f"render {whole} frames {c:.3f} s ({whole / c:.0f}/s) | {n} frames {d:.3f} s ({n / d:.0f}/s)", flush=True)


if __name__ == "__main__":
    main()
