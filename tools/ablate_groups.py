"""Marginal cost of layer groups INSIDE the overlapped pipeline (timing experiment; frames are wrong by construction).

bench.py's `layers` rows are isolated launches; under three graph lanes a short launch may cost less (it fills gaps of the other lanes) or
more (it competes for queues) than its isolated duration.  This tool captures the bench workload with the launches of one group of layers
left out and reports frames/s against the full forward: the difference is what the group costs where it runs.

    python tools/ablate_groups.py [--lanes 3] [--steps 6]

Groups: `small` = conv1 ... convs.4 with their tails and ToRGB passes (the 4^2 ... 32^2 block), `tails` = the blur + noise + act launches
of the two-launch up-sampling layers, `fused` = convs.12 / convs.14 as one kernel, `w2dw` = convs.15, `rgb` = separate ToRGB passes."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
from maua_stylegan2_amd import _lib  # noqa: E402
from maua_stylegan2_amd.models import stylegan2 as sg  # noqa: E402

SKIP = set()
_orig_run = sg.StyledConv.run
_orig_rgb = sg.ToRGB.run
_orig_blur = None


def styled_run(self, x, s, s_off, d, noise, bufs, tag, rgb=None, **kw):
    if tag in SKIP:
        b, _, h, w = x.shape
        up = 2 if self.conv.upsample else 1
        self.posted = kw.get("post_off") is not None  # (the consumer keeps its pre-scaled instance)
        if rgb is not None:
            rgb["done"] = True
        return bufs(tag, (b, self.conv.out_channel, up * h, up * w))
    return _orig_run(self, x, s, s_off, d, noise, bufs, tag, rgb=rgb, **kw)


def rgb_run(self, x, s, s_off, skip, out):
    if "rgb" in SKIP or ("small" in SKIP and x.shape[-1] <= 16):
        return out
    return _orig_rgb(self, x, s, s_off, skip, out)


class LibProxy:
    """Forwards to the library; drops maua_blur_noise_act_f32 launches when `tails` is skipped."""

    def __init__(self, lib):
        self._lib = lib

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name == "maua_blur_noise_act_f32" and "tails" in SKIP:
            return lambda *a: 0
        return fn


def measure(args, dev):
    wl = bench.Workload(1024, 8, args.lanes, dev, 0, False)
    bench.time_region(wl, 2, 15, "synth", False, 1)
    best = 0.0
    for _ in range(args.repeat):
        dt = bench.time_region(wl, args.steps, 15, "synth", False, 1)
        best = max(best, args.steps * 15 * 8 / dt)
    del wl
    torch.cuda.empty_cache()
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--groups", default="none,small,tails,fused,w2dw,none")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    sg.StyledConv.run = styled_run
    sg.ToRGB.run = rgb_run
    real_load = _lib.load
    proxy = LibProxy(real_load())
    sg._lib.load = lambda: proxy
    sets = {"none": set(), "small": {"small", "conv1", "convs.0", "convs.1", "convs.2", "convs.3", "convs.4"}, "tails": {"tails"},
            "fused": {"convs.12", "convs.14"}, "w2dw": {"convs.15"}, "rgb": {"rgb"}, "convs.5": {"convs.5"},
            "mid": {"convs.6", "convs.7", "convs.8", "convs.9", "convs.10", "convs.11"}, "convs.13": {"convs.13"}}
    out = {}
    base = None
    for name in args.groups.split(","):
        SKIP.clear()
        SKIP.update(sets[name])
        v = measure(args, dev)
        if name == "none" and base is None:
            base = v
        out.setdefault(name, []).append(v)
        ms = 8000.0 / v
        print(f"skip {name:8s}: {v:8.1f} frames/s  {ms:.3f} ms/batch  saves {8000.0 / base - ms:+.3f} ms/batch", flush=True)
    print(json.dumps({"lanes": args.lanes, "frames_per_s": out}))


if __name__ == "__main__":
    main()
