"""End-to-end run of BASELINE config 3 through the drop-in surface: generate() with the default audio-reactive plugin on a
seeded 1024^2 checkpoint and a synthetic 30 s track (900 frames @30 fps), frames delivered as uint8 NHWC to a counting
sink in host memory (encoder excluded).  Prints preprocessing time and the PCIe-inclusive render rate.

    python tools/e2e_config3.py [--seconds 30] [--size 1024]
"""
import argparse
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maua_stylegan2_amd import generate_audiovisual as gav, render, seeding  # noqa: E402
from maua_stylegan2_amd.audioreactive.examples import default as plugin  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--stage-times", action="store_true", help="print device-synchronised wall time per preprocessing stage")
    ap.add_argument("--repeat", type=int, default=1, help="run generate() this many times in one process (the later runs are warm)")
    ap.add_argument("--gc", choices=["full", "young", "none"], default="full",
                    help="A/B of the collection generate() runs before a LOAD of the generator (the product skips it when the generator is kept from the previous job): "
                         "as the product, generation 0 only, or never")
    args = ap.parse_args()
    work = tempfile.mkdtemp(prefix="maua_e2e_")
    os.chdir(work)
    ckpt = os.path.join(work, "seeded.pt")
    torch.save({"g_ema": seeding.seeded_state_dict(args.size, seed=0)}, ckpt)
    audio = seeding.synthetic_audio(args.seconds)
    wav = os.path.join(work, "track.wav")
    with wave.open(wav, "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((np.clip(audio, -1, 1) * 32767).astype(np.int16).tobytes())

    counted = {"frames": 0, "bytes": 0, "checksum": 0}

    class CountingSink(render.FrameSink):
        def __init__(self, *a, **k):
            self.count = 0

        def write(self, frame):
            counted["frames"] += 1
            counted["bytes"] += frame.nbytes
            if counted["frames"] % 97 == 1:
                counted["checksum"] += int(frame[::16, ::16].sum())
            self.count += 1

        def close(self):
            pass

    render.FrameSink = CountingSink
    if args.stage_times:
        from maua_stylegan2_amd import audioreactive as ar
        from maua_stylegan2_amd.audioreactive import signal as sig

        stages = {}

        def timed(owner, name, label=None):
            fn = getattr(owner, name)

            def wrapper(*a, **k):
                torch.cuda.synchronize()
                t = time.time()
                out = fn(*a, **k)
                torch.cuda.synchronize()
                acc = stages.setdefault(label or name, [0.0, 0])
                acc[0] += time.time() - t
                acc[1] += 1
                return out

            setattr(owner, name, wrapper)

        for owner, name in ((ar, "load_audio"), (ar, "generate_latents"), (plugin, "initialize"), (plugin, "get_latents"),
                            (plugin, "get_noise"), (gav, "load_generator"), (render, "render")):
            timed(owner, name)
        for name in ("onsets", "chroma", "hpss", "onset_strength_bands", "estimate_tuning", "resample", "gaussian_filter", "cqt_magnitude",
                     "nn_filter", "cens", "stft_power", "percentile_clip"):
            timed(sig, name, "  signal." + name)
            if hasattr(ar, name):
                setattr(ar, name, getattr(sig, name))
        timed(gav.gc, "collect", "  gc.collect")
    if args.gc != "full":
        import gc as _gc

        class _Gc:  # (stands in for the module inside generate_audiovisual only)
            def __getattr__(self, name):
                return getattr(_gc, name)

            def collect(self, *a):
                return _gc.collect(0) if args.gc == "young" else 0

        gav.gc = _Gc()
    for run in range(args.repeat):
        counted.update(frames=0, bytes=0, checksum=0)
        if args.stage_times:
            stages.clear()
        t0 = time.time()
        gav.generate(ckpt, wav, initialize=plugin.initialize, get_latents=plugin.get_latents, get_noise=plugin.get_noise,
                     G_res=args.size, out_size=args.size, fps=30, batch=args.batch, output_file=os.path.join(work, "out.mp4"))
        total = time.time() - t0
        if args.stage_times:
            for label, (sec, calls) in stages.items():
                print(f"STAGE {label:28s} {sec:7.3f} s  ({calls} calls)")
        print(f"E2E run {run} frames={counted['frames']} bytes={counted['bytes']} checksum={counted['checksum']} total_wall_s={total:.2f}")


if __name__ == "__main__":
    main()
