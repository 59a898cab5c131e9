"""Audio-reactive video synthesis — drop-in for /root/reference/generate_audiovisual.py on MI355X.

``generate(...)`` keeps the reference's keyword contract (generate_audiovisual.py:59-91), the ``args`` namespace it
builds and extends (:94-113), the callback protocol (initialize / get_latents / get_noise / get_bends / get_rewrites /
get_truncation, :115-186) and the CLI flags + ``OVERRIDE`` dict (:234-299).  Underneath: HIP STFT/mel/chroma features,
HBM-resident latents and noise, hipGraph-replayed generator, uint8 frame epilogue, frame sharding over the GPUs of the
node when launched with torchrun (``--dataparallel`` is accepted and ignored: there is no DataParallel here).

Existing plugin files written for the reference (``import audioreactive as ar``) work unchanged because importing this
module registers ``audioreactive`` / ``render`` / ``models.stylegan2`` / ``op`` aliases for this package's mirrors.
"""
import argparse
import gc
import importlib
import importlib.util
import os
import random
import sys
import time
import traceback
import uuid

import numpy as np
import torch as th

from . import audioreactive as ar
from . import render, sharding
from .models.stylegan2 import Generator


def _install_aliases():
    from . import models, op
    from .models import stylegan2

    for name, mod in [("audioreactive", ar), ("render", render), ("op", op), ("models", models),
                      ("models.stylegan2", stylegan2)]:
        sys.modules.setdefault(name, mod)


_install_aliases()


def get_noise_range(out_size, generator_resolution, is_stylegan1):
    """Number of noise scales for an output size / generator resolution (reference :22-34)."""
    log_max_res = int(np.log2(out_size))
    log_min_res = 2 + (log_max_res - int(np.log2(generator_resolution)))
    if is_stylegan1:
        return log_min_res, log_max_res + 1, (lambda x: x)
    return 2 * log_min_res + 1, 2 * (log_max_res + 1), (lambda x: int(x / 2))


def load_generator(ckpt, is_stylegan1, G_res, out_size, noconst, latent_dim, n_mlp, channel_multiplier, dataparallel,
                   base_res_factor):
    """Reference :37-56.  Rank 0 reads the checkpoint; other ranks receive the weights over RCCL."""
    if is_stylegan1:
        raise NotImplementedError("--stylegan1: only the StyleGAN2 generator is built (SURVEY.md §2 row 13)")
    rank, world = sharding.rank_world()
    generator = Generator(G_res, latent_dim, n_mlp, channel_multiplier=channel_multiplier, constant_input=not noconst,
                          checkpoint=ckpt if rank == 0 else None, output_size=out_size,
                          base_res_factor=base_res_factor).cuda()
    sharding.broadcast_module(generator)
    return generator.eval()


def generate(ckpt, audio_file, initialize=None, get_latents=None, get_noise=None, get_bends=None, get_rewrites=None,
             get_truncation=None, output_dir="./output", audioreactive_file="audioreactive/examples/default.py",
             offset=0, duration=-1, latent_file=None, shuffle_latents=False, G_res=1024, out_size=1024, fps=30,
             latent_count=12, batch=8, dataparallel=False, truncation=1.0, stylegan1=False, noconst=False,
             latent_dim=512, n_mlp=8, channel_multiplier=2, randomize_noise=False, ffmpeg_preset="slow",
             base_res_factor=1, output_file=None, args=None):
    if args is None:  # called directly (notebook): build the namespace from the local variables, as the reference does
        kwargs = locals()
        args = argparse.Namespace()
        for k, v in kwargs.items():
            setattr(args, k, v)

    ar.set_SMF(args.fps / 30)  # smoothing independent of frame rate
    time_taken = time.time()
    th.set_grad_enabled(False)

    audio, sr, duration = ar.load_audio(audio_file, offset, duration)
    args.audio = audio
    args.sr = sr
    n_frames = int(round(duration * fps))
    args.duration = duration
    args.n_frames = n_frames

    if initialize is not None:
        args = initialize(args)

    from .audioreactive.examples import default as default_plugin

    print("\ngenerating latents...")
    if get_latents is None:
        get_latents = default_plugin.get_latents
    if latent_file is not None:
        latent_selection = ar.load_latents(latent_file)
    else:
        latent_selection = ar.generate_latents(args.latent_count, ckpt, G_res, noconst, latent_dim, n_mlp, channel_multiplier)
        latent_selection = sharding.broadcast_tensor(latent_selection.cuda()).cpu()
    if shuffle_latents:
        random_indices = random.sample(range(len(latent_selection)), len(latent_selection))
        latent_selection = latent_selection[random_indices]
    os.makedirs("workspace", exist_ok=True)
    np.save("workspace/last-latents.npy", latent_selection.numpy())

    latents = get_latents(selection=latent_selection, args=args)
    print(f"{list(latents.shape)} amplitude={latents.std()}\n")

    print("generating noise...")
    if get_noise is None:
        get_noise = default_plugin.get_noise
    noise = []
    range_min, range_max, exponent = get_noise_range(out_size, G_res, stylegan1)
    for scale in range(range_min, range_max):
        h = (2 if out_size == 1080 else 1) * 2 ** exponent(scale)
        w = (2 if out_size == 1920 else 1) * 2 ** exponent(scale)
        noise.append(get_noise(height=h, width=w, scale=scale - range_min, num_scales=range_max - range_min, args=args))
        if noise[-1] is not None:
            print(list(noise[-1].shape), f"amplitude={noise[-1].std()}")
        gc.collect()
    print()

    if get_bends is not None:
        print("generating network bends...")
        bends = get_bends(args=args)
    else:
        bends = []
    if get_rewrites is not None:
        print("generating model rewrites...")
        rewrites = get_rewrites(args=args)
    else:
        rewrites = {}
    if get_truncation is not None:
        print("generating truncation...")
        truncation = get_truncation(args=args)
    else:
        truncation = float(truncation)

    # one process per GPU: callbacks run on every rank (they may draw random numbers), rank 0's results win
    if sharding.rank_world()[1] > 1:
        latents = sharding.broadcast_tensor(latents.cuda().float().contiguous())
        noise = [None if nz is None else sharding.broadcast_tensor(nz.cuda().float().contiguous()) for nz in noise]
        if not isinstance(truncation, float):
            truncation = sharding.broadcast_tensor(truncation.cuda().float().contiguous())
        for bend in bends:
            if "modulation" in bend:
                bend["modulation"] = sharding.broadcast_tensor(bend["modulation"].cuda().float().contiguous())

    gc.collect()
    generator = load_generator(ckpt=ckpt, is_stylegan1=stylegan1, G_res=G_res, out_size=out_size, noconst=noconst,
                               latent_dim=latent_dim, n_mlp=n_mlp, channel_multiplier=channel_multiplier,
                               dataparallel=dataparallel, base_res_factor=base_res_factor)
    print(f"\npreprocessing took {time.time() - time_taken:.2f}s\n")

    print(f"rendering {n_frames} frames...")
    if output_file is None:
        checkpoint_title = str(ckpt).split("/")[-1].split(".")[0].lower()
        track_title = audio_file.split("/")[-1].split(".")[0].lower()
        output_file = f"{output_dir}/{track_title}_{checkpoint_title}_{uuid.uuid4().hex[:8]}.mp4"
    t0 = time.time()
    n_written = render.render(generator=generator, latents=latents, noise=noise, audio_file=audio_file, offset=offset,
                              duration=duration, batch_size=batch, truncation=truncation, bends=bends, rewrites=rewrites,
                              out_size=out_size, output_file=output_file, randomize_noise=randomize_noise,
                              ffmpeg_preset=ffmpeg_preset)
    dt = time.time() - t0
    print(f"\nrendered {n_written} frames in {dt:.2f}s ({n_written / max(dt, 1e-9):.1f} frames/s)")
    print(f"total time taken: {(time.time() - time_taken)/60:.2f} minutes")
    return output_file


def load_plugin(audioreactive_file):
    """Reference :262-292: resolve the six optional callbacks and the OVERRIDE dict from a plugin file."""
    if os.path.exists(audioreactive_file):
        spec = importlib.util.spec_from_file_location("maua_audioreactive_plugin", audioreactive_file)
        module = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(module)
    else:
        module = importlib.import_module(audioreactive_file.replace(".py", "").replace("/", "."))
    funcs = {}
    for func in ["initialize", "get_latents", "get_noise", "get_bends", "get_rewrites", "get_truncation"]:
        funcs[func] = getattr(module, func, None)
        if funcs[func] is None:
            print(f"No '{func}' function found in --audioreactive_file, using default...")
    return funcs, dict(getattr(module, "OVERRIDE", {}))


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--ckpt", type=str)
    parser.add_argument("--audio_file", type=str)
    parser.add_argument("--audioreactive_file", type=str, default="audioreactive/examples/default.py")
    parser.add_argument("--output_dir", type=str, default="./output")
    parser.add_argument("--offset", type=float, default=0)
    parser.add_argument("--duration", type=float, default=-1, help="length of rendered video in seconds")
    parser.add_argument("--latent_file", type=str, default=None)
    parser.add_argument("--shuffle_latents", action="store_true")
    parser.add_argument("--G_res", type=int, default=1024)
    parser.add_argument("--out_size", type=int, default=1024, help="rendered video size. Options: 512, 1024, 1920")
    parser.add_argument("--fps", type=int, default=30)
    parser.add_argument("--latent_count", type=int, default=12)
    parser.add_argument("--batch", type=int, default=8)
    parser.add_argument("--dataparallel", action="store_true")
    parser.add_argument("--truncation", type=float, default=1.0)
    parser.add_argument("--stylegan1", action="store_true")
    parser.add_argument("--noconst", action="store_true")
    parser.add_argument("--latent_dim", type=int, default=512)
    parser.add_argument("--n_mlp", type=int, default=8)
    parser.add_argument("--channel_multiplier", type=int, default=2)
    parser.add_argument("--randomize_noise", action="store_true")
    parser.add_argument("--base_res_factor", type=float, default=1)
    parser.add_argument("--ffmpeg_preset", type=str, default="slow")
    parser.add_argument("--output_file", type=str, default=None)
    args = parser.parse_args(argv)
    os.makedirs(args.output_dir, exist_ok=True)

    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not th.distributed.is_initialized():
        th.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        th.distributed.init_process_group("nccl")

    plugin = args.audioreactive_file
    if plugin == "audioreactive/examples/default.py" and not os.path.exists(plugin):
        plugin = os.path.join(os.path.dirname(os.path.abspath(__file__)), "audioreactive", "examples", "default.py")
    try:
        funcs, override = load_plugin(plugin)
    except Exception:
        print("Error while loading --audioreactive_file...")
        traceback.print_exc()
        sys.exit(1)
    arg_dict = vars(args).copy()
    for arg, val in override.items():
        arg_dict[arg] = val
        setattr(args, arg, val)
    ckpt = arg_dict.pop("ckpt", None)
    audio_file = arg_dict.pop("audio_file", None)
    generate(ckpt=ckpt, audio_file=audio_file, **funcs, **arg_dict, args=args)


if __name__ == "__main__":
    main()
