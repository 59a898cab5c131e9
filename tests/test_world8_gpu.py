"""GPU: BASELINE configs 4 and 5 AT THEIR STATED SHAPE — the 1024^2 generator, 60 s at 30 fps = 1800 frames, EIGHT contiguous shards of
225 frames (reference render.py:140-182 slices the same ranges; the DataParallel role of generate_audiovisual.py:54-55) — on the one
MI355X a box has: ``tests/played_world.py`` plays the eight ranks one after another through the product's unmodified multi-rank code
(flat weight broadcast, object broadcasts, scatter of the per-frame inputs, one asynchronous gather per batch-round into rank 0's HBM store,
pinned ring, sink thread); only the transport between the ranks is an in-process record / replay.  What a real 8-GPU node adds to this is
RCCL moving the same bytes over xGMI, and the rate.

Equality rule: a frame rendered inside a full batch of 8 is bit-identical whichever job / rank / lane rendered it.  Each 225-frame block
ends in a ONE-frame eager tail (225 = 28 x 8 + 1) whose 4^2..32^2 split-K layers run at another depth than in a batch of 8: those eight
frames (224, 449, ..., 1799) may differ from the single-rank job by one grey level (models/stylegan2.py:492-576 has no batch coupling;
the ulp comes from the summation order of this implementation's split-K slabs)."""
import gc
import os
import random
import wave

import numpy as np
import pytest
import torch
import xxhash

from maua_stylegan2_amd import seeding

from played_world import PlayedWorld

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)

SIZE, FPS, SECONDS, WORLD, BATCH = 1024, 30, 60.0, 8, 8
N = int(round(SECONDS * FPS))  # 1800
PER = N // WORLD  # 225
TAILS = {r * PER + PER - 1 for r in range(WORLD)}  # the one-frame eager tail of every block


def _digest_sink(render, keep):
    """FrameSink stand-in: xxh64 of every delivered frame (in delivery order) + full copies of the frames in ``keep``."""
    log = {"digests": [], "kept": {}}

    class DigestSink(render.FrameSink):
        def __init__(self, *a, **k):
            self.count = 0

        def write(self, frame):
            assert frame.shape == (SIZE, SIZE, 3) and frame.dtype == np.uint8
            log["digests"].append(xxhash.xxh64_intdigest(memoryview(np.ascontiguousarray(frame)).cast("B")))
            if self.count in keep:
                log["kept"][self.count] = np.array(frame, copy=True)
            self.count += 1

        def close(self):
            pass

    return DigestSink, log


def _free_device_memory():
    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def _compare_jobs(one, eight, keep):
    """The 8-rank delivery against the 1-rank delivery: every frame once and in order (digests), bit-identical outside the eager tails,
    within one grey level on them."""
    assert len(one["digests"]) == len(eight["digests"]) == N
    differing = [i for i in range(N) if one["digests"][i] != eight["digests"][i]]
    assert set(differing) <= TAILS, f"frames outside the eager tails differ between the 1-rank and the 8-rank job: {differing[:12]}"
    assert len(set(eight["digests"])) > N - 8, "frames repeat: a rank rendered the wrong block"
    worst = 0
    for i in sorted(TAILS & set(keep)):
        diff = np.abs(one["kept"][i].astype(np.int16) - eight["kept"][i].astype(np.int16))
        worst = max(worst, int(diff.max()))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-2, (i, int(diff.max()), float((diff > 0).mean()))
    return differing, worst


def test_world8_config4_1800_frames_of_the_1024_generator_equal_the_single_rank_render(gpu, monkeypatch):
    """BASELINE config 4 through render(): 1800 frames, batches of 8, 3 graph lanes per rank, per-frame noise up to 64^2 and checkpoint
    buffers above.  (i) the single-rank render (no process group: pinned ring + sink thread); (ii) eight played ranks, each through
    render_shard's gather transport — scatter of latents / noise from rank 0, 29 gathers per rank, rank 0's sink delivering all 1800 frames.
    (ii) == (i) frame for frame (see the equality rule above); three frames of (ii) — rank 1's second frame, rank 4's eager tail, a
    mid-block frame of rank 7 — against the ORACLE generator (<= 1 grey level on an unsaturated checkpoint)."""
    from maua_stylegan2_amd import render, sharding
    from maua_stylegan2_amd.models.stylegan2 import Generator
    from oracle import stylegan2_oracle as so

    sd = seeding.seeded_state_dict(SIZE, seed=0, rgb_gain=0.12)
    g = Generator(SIZE, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(sd, strict=True)
    g = g.to(gpu).eval()
    lat = seeding.seeded_latents(N, 18, seed=300)
    noise = [torch.from_numpy(seeding.seeded_array(301, f"n{i}", (N, 1, r, r))) if r <= 64 else None
             for i, r in enumerate(seeding.noise_sizes(SIZE))]
    picks = [PER + 1, 4 * PER + PER - 1, 7 * PER + 100]
    keep = set(picks) | TAILS

    sink_cls, one = _digest_sink(render, keep)
    monkeypatch.setattr(render, "FrameSink", sink_cls)
    assert render.render(g, lat, noise, 0, SECONDS, BATCH, SIZE, None) == N

    world = PlayedWorld(WORLD)
    logs = {}

    def job(rank, final):
        sink_cls, logs[(rank, final)] = _digest_sink(render, keep)
        monkeypatch.setattr(render, "FrameSink", sink_cls)
        dev = torch.device(gpu)
        lat_r = sharding.scatter_frames(lat.to(dev) if rank == 0 else None, N, device=dev)
        noise_r = [sharding.scatter_frames(None if nz is None or rank else nz.to(dev), N, device=dev) for nz in noise]
        lo, hi = sharding.shard_bounds(N, rank, WORLD)
        assert (lo, hi) == (rank * PER, (rank + 1) * PER) and lat_r.shape[0] == PER
        assert [nz is None for nz in noise_r] == [nz is None for nz in noise]
        written = render.render_shard(g, lat_r, noise_r, 0, SECONDS, BATCH, SIZE, None, None, 1.0, [], {}, False, "slow", (lo, hi, N))
        torch.cuda.synchronize()
        return written

    results = world.play(job, reseed=_free_device_memory)
    assert results == [N] + [0] * (WORLD - 1) + [N]  # (rank 0's first pass hands out the peers' still empty slots: its sink is discarded)
    assert all(len(world.rounds[p]) == 29 for p in range(1, WORLD)), {p: len(v) for p, v in world.rounds.items()}
    eight = logs[(0, True)]
    differing, worst = _compare_jobs(one, eight, keep)
    print(f"[config 4 shape, 8 played ranks x 225 frames @1024^2] {N - len(differing)} of {N} frames bit-identical to the single-rank render, "
          f"{len(differing)} eager-tail frames differ by <= {worst} grey level; collectives: {world.stats}")
    assert world.stats["bytes_gathered"] == 7 * 29 * BATCH * SIZE * SIZE * 3
    for i in picks:
        noise_i = [sd[f"noises.noise_{k}"] if nz is None else nz[i: i + 1] for k, nz in enumerate(noise)]
        want_f = so.generator_forward(sd, lat[i: i + 1], noise_i)
        assert seeding.clamped_fraction(want_f) < 0.05
        diff = np.abs(eight["kept"][i].astype(np.int16) - so.frames_to_uint8(want_f)[0].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))


def test_world8_config5_bends_and_device_front_end_through_generate(gpu, tmp_path, monkeypatch):
    """BASELINE config 5 through the drop-in surface: ``generate()`` on a 60 s track with the default plugin's callbacks (HIP HPSS / onset /
    chroma kernels, chroma-weighted latents, reactive noise <= 256^2) PLUS network bending — a Translate at layer id 4 driven by the bass
    onsets with th.randn-drawn bend noise and a Zoom at layer id 5 driven by the treble onsets (usage audioreactive/examples/tauceti.py:94-159,
    transforms audioreactive/bend.py:52-102), captured inside every rank's graph lanes.  A plugin with get_bends runs its front end on
    EVERY rank (closures cannot travel), re-seeded from one broadcast seed.  Eight played ranks (each: load_generator -> flat weight
    broadcast -> render.prepare -> front end -> scatter -> 29 gathers) against the same job under a process group of ONE rank: delivered
    videos equal frame for frame (equality rule above); three frames against the ORACLE generator with the oracle's own warps."""
    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render
    from maua_stylegan2_amd.audioreactive import bend
    from maua_stylegan2_amd.audioreactive.examples import default as plugin
    from oracle import signal_oracle
    from oracle import stylegan2_oracle as so

    monkeypatch.chdir(tmp_path)
    sd = seeding.seeded_state_dict(SIZE, seed=0, rgb_gain=0.12)
    torch.save({"g_ema": sd}, "seeded1024.pt")
    audio = seeding.synthetic_audio(SECONDS)
    with wave.open("track.wav", "wb") as f:
        f.setnchannels(1), f.setsampwidth(2), f.setframerate(22050)
        f.writeframes((np.clip(audio, -1, 1) * 32767).astype(np.int16).tobytes())
    np.save("selection.npy", seeding.seeded_array(42, "selection", (12, 18, 512)))
    h = w = 16
    picks = [0, 4 * PER + 100, N - 1]
    keep = set(picks) | TAILS
    seen = {}

    def get_latents(selection, args):
        seen["latents"] = plugin.get_latents(selection, args)
        return seen["latents"]

    def get_noise(height, width, scale, num_scales, args):
        nz = plugin.get_noise(height, width, scale, num_scales, args)
        seen.setdefault("noise", []).append(None if nz is None else nz.detach().cpu())
        return nz

    def get_bends(args):
        lo_on, hi_on = args.lo_onsets.detach().float().cpu(), args.hi_onsets.detach().float().cpu()
        bnoise = 0.2 * torch.randn(1, 1, h, 5 * w)  # drawn inside the callback, as tauceti.py does: must agree across ranks
        saw = (torch.arange(args.n_frames, dtype=torch.float32) * 0.37 + 6.0 * lo_on) % (1.5 * w)  # scrolls by more than one width
        shift = torch.stack([saw, torch.zeros(args.n_frames)], 1)
        zoom = 1.0 + 0.25 * hi_on
        seen.update(bnoise=bnoise, shift=shift, zoom=zoom)
        return [{"layer": 4, "modulation": shift.clone(), "transform": lambda b: bend.Translate(b, h, w, bnoise)},
                {"layer": 5, "modulation": zoom.clone(), "transform": lambda b: bend.Zoom(b, h, w)}]

    kw = dict(ckpt="seeded1024.pt", audio_file="track.wav", initialize=plugin.initialize, get_latents=get_latents, get_noise=get_noise,
              get_bends=get_bends, latent_file="selection.npy", G_res=SIZE, out_size=SIZE, fps=FPS, batch=BATCH)

    def reseed():
        _free_device_memory()
        assert torch.cuda.memory_allocated() < 40 * 2 ** 30, "a previous rank's generator / lanes are still alive"
        random.seed(123), np.random.seed(123), torch.manual_seed(123), torch.cuda.manual_seed_all(123)

    logs, bend_noise = {}, {}

    def run(world_size):
        world = PlayedWorld(world_size)

        def job(rank, final):
            seen.clear()
            sink_cls, logs[(world_size, rank, final)] = _digest_sink(render, keep)
            monkeypatch.setattr(render, "FrameSink", sink_cls)
            gav.generate(output_file=str(tmp_path / f"w{world_size}_r{rank}.mp4"), **kw)
            torch.cuda.synchronize()
            bend_noise[(world_size, rank, final)] = seen["bnoise"].clone()
            return dict(seen) if rank == 0 and final else None

        return world, world.play(job, reseed=reseed)

    world1, res1 = run(1)
    world8, res8 = run(WORLD)
    one, eight = logs[(1, 0, True)], logs[(WORLD, 0, True)]
    front = res8[-1]  # what rank 0's callbacks returned in the delivering pass
    assert tuple(front["latents"].shape) == (N, 18, 512) and len(front["noise"]) == 17
    assert torch.equal(front["latents"].cpu(), res1[-1]["latents"].cpu()), "the front end is not reproducible from the seeds"
    for key, val in bend_noise.items():  # every rank drew the SAME bend noise (re-seeded immediately before get_bends)
        assert torch.equal(val, bend_noise[(WORLD, 0, True)]), key
    assert all(len(world8.rounds[p]) == 29 for p in range(1, WORLD))
    assert world8.stats["scatter"] > 0 and world8.stats["broadcast"] > 0 and world8.stats["gather"] == (WORLD + 1) * 29
    differing, worst = _compare_jobs(one, eight, keep)
    print(f"[config 5 shape, 8 played ranks through generate()] {N - len(differing)} of {N} frames bit-identical to the one-rank job, "
          f"{len(differing)} eager-tail frames differ by <= {worst} grey level; collectives of the 8-rank job: {world8.stats}")

    shift, zoom, bnoise = front["shift"], front["zoom"], front["bnoise"]
    assert float(zoom.max()) > 1.05 and float(shift[:, 0].max()) > w, "the modulations must move the image for this test to mean anything"

    def o_translate(i):
        pads = [(int(w / 2), int(w / 2), 0, 0), (w, w, 0, 0), (w, 0, 0, 0)]
        m = bend._inverse_maps_translate(shift[i: i + 1]).numpy()
        return lambda t: torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, pads, bnoise.numpy())).float()

    def o_zoom(i):
        pad = max(h, w) - 1
        m = bend._inverse_maps_scale(zoom[i: i + 1], w + 2 * pad, h + 2 * pad).numpy()
        return lambda t: torch.from_numpy(signal_oracle.affine_reflect_warp(t.numpy(), m, (pad,) * 4)).float()

    for i in picks:
        noise_i = [sd[f"noises.noise_{k}"] if nz is None else nz[i: i + 1] for k, nz in enumerate(front["noise"])]
        lat_i = front["latents"][i: i + 1].cpu().float()
        want_f = so.generator_forward(sd, lat_i, noise_i, bends={4: o_translate(i), 5: o_zoom(i)})
        plain = so.generator_forward(sd, lat_i, noise_i)
        print(f"[config 5 shape, frame {i}] oracle image std {float(want_f.std()):.3f}, clamped {100 * seeding.clamped_fraction(want_f):.1f} %, "
              f"bends move the image by {float((want_f - plain).abs().mean()):.3f} on average")
        assert seeding.clamped_fraction(want_f) < 0.05 and float((want_f - plain).abs().mean()) > 0.02
        diff = np.abs(eight["kept"][i].astype(np.int16) - so.frames_to_uint8(want_f)[0].astype(np.int16))
        assert diff.max() <= 1 and (diff > 0).mean() < 2e-2, (i, int(diff.max()), float((diff > 0).mean()))
