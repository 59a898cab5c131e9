"""GPU: the MULTI-RANK paths on the device.  The boxes here have one GPU and RCCL wants one GPU per rank, so the two ranks of these tests are
two processes that SHARE the GPU with their process group over gloo (which moves device tensors for broadcast / scatter / gather): everything
but the RCCL transport itself runs for real — flat weight broadcast, scatter of the per-frame inputs, captured lanes per rank, one
asynchronous gather per batch-round into rank 0's HBM store, the pinned staging ring, the sink thread, and the alternative per-rank D2H
transport through pinned shared memory (sharding.HostFrameStore, MAUA_FRAME_TRANSPORT=host).  The CPU twins are in
tests/test_sharding_gloo.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, n_frames, batch, h, w):
    import threading

    from maua_stylegan2_amd import render, sharding

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        token = sharding.broadcast_object(f"g{os.getpid():x}" if rank == 0 else None)
        lo, hi = sharding.shard_bounds(n_frames, rank, world)
        store = sharding.HostFrameStore(n_frames, batch, (h, w, 3), dev, token)
        assert store._registered, "the shared-memory segment must be pinned in place for the copies to be asynchronous"
        seen = []

        class Sink:
            def write(self, frame):
                assert frame.shape == (h, w, 3)
                seen.append((int(frame[0, 0, 0]), int(frame[h - 1, w - 1, 2]), int(frame.astype(np.int64).sum())))

        worker = reader = None
        if rank == 0:
            worker = render.SinkWorker(Sink())

            def run():
                for _, count, host in store.rounds_in_order():
                    worker.submit(None, host.numpy(), count, None)

            reader = threading.Thread(target=run, daemon=True)
            reader.start()
        stream = torch.cuda.Stream(dev)
        produced = torch.empty((batch, h, w, 3), dtype=torch.uint8, device=dev)  # ONE device buffer, overwritten every round like a lane's
        k = 0
        with torch.cuda.stream(stream):
            for first in range(lo, hi, batch):
                count = min(batch, hi - first)
                for i in range(count):
                    produced[i].fill_((first + i) % 251)
                    produced[i, h - 1, w - 1, 2] = (first + i) % 7
                store.push(k, produced[:count])
                k += 1
        store.finish()
        if rank == 0:
            reader.join(timeout=60)
            assert not reader.is_alive()
            worker.close()
            assert len(seen) == n_frames
            for i, (first_px, last_px, total) in enumerate(seen):
                assert first_px == i % 251 and last_px == i % 7, (i, first_px, last_px)
                assert total == (h * w * 3 - 1) * (i % 251) + i % 7, i
            np.save(os.path.join(out_dir, "ok.npy"), np.array([n_frames]))
        store.close()
    finally:
        dist.destroy_process_group()


def test_host_frame_store_with_pinned_segments_on_the_device(gpu, tmp_path):
    """Two ranks on one GPU: every rank's rounds reach its own pinned shared-memory segment by asynchronous D2H copies (the device buffer is
    reused every round: a copy that had not finished before the next round overwrote it would show up as a wrong frame), rank 0's reader
    thread + SinkWorker deliver all frames in global order."""
    n_frames, batch, h, w = 37, 4, 96, 128
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), n_frames, batch, h, w), nprocs=2, join=True)
    assert int(np.load(tmp_path / "ok.npy")[0]) == n_frames


def _render_worker(rank, world, port, out_dir, n_frames, batch):
    from maua_stylegan2_amd import render, seeding, sharding
    from maua_stylegan2_amd.models.stylegan2 import Generator

    torch.set_grad_enabled(False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        size = 64
        g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        g.load_state_dict(seeding.seeded_state_dict(size, seed=4), strict=True)
        g = g.to(dev).eval()
        lat = seeding.seeded_latents(n_frames, g.n_latent, seed=6)
        noise = seeding.seeded_noise(n_frames, size, seed=7)
        noise[-1] = None
        frames = []

        class KeepingSink(render.FrameSink):
            def __init__(self, *a, **k):
                self.count = 0

            def write(self, frame):
                frames.append(np.array(frame, copy=True))
                self.count += 1

            def close(self):
                pass

        render.FrameSink = KeepingSink
        # (out_size must be one of the reference's sizes: render through the internal entry with the generator's own frame shape)
        import maua_stylegan2_amd.render as R

        keep = R._output_dims
        R._output_dims = lambda out_size: (size, size)
        try:
            written = R.render_shard(g, lat, noise, 0, n_frames / 30.0, batch, size, None, None, 1.0, [], {}, False, "slow", None,
                                     transport="host")
        finally:
            R._output_dims = keep
        if rank == 0:
            assert written == n_frames and len(frames) == n_frames
            want = np.zeros((n_frames, size, size, 3), np.uint8)
            for first, u8 in render.synthesize(g, lat, noise, batch):
                want[first: first + u8.shape[0]] = u8.cpu().numpy()
            got = np.stack(frames)
            assert np.array_equal(got, want), "frames of the two-rank render differ from the single-rank render"
            lo1, _ = sharding.shard_bounds(n_frames, 1, world)
            assert not np.array_equal(got[lo1], got[0])
            np.save(os.path.join(out_dir, "render_ok.npy"), np.array([n_frames]))
        else:
            assert written == 0
    finally:
        dist.destroy_process_group()


def test_two_rank_render_through_the_host_transport_equals_single_rank(gpu, tmp_path):
    """The multi-rank branch of render_shard ON THE DEVICE with the host transport: two processes (sharing the box's one GPU; process group
    over gloo) each render their contiguous block of a 23-frame sequence through captured lanes + an eager tail and hand the rounds to rank
    0's sink through pinned shared memory; the delivered frames equal the single-rank render bit for bit and arrive in order."""
    n_frames, batch = 23, 4
    mp.spawn(_render_worker, args=(2, _free_port(), str(tmp_path), n_frames, batch), nprocs=2, join=True)
    assert int(np.load(tmp_path / "render_ok.npy")[0]) == n_frames


def _render_gather_worker(rank, world, port, out_dir, n_frames, batch):
    """As _render_worker, default transport: one asynchronous gather of every batch-round into rank 0's HBM store (sharding.FrameStream),
    pinned ring + sink thread on rank 0."""
    from maua_stylegan2_amd import render, seeding, sharding
    from maua_stylegan2_amd.models.stylegan2 import Generator
    import maua_stylegan2_amd.render as R

    torch.set_grad_enabled(False)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        size = 64
        g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        if rank == 0:
            g.load_state_dict(seeding.seeded_state_dict(size, seed=4), strict=True)
        g = g.to(dev).eval()
        sharding.broadcast_module(g)  # rank 1 starts from its own random init: the weights come from rank 0
        lat = seeding.seeded_latents(n_frames, g.n_latent, seed=6)
        noise = seeding.seeded_noise(n_frames, size, seed=7)
        frames = []

        class KeepingSink(render.FrameSink):
            def __init__(self, *a, **k):
                self.count = 0

            def write(self, frame):
                frames.append(np.array(frame, copy=True))
                self.count += 1

            def close(self):
                pass

        render.FrameSink = KeepingSink
        keep = R._output_dims
        R._output_dims = lambda out_size: (size, size)
        try:
            written = R.render_shard(g, lat, noise, 0, n_frames / 30.0, batch, size, None, None, 1.0, [], {}, False, "slow", None)
        finally:
            R._output_dims = keep
        if rank == 0:
            assert written == n_frames and len(frames) == n_frames
            want = np.zeros((n_frames, size, size, 3), np.uint8)
            for first, u8 in render.synthesize(g, lat, noise, batch):
                want[first: first + u8.shape[0]] = u8.cpu().numpy()
            assert np.array_equal(np.stack(frames), want), "frames of the two-rank render differ from the single-rank render"
            np.save(os.path.join(out_dir, "gather_ok.npy"), np.array([n_frames]))
    finally:
        dist.destroy_process_group()


def test_two_rank_render_through_the_gather_transport_equals_single_rank(gpu, tmp_path):
    """The DEFAULT multi-rank branch of render_shard on the device — flat weight broadcast, one asynchronous gather per batch-round into rank
    0's HBM store, pinned ring, sink thread — under two processes sharing the box's GPU (gloo moves the device tensors; RCCL needs one GPU
    per rank, which no box here has): frames bit-equal to the single-rank render, in order, ragged last rounds included."""
    n_frames, batch = 23, 4
    mp.spawn(_render_gather_worker, args=(2, _free_port(), str(tmp_path), n_frames, batch), nprocs=2, join=True)
    assert int(np.load(tmp_path / "gather_ok.npy")[0]) == n_frames


def _generate_worker(rank, world, port, work):
    import scipy.io.wavfile

    from maua_stylegan2_amd import generate_audiovisual as gav
    from maua_stylegan2_amd import render, seeding
    from maua_stylegan2_amd.audioreactive.examples import default as plugin

    torch.set_grad_enabled(False)
    os.chdir(work)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    render.shutil.which = lambda name: None  # raw rgb24 sink
    sr = 22050
    if rank == 0:
        torch.save({"g_ema": seeding.seeded_state_dict(512, seed=1)}, "seeded512.pt")
        scipy.io.wavfile.write("track.wav", sr, (seeding.synthetic_audio(2.0, sr) * 32767).astype(np.int16))
        np.save("lat.npy", seeding.seeded_latents(12, 16, seed=3).numpy())
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    kw = dict(ckpt="seeded512.pt", audio_file="track.wav", initialize=plugin.initialize, get_latents=plugin.get_latents,
              get_noise=plugin.get_noise, latent_file="lat.npy", G_res=512, out_size=512, fps=12, batch=4)
    def seed_everything():  # the default plugin draws its reactive noise with torch.randn: both jobs start from the same generators
        import random

        random.seed(123), np.random.seed(123), torch.manual_seed(123), torch.cuda.manual_seed_all(123)

    try:
        dist.barrier()
        seed_everything()
        gav.generate(output_file=os.path.join(work, "two.mp4"), **kw)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    if rank == 0:
        seed_everything()
        gav.generate(output_file=os.path.join(work, "one.mp4"), **kw)  # the same job on one rank
        two = np.fromfile(os.path.join(work, "two.mp4.rgb24"), dtype=np.uint8)
        one = np.fromfile(os.path.join(work, "one.mp4.rgb24"), dtype=np.uint8)
        assert two.size == one.size == 24 * 512 * 512 * 3
        diff = np.abs(two.astype(np.int16) - one.astype(np.int16))
        np.save(os.path.join(work, "generate_ok.npy"), np.array([int(diff.max()), float((diff > 0).mean())]))


def test_two_rank_generate_on_the_device(gpu, tmp_path):
    """generate() itself under two ranks ON THE DEVICE (gloo over the shared GPU): rank 0 reads the checkpoint and runs the audio front end +
    callbacks, the weights travel as one flat broadcast, the per-frame inputs are scattered, both ranks render their block through captured
    lanes and the frames are gathered to rank 0's sink.  With the random generators seeded alike, the delivered file equals the single-rank
    job's (<= 1 grey level: a block boundary may fall inside a batch, and batch 4 vs a ragged batch changes the split-K depth of the small
    layers by an ulp)."""
    mp.spawn(_generate_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    worst, share = np.load(tmp_path / "generate_ok.npy")
    print(f"[two-rank generate on the device] max grey-level difference vs the single-rank job {int(worst)}, differing bytes {100 * share:.4f} %")
    assert worst <= 1 and share < 1e-3, (worst, share)
