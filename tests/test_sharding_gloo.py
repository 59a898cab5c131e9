"""CPU, world_size 2 over gloo: the frame-shard / weight-broadcast / ordered-gather logic of the multi-GPU path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from maua_stylegan2_amd import sharding


def test_shard_bounds_cover_every_frame_once():
    for n in [0, 1, 7, 8, 9, 225, 1799, 1800]:
        for world in [1, 2, 3, 4, 8]:
            seen = []
            for r in range(world):
                lo, hi = sharding.shard_bounds(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo <= sharding.max_shard(n, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(n))
    assert sharding.shard_bounds(1800, 3, 8) == (675, 900)  # config 4: 225 contiguous frames per GPU


def test_gather_frames_single_process_streams_in_chunks():
    """No process group: the shard itself is handed out frame by frame, fetched ``chunk`` frames at a time."""
    shard = torch.arange(7, dtype=torch.uint8).reshape(7, 1, 1, 1).repeat(1, 2, 3, 3)
    frames = sharding.gather_frames(shard, 5, chunk=2)
    assert not isinstance(frames, (list, tuple))  # an iterator: host memory does not scale with the video length
    got = list(frames)
    assert len(got) == 5 and [int(f[0, 0, 0]) for f in got] == [0, 1, 2, 3, 4]
    assert all(f.shape == (2, 3, 3) and f.dtype == torch.uint8 for f in got)
    assert list(sharding.gather_frames(shard, 0)) == []


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_frames, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # weights: only rank 0 has the "checkpoint"
        lin = torch.nn.Linear(4, 3)
        lin.register_buffer("kernel", torch.full((2, 2), float(rank)))
        if rank == 0:
            with torch.no_grad():
                lin.weight.fill_(1.25), lin.bias.fill_(-0.5)
        sharding.broadcast_module(lin)
        assert float(lin.weight.sum()) == 1.25 * 12 and float(lin.kernel.sum()) == 0.0
        tl = torch.full((1, 8), float(rank + 7))
        sharding.broadcast_tensor(tl)
        assert float(tl[0, 0]) == 7.0
        # frames: every rank "renders" its shard; the frame index is encoded in the pixels
        lo, hi = sharding.shard_bounds(n_frames, rank, world)
        shard = torch.zeros((sharding.max_shard(n_frames, world), 4, 5, 3), dtype=torch.uint8)
        for i in range(lo, hi):
            shard[i - lo] = i % 251
        frames = sharding.gather_frames(shard, n_frames, chunk=3)
        if rank == 0:
            frames = list(frames)
            assert len(frames) == n_frames
            for i, f in enumerate(frames):
                assert f.shape == (4, 5, 3) and int(f[0, 0, 0]) == i % 251 and int(f.max()) == int(f.min())
            np.save(os.path.join(out_dir, "ok.npy"), np.array([n_frames]))
        else:
            assert frames is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [9, 16])
def test_two_rank_broadcast_and_ordered_gather(tmp_path, n_frames):
    mp.spawn(_worker, args=(2, _free_port(), n_frames, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "ok.npy")[0]) == n_frames


class _FakeGenerator:
    """Just enough of the Generator surface for render.render's rank logic (device of the constant input)."""

    class _In:
        input = torch.zeros(1)

    input = _In()


def _render_worker(rank, world, port, n_frames, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from maua_stylegan2_amd import render

        def fake_synthesize(generator, latents, noise, batch_size, truncation=1.0, bends=(), rewrites=None,
                            randomize_noise=False, use_graph=True, frame_range=None):
            lo, hi = frame_range
            for n in range(lo, hi, batch_size):
                m = min(n + batch_size, hi)
                u8 = torch.zeros((m - n, 512, 512, 3), dtype=torch.uint8)
                for i in range(n, m):
                    u8[i - n] = (latents[i, 0, 0].item() % 251)  # frame content identifies the frame
                yield n, u8

        render.synthesize = fake_synthesize
        render.shutil.which = lambda name: None  # raw .rgb24 sink
        latents = torch.arange(n_frames, dtype=torch.float32).reshape(n_frames, 1, 1).repeat(1, 2, 4)
        out = os.path.join(out_dir, "sharded.mp4")
        written = render.render(_FakeGenerator(), latents, [None], 0, n_frames / 30, 3, 512, out)
        if rank == 0:
            assert written == n_frames
            raw = np.fromfile(out + ".rgb24", dtype=np.uint8).reshape(n_frames, 512, 512, 3)
            assert [int(f[0, 0, 0]) for f in raw] == [i % 251 for i in range(n_frames)]
            np.save(os.path.join(out_dir, "render_ok.npy"), np.array([written]))
        else:
            assert written == 0 and not os.path.exists(out + ".rgb24.rank1")
    finally:
        dist.destroy_process_group()


def test_render_two_ranks_ordered_sink(tmp_path):
    """render.render with torch.distributed initialised: each rank renders its contiguous shard, rank 0 writes every
    frame in order (generator and HIP epilogue replaced by a CPU stand-in: this exercises the rank logic only)."""
    n_frames = 11
    mp.spawn(_render_worker, args=(2, _free_port(), n_frames, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "render_ok.npy")[0]) == n_frames
