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


def _stream_worker(rank, world, port, n_frames, batch, out_dir):
    import time

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # ---- scatter_frames: rank 0 owns the per-frame inputs, every rank gets exactly its block; None stays None
        full = torch.arange(n_frames * 6, dtype=torch.float32).reshape(n_frames, 3, 2) if rank == 0 else None
        lo, hi = sharding.shard_bounds(n_frames, rank, world)
        mine = sharding.scatter_frames(full, n_frames)
        want = torch.arange(n_frames * 6, dtype=torch.float32).reshape(n_frames, 3, 2)[lo:hi]
        assert mine.shape == want.shape and torch.equal(mine, want)
        assert sharding.scatter_frames(None, n_frames) is None
        assert sharding.broadcast_object({"a": rank} if rank == 0 else None) == {"a": 0}

        # ---- generate()'s hand-over: rank 0 ran the callbacks, the others hold nothing before the scatter
        from maua_stylegan2_amd import generate_audiovisual as gav

        lat_full = torch.arange(n_frames * 8, dtype=torch.float32).reshape(n_frames, 2, 4)
        nz_full = torch.arange(n_frames * 4, dtype=torch.float32).reshape(n_frames, 1, 2, 2)
        tr_full = torch.linspace(0.5, 1.0, n_frames)
        mod_full = torch.arange(n_frames * 2, dtype=torch.float32).reshape(n_frames, 2)
        bends = [{"layer": 3, "modulation": mod_full.clone() if rank == 0 else torch.zeros_like(mod_full), "transform": None},
                 {"layer": 0, "transform": None}]
        rewrites = {"conv1.conv.weight": [None, mod_full[:, 0].clone() if rank == 0 else torch.zeros(n_frames)]}
        if rank == 0:
            lat_s, nz_s, tr_s = gav._scatter_from_rank0(lat_full, [nz_full, None], tr_full, bends, rewrites, n_frames)
        else:
            lat_s, nz_s, tr_s = gav._scatter_from_rank0(None, [], 1.0, bends, rewrites, n_frames)
        assert torch.equal(lat_s, lat_full[lo:hi]) and torch.equal(nz_s[0], nz_full[lo:hi]) and nz_s[1] is None
        assert torch.allclose(tr_s, tr_full[lo:hi]) and torch.equal(bends[0]["modulation"], mod_full[lo:hi])
        assert torch.equal(rewrites["conv1.conv.weight"][1], mod_full[lo:hi, 0])
        lat_s, nz_s, tr_s = gav._scatter_from_rank0(lat_full if rank == 0 else None, [None] if rank == 0 else [], 0.7, [], {},
                                                    n_frames)
        assert tr_s == 0.7 and nz_s == [None]

        # ---- FrameStream: batch-rounds travel while the shards are still being produced
        stream = sharding.FrameStream(n_frames, batch, (4, 5, 3), torch.device("cpu"))
        assert stream.rounds == -(-sharding.max_shard(n_frames, world) // batch)
        got, first_peer_frame_at = [], None
        k = 0
        for first in range(lo, hi, batch):
            if rank == 1:
                time.sleep(0.25)  # rank 1 is the slow producer: its block takes >= rounds * 0.25 s
            count = min(batch, hi - first)
            u8 = torch.zeros((count, 4, 5, 3), dtype=torch.uint8)
            for i in range(count):
                u8[i] = (first + i) % 251
            stream.push(k, u8)
            k += 1
            if rank == 0:
                got += list(stream.drain(block=False))
        if rank == 0:
            # rank 0's own block is already on its way to the sink although no later round of the slow peer exists yet
            assert [i for i, _ in got] == list(range(lo, hi))
        stream.finish()
        finished_at = time.monotonic()
        if rank == 0:
            for item in stream.drain(block=True):
                if first_peer_frame_at is None and item[0] >= sharding.shard_bounds(n_frames, 1, world)[0]:
                    first_peer_frame_at = time.monotonic()  # = the moment the peer's FIRST round was on rank 0's host
                got.append(item)
            assert [i for i, _ in got] == list(range(n_frames))
            for i, f in got:
                assert f.shape == (4, 5, 3) and int(f.min()) == int(f.max()) == i % 251
        else:
            stream.wait_all()
        # rank 0 had rank 1's FIRST round on its host before rank 1 finished producing its block
        stamps = [torch.zeros(2, dtype=torch.float64) for _ in range(world)] if rank == 0 else None
        dist.gather(torch.tensor([finished_at, first_peer_frame_at or 0.0], dtype=torch.float64), stamps, dst=0)
        if rank == 0:
            if stream.rounds >= 3:
                assert first_peer_frame_at is not None and first_peer_frame_at < float(stamps[1][0]), (
                    first_peer_frame_at, float(stamps[1][0]))
            np.save(os.path.join(out_dir, "stream_ok.npy"), np.array([n_frames]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames,batch", [(23, 3), (8, 4), (5, 8)])
def test_two_rank_scatter_and_streamed_ordered_frames(tmp_path, n_frames, batch):
    """world_size 2 over gloo: scatter_frames hands every rank only its block; FrameStream delivers the frames to rank 0 in
    global order, and (23 frames, batch 3: 4 rounds) rank 0 holds the slow rank's first round before that rank has finished
    its block — the transfer overlaps production instead of waiting for the end of the shard."""
    mp.spawn(_stream_worker, args=(2, _free_port(), n_frames, batch, str(tmp_path)), nprocs=2, join=True)
    assert int(np.load(tmp_path / "stream_ok.npy")[0]) == n_frames
