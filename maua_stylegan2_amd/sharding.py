"""Frame sharding across the GPUs of one node — replaces ``th.nn.DataParallel(generator)``
(/root/reference/generate_audiovisual.py:54-55), which re-broadcasts ~133 MB of weights and gathers fp32 images on
every forward.

Frames are independent units: one process per GPU (torchrun), rank r renders the contiguous block
``shard_bounds(n_frames, r, world)``, with NO collective on the data path.  Collectives (RCCL over xGMI through
``torch.distributed``; ``gloo`` in the CPU tests) appear exactly twice per render:
  * ``broadcast_module``: weights + buffers from rank 0, once (only rank 0 needs the checkpoint);
  * ``gather_frames``: finished uint8 NHWC frames to rank 0 (3 MiB/frame at 1024^2 — 4x less than the fp32 images
    DataParallel gathers), each peer over its own direct xGMI link.
``truncation_latent`` is random in the reference (models/stylegan2.py:539-540); ``broadcast_tensor`` keeps it identical
on every rank.
"""
import torch as th
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def max_shard(n_frames, world):
    return (n_frames + world - 1) // world


def shard_bounds(n_frames, rank, world):
    """Contiguous [lo, hi) block of rank ``rank``: ceil(n/world) frames each, the last ranks may get fewer or none."""
    per = max_shard(n_frames, world)
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def broadcast_module(module, src=0):
    """Make every rank's parameters and buffers equal rank ``src``'s (one broadcast per tensor, done once)."""
    rank, world = rank_world()
    if world == 1:
        return module
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src)
    return module


def broadcast_tensor(t, src=0):
    rank, world = rank_world()
    if world > 1:
        dist.broadcast(t, src)
    return t


def gather_frames(shard, n_frames, dst=0, chunk=64):
    """``shard``: uint8 [max_shard, H, W, 3] on every rank (rows beyond the rank's block are ignored).
    Returns on ``dst`` an iterator over the n_frames frames in order (CPU uint8 tensors), elsewhere None.  The gathered
    blocks stay on ``dst``'s device (3 MiB per 1024^2 frame) and are brought to the host ``chunk`` frames at a time while
    the iterator is consumed, so host memory does not grow with the length of the video."""
    rank, world = rank_world()
    if world == 1:
        blocks = [(shard, min(n_frames, shard.shape[0]))]
    else:
        bufs = [th.empty_like(shard) for _ in range(world)] if rank == dst else None
        dist.gather(shard, bufs, dst=dst)
        if rank != dst:
            return None
        blocks = []
        for r in range(world):
            lo, hi = shard_bounds(n_frames, r, world)
            blocks.append((bufs[r], hi - lo))

    def frames():
        for block, count in blocks:
            for c in range(0, count, chunk):
                host = block[c: min(c + chunk, count)].cpu()
                for i in range(host.shape[0]):
                    yield host[i]

    return frames()
