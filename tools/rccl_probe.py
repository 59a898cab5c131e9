#!/usr/bin/env python3
"""uint8 gather, scatter_frames and FrameStream on the `nccl` (= RCCL) backend, one rank per GPU:

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/rccl_probe.py

On a 1-GPU box (gpurun) both ranks land on device 0 and RCCL refuses the communicator ("Duplicate GPU detected : rank 0 and rank 1
both on CUDA device", NCCL 2.26.6 / ROCm 7.0, tried in round 2): the multi-rank data path is covered by the gloo tests
(tests/test_sharding_gloo.py) and remains unmeasured on RCCL until a multi-GPU node runs this script."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    rank = int(os.environ["RANK"])
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    x = torch.full((4, 8, 8, 3), rank + 1, dtype=torch.uint8, device=dev)
    out = [torch.empty_like(x) for _ in range(dist.get_world_size())] if rank == 0 else None
    dist.gather(x, out, dst=0)
    torch.cuda.synchronize()
    if rank == 0:
        print("gather ok", [int(o.float().mean()) for o in out])
    from maua_stylegan2_amd import sharding

    t = torch.arange(10, dtype=torch.float32, device=dev).reshape(10, 1) if rank == 0 else None
    mine = sharding.scatter_frames(t, 10, src=0, device=dev)
    print(rank, "scatter ok", mine.flatten().tolist())
    fs = sharding.FrameStream(10, 2, (8, 8, 3), dev, dst=0)
    lo, hi = sharding.shard_bounds(10, rank, dist.get_world_size())
    for k in range((hi - lo + 1) // 2):
        n = min(2, hi - lo - 2 * k)
        fs.push(k, torch.stack([torch.full((8, 8, 3), lo + 2 * k + i, dtype=torch.uint8, device=dev) for i in range(n)]))
    fs.finish()
    got = [(i, int(f.float().mean())) for i, f in fs.drain(block=True)] if rank == 0 else []
    print(rank, "stream ok", got)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
