#!/usr/bin/env python3
"""What one FrameStream round costs on the `nccl` (= RCCL) backend, one rank (gpurun box):

    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/gather_probe.py

Per variant: BURSTS bursts of 60 rounds of 8 x 1024 x 1024 x 3 uint8 (25 MB); per burst the host time per round (the Python thread that
also launches the graph replays) and the wall time per round up to a device-wide synchronize."""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
BURSTS = int(os.environ.get("BURSTS", 6))


def main():
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    from maua_stylegan2_amd import sharding

    B, size, rounds = 8, 1024, 60
    u8 = torch.randint(0, 255, (B, size, size, 3), dtype=torch.uint8, device=dev)
    store = torch.empty((rounds, B, size, size, 3), dtype=torch.uint8, device=dev)
    side = torch.cuda.Stream(dev)
    out = {}

    def bursts(name, fn, after=None):
        rows = []
        for _ in range(BURSTS):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(rounds):
                fn(k)
            host = (time.perf_counter() - t0) / rounds
            torch.cuda.synchronize()
            rows.append((round(host * 1e6, 1), round((time.perf_counter() - t0) / rounds * 1e6, 1)))
            if after:
                after()
        out[name] = {"host_us_per_round": [r[0] for r in rows], "wall_us_per_round": [r[1] for r in rows]}

    works = []

    def drop():
        for w in works:
            w.wait()
        works.clear()

    import gc
    t0 = time.perf_counter()
    n = gc.collect()
    out["gc.collect() of the whole heap (torch + package imported, process group up)"] = {"ms": round((time.perf_counter() - t0) * 1e3, 1), "unreachable": n,
                                                                                      "tracked_objects": len(gc.get_objects()), "thresholds": gc.get_threshold()}
    stalls = []
    gc.callbacks.append(lambda phase, info: stalls.append((phase, info["generation"], time.perf_counter())))
    fs = sharding.FrameStream(rounds * B, B, (size, size, 3), dev)
    bursts("FrameStream.push (+ wait_all, reset per burst)", lambda k: fs.push(k, u8), after=lambda: (fs.wait_all(), fs.reset()))
    bursts("copy on the current stream", lambda k: store[k].copy_(u8))

    def side_copy(k):
        with torch.cuda.stream(side):
            store[k].copy_(u8)
    bursts("copy on a side stream", side_copy)
    bursts("dist.gather(async_op=True)", lambda k: works.append(dist.gather(u8, [store[k]], dst=0, async_op=True)), after=drop)
    bursts("dist.gather, works never waited for", lambda k: works.append(dist.gather(u8, [store[k]], dst=0, async_op=True)), after=works.clear)
    bursts("dist.gather(async_op=False)", lambda k: dist.gather(u8, [store[k]], dst=0))
    bursts("dist.all_gather_into_tensor(async_op=True)", lambda k: works.append(dist.all_gather_into_tensor(store[k], u8, async_op=True)), after=drop)
    gen2 = [(b[2] - a[2]) * 1e3 for a, b in zip(stalls[::2], stalls[1::2]) if a[1] == 2]
    out["generation-2 collections during the bursts"] = {"count": len(gen2), "ms_each": [round(v, 1) for v in gen2]}
    for name, rec in out.items():
        print(json.dumps({name: rec}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
