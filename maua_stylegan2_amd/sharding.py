"""Frame sharding across the GPUs of one node — replaces ``th.nn.DataParallel(generator)``
(/root/reference/generate_audiovisual.py:54-55), which re-broadcasts ~133 MB of weights and gathers fp32 images on
every forward.

Frames are independent units: one process per GPU (torchrun), rank r renders the contiguous block
``shard_bounds(n_frames, r, world)``.  Collectives (RCCL over xGMI through ``torch.distributed``; ``gloo`` in the CPU
tests):
  * ``broadcast_module``: weights + buffers from rank 0, once (only rank 0 reads the checkpoint);
  * ``scatter_frames``: rank 0 ran the audio front end and the plugin callbacks; every rank receives only ITS block of the
    per-frame inputs (latents, reactive noise maps, truncation) — 1/world of what a broadcast would move;
  * ``FrameStream``: finished uint8 NHWC frames (3 MiB/frame at 1024^2 — 4x less than the fp32 images DataParallel gathers)
    travel to rank 0 one batch-round at a time, as asynchronous gathers that overlap the next batches' compute; each peer
    writes over its own direct xGMI link into its slot of rank 0's HBM store, and rank 0's ordered sink consumes frames
    while the shards are still rendering.  No all-reduce anywhere on the path.
``truncation_latent`` is random in the reference (models/stylegan2.py:539-540); generate() draws it on rank 0 and
``broadcast_tensor`` makes it identical on every rank.
"""
import queue
import time

import torch as th
import torch.distributed as dist


def rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def grouped():
    """True when a process group exists — whatever its size.  Every helper below takes its REAL collective branch then, also in a
    group of one rank: `torchrun --nproc-per-node 1` on the nccl backend issues the same dist.broadcast / dist.scatter / asynchronous
    dist.gather sequence as an 8-rank job (round 4 returned early for world == 1, so a one-rank RCCL run proved nothing about the
    collectives).  Without a process group there is nothing to call and the local short cuts apply."""
    return dist.is_available() and dist.is_initialized()


def max_shard(n_frames, world):
    return (n_frames + world - 1) // world


def shard_bounds(n_frames, rank, world):
    """Contiguous [lo, hi) block of rank ``rank``: ceil(n/world) frames each, the last ranks may get fewer or none."""
    per = max_shard(n_frames, world)
    lo = min(rank * per, n_frames)
    return lo, min(lo + per, n_frames)


def broadcast_module(module, src=0):
    """Make every rank's parameters and buffers equal rank ``src``'s: ONE broadcast per dtype of a flat staging buffer (the
    1024^2 generator is 171 tensors / 133 MB; per-tensor broadcasts were 171 collectives, most of them latency-bound)."""
    rank, world = rank_world()
    if not grouped():
        return module
    by_dtype = {}
    # (the tensors themselves, not .data: copy_ below then advances their version counters, which is what the packed-weight cache,
    # the style tables and the graph lanes are keyed on — a second broadcast into a generator that has already packed / captured
    # must invalidate them)
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    for dtype in sorted(by_dtype, key=str):
        tensors = by_dtype[dtype]
        flat = th.empty(sum(t.numel() for t in tensors), dtype=dtype, device=tensors[0].device)
        with th.no_grad():
            if rank == src:
                th.cat([t.detach().reshape(-1) for t in tensors], out=flat)
            dist.broadcast(flat, src)
            if rank != src:
                off = 0
                for t in tensors:
                    t.copy_(flat[off: off + t.numel()].view_as(t))
                    off += t.numel()
    return module


def broadcast_tensor(t, src=0):
    if grouped():
        dist.broadcast(t, src)
    return t


def gather_frames(shard, n_frames, dst=0, chunk=64):
    """``shard``: uint8 [max_shard, H, W, 3] on every rank (rows beyond the rank's block are ignored).
    Returns on ``dst`` an iterator over the n_frames frames in order (CPU uint8 tensors), elsewhere None.  The gathered
    blocks stay on ``dst``'s device (3 MiB per 1024^2 frame) and are brought to the host ``chunk`` frames at a time while
    the iterator is consumed, so host memory does not grow with the length of the video."""
    rank, world = rank_world()
    if not grouped():
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


def broadcast_object(obj, src=0):
    """Small picklable metadata (shapes, which noise scales are None) from ``src`` to every rank."""
    rank, world = rank_world()
    if not grouped():
        return obj
    box = [obj if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def scatter_frames(t, n_frames, src=0, device=None):
    """``t`` [n_frames, ...] exists on ``src`` (others pass None): every rank gets its contiguous block
    ``t[shard_bounds(n_frames, rank, world)]`` on ``device`` and nothing else.  ``None`` on ``src`` stays ``None`` everywhere
    (a noise scale served by the checkpoint's buffer)."""
    rank, world = rank_world()
    if not grouped():
        return t if t is None or device is None else t.to(device)
    meta = broadcast_object(None if t is None else (tuple(t.shape[1:]), str(t.dtype).replace("torch.", "")), src)
    if meta is None:
        return None
    shape, dtype = meta[0], getattr(th, meta[1])
    per = max_shard(n_frames, world)
    if device is None:
        device = t.device if t is not None else th.device("cpu")
    mine = th.empty((per,) + shape, dtype=dtype, device=device)
    chunks = None
    if rank == src:
        if t.shape[0] != n_frames:
            raise RuntimeError(f"per-frame tensor has {t.shape[0]} entries for {n_frames} frames")
        # scatter wants equally sized chunks: rank r's is the `per` frames starting at its block — a VIEW of the sequence (it may run
        # into the next rank's frames, which the receiver drops); only a block that would run past the end gets a zero-padded copy
        # of its own (round 3 built a padded [world * per] copy of every per-frame tensor: ~1.3 GB of reactive noise twice on rank 0)
        t = t.to(device).contiguous()
        chunks = []
        for r in range(world):
            lo_r, hi_r = shard_bounds(n_frames, r, world)
            if lo_r + per <= n_frames:
                chunks.append(t[lo_r: lo_r + per])
            else:
                tail = th.zeros((per,) + shape, dtype=dtype, device=device)
                tail[: hi_r - lo_r].copy_(t[lo_r: hi_r])
                chunks.append(tail)
    dist.scatter(mine, chunks, src=src)
    lo, hi = shard_bounds(n_frames, rank, world)
    return mine[: hi - lo]


class FrameStream:
    """Ordered multi-rank frame sink transport: every rank calls ``push(k, u8)`` once per batch-round k = 0, 1, ... with the
    frames it produced in that round ([b <= batch, H, W, 3] uint8, or None when its block is exhausted); each push starts
    one asynchronous ``gather`` of that round straight into ``dst``'s store [world, rounds * batch, H, W, 3] (device memory:
    675 MiB per peer at 1800 frames / 8 GPUs), so transfers overlap the following rounds' compute.  On ``dst``,
    ``drain()`` yields the frames whose round has landed, in global frame order (rank-major: the blocks are contiguous),
    fetched to the host one round at a time — reference render.py:94-113 semantics (an ordered writer that consumes while
    the generator runs) without its per-frame D2H."""

    def __init__(self, n_frames, batch_size, frame_shape, device, dst=0, ring_slots=6):
        self.rank, self.world = rank_world()
        self.grouped = grouped()  # a group of ONE rank still gathers for real (into its own store)
        self.n_frames, self.batch, self.dst = int(n_frames), int(batch_size), dst
        self.per = max_shard(n_frames, self.world)
        self.rounds = (self.per + self.batch - 1) // self.batch
        self.lo, self.hi = shard_bounds(n_frames, self.rank, self.world)
        self.shape = tuple(frame_shape)
        slots = max(self.rounds, 1) * self.batch
        # this rank's frames, kept until the gather has read them (the producer recycles its batch buffers).  Deliberately NOT
        # zero-filled here: the constructor runs on whatever stream produced the first batch, the pushes of the other graph lanes
        # run on theirs, and a fill kernel queued behind the first lane's replay could land on top of a faster lane's frames
        # (seen once as a corrupted shard in the GPU suite).  Every slot is written completely by its own push, on its own stream.
        self.mine = th.empty((slots,) + self.shape, dtype=th.uint8, device=device)
        self.store = th.empty((self.world, slots) + self.shape, dtype=th.uint8, device=device) if self.rank == dst else None
        self.works = []
        self.pushed = 0
        self.bytes_gathered = 0  # bytes this rank has handed to / received from dist.gather (bench.py's `rccl` block)
        self._cursor = (0, 0)  # (rank, round) of the next frames to hand out on dst
        # dst's way to the host: a ring of pinned staging buffers filled by asynchronous copies on a copy stream, so that
        # fetching round k overlaps the Python thread launching the next replays (as render() does on one GPU); on a CPU
        # "device" (gloo tests) the rounds are handed out in place
        self._on_gpu = th.device(device).type == "cuda"
        self._ring, self._copy_stream, self._inflight = [], None, []
        self._pushed_events = []
        self._free = queue.Queue()  # ring slots a consumer has released (the sink may run on its own thread: render.SinkWorker)
        if self.rank == dst and self._on_gpu:
            self._copy_stream = th.cuda.Stream(device)
            self._ring = [th.empty((self.batch,) + self.shape, dtype=th.uint8).pin_memory() for _ in range(ring_slots)]
            for i in range(ring_slots):
                self._free.put(i)

    def push(self, k, u8):
        if k != self.pushed:
            raise RuntimeError(f"FrameStream.push: round {k} out of order (expected {self.pushed})")
        slot = self.mine[k * self.batch: (k + 1) * self.batch]
        filled = 0
        if u8 is not None:
            filled = u8.shape[0]
            slot[:filled].copy_(u8)
        if filled < self.batch:
            slot[filled:].zero_()  # ragged / empty round: defined bytes travel (dst never hands them out)
        if self._copy_stream is not None:  # dst: the staging copy of this round must run behind the copy above
            ev = th.cuda.Event()
            ev.record(th.cuda.current_stream(self.mine.device))
            self._pushed_events.append(ev)
        if self.grouped:
            into = [self.store[p, k * self.batch: (k + 1) * self.batch] for p in range(self.world)] if self.rank == self.dst else None
            self.works.append(dist.gather(slot, into, dst=self.dst, async_op=True))
            self.bytes_gathered += slot.numel() * (self.world if self.rank == self.dst else 1)
        elif self.store is not None:
            self.store[0, k * self.batch: (k + 1) * self.batch].copy_(slot)
            self.works.append(None)
        self.pushed += 1

    def finish(self):
        """Keep taking part in the remaining rounds (ranks whose block was short or empty) — every rank must issue the
        same sequence of collectives."""
        while self.pushed < self.rounds:
            self.push(self.pushed, None)

    def wait_all(self):
        for w in self.works:
            if w is not None:
                w.wait()

    def reset(self):
        """Start over at round 0 in the same buffers (every collective of the previous pass is waited for first).  For callers that
        stream more rounds than they want to hold — bench.py's gathered region walks two streams of a fixed number of rounds
        alternately, so that rank 0's store stays bounded whatever --steps is."""
        for w in self.works:  # completion as seen from the HOST (Work.wait() would only order the current stream behind the collective,
            while w is not None and not w.is_completed():  # and the next pass's pushes run on other streams)
                time.sleep(0.0002)
        self.works, self.pushed, self._cursor = [], 0, (0, 0)
        self._pushed_events, self._inflight = [], []

    def _landed(self, k, block):
        if k >= len(self.works):
            return False
        w = self.works[k]
        if w is None:
            return True
        if block:
            w.wait()
            return True
        return w.is_completed()

    def _next_round(self, block):
        """dst: (first_frame, count, device tensor, event the data is ready behind) of the next round in global frame order
        (rank-major: the blocks are contiguous) if it is available, else None; advances the cursor."""
        p, k = self._cursor
        while p < self.world:
            lo, hi = shard_bounds(self.n_frames, p, self.world)
            first = lo + k * self.batch
            if first >= hi:  # this rank's block is done (or empty): next rank
                p, k = p + 1, 0
                self._cursor = (p, k)
                continue
            count = min(self.batch, hi - first)
            ready = None
            if p == self.rank:  # dst's own frames never wait for a transfer: they are local as soon as they are pushed
                if k >= self.pushed:
                    return None
                src = self.mine[k * self.batch: k * self.batch + count]
                if self._pushed_events:
                    ready = self._pushed_events[k]
            else:
                if not self._landed(k, block):
                    return None
                src = self.store[p, k * self.batch: k * self.batch + count]
                if self._on_gpu:  # (a completed / waited-for work orders the CURRENT stream behind the transfer)
                    ready = th.cuda.Event()
                    ready.record(th.cuda.current_stream(self.mine.device))
            self._cursor = (p, k + 1)
            return first, count, src, ready
        return None

    def drain_rounds(self, block=False):
        """dst only: yield (first_frame_index, count, uint8 host tensor [count, H, W, 3], release) for every ROUND that is next in
        global frame order and has arrived.  The host tensor is a slot of the pinned ring (the round itself on a CPU "device"); the
        consumer calls ``release()`` when it is done with it — possibly from another thread (render.SinkWorker), which is what keeps
        a slow sink off the thread that launches the graphs.  ``block=False`` never waits: not for a transfer, not for a copy and
        not for a ring slot (the caller comes back after its next replay); ``block=True`` waits for all three."""
        if self.rank != self.dst:
            return
        while True:
            while self._on_gpu or not self._inflight:  # start copies while there are rounds and ring slots (CPU: one round at a time)
                slot = None
                if self._on_gpu:
                    try:
                        slot = self._free.get(block=block and not self._inflight)
                    except queue.Empty:
                        break
                nxt = self._next_round(block and not self._inflight)  # (never wait for a later round while an earlier one can be handed out)
                if nxt is None:
                    if slot is not None:
                        self._free.put(slot)
                    break
                first, count, src, ready = nxt
                if not self._on_gpu:
                    self._inflight.append((first, count, src, None, None))
                    continue
                host = self._ring[slot][:count]
                with th.cuda.stream(self._copy_stream):
                    if ready is not None:
                        self._copy_stream.wait_event(ready)
                    host.copy_(src, non_blocking=True)
                    done = th.cuda.Event()
                    done.record(self._copy_stream)
                self._inflight.append((first, count, host, done, slot))
            if not self._inflight:
                return
            first, count, host, done, slot = self._inflight[0]
            if done is not None:
                if not block and not done.query():
                    return  # still on its way: the caller comes back after its next replay
                done.synchronize()
            self._inflight.pop(0)
            yield first, count, host, (lambda s=slot: self._free.put(s)) if slot is not None else (lambda: None)

    def drain(self, block=False):
        """dst only: yield (global_frame_index, uint8 CPU tensor [H, W, 3]) for every frame that is next in order and whose
        round has arrived (see drain_rounds); the round's ring slot is released when the consumer asks for the frame behind it."""
        for first, count, host, release in self.drain_rounds(block):
            try:
                for i in range(count):
                    yield first + i, host[i]
            finally:
                release()


def _tracker_pid():
    """pid of the multiprocessing resource tracker this process reports to (0 if it cannot be told)."""
    try:
        from multiprocessing import resource_tracker

        resource_tracker.ensure_running()
        return int(getattr(resource_tracker._resource_tracker, "_pid", 0) or 0)
    except Exception:  # noqa: BLE001
        return 0


class HostFrameStore:
    """Alternative frame transport for a multi-GPU job on ONE host (``MAUA_FRAME_TRANSPORT=host`` / ``render(..., transport="host")``):
    every rank copies its batch-rounds to the host over ITS OWN PCIe link, into a POSIX shared-memory segment it owns (pinned in
    place with hipHostRegister, so the copy is an asynchronous DMA), and publishes a round counter in the segment's header; rank 0's
    sink thread maps the peers' segments and reads the frames in global order.  Nothing travels over xGMI and nothing funnels
    through rank 0's single PCIe link — with the default transport (FrameStream: RCCL gather into rank 0's HBM, then rank 0's D2H)
    8 x 1200 frames/s of 1024^2 frames are 30 GB/s through one link and one process.  The default stays the RCCL gather that
    BASELINE.json's north_star names; this path exists for jobs whose encoder keeps up with more than one link's worth of frames.
    UNMEASURED on more than one GPU (no multi-GPU box in this environment); covered by a world-size-2 gloo test.

    Segment of rank r: [int64 published_rounds, int64 reserved] + rounds x batch x frame bytes (the rank's whole block: a peer never
    stalls on a slow sink, as with FrameStream's HBM store)."""

    HEADER = 64

    def __init__(self, n_frames, batch_size, frame_shape, device, token):
        import threading
        from multiprocessing import shared_memory

        import numpy as np

        self._np, self._shm_mod = np, shared_memory
        self.rank, self.world = rank_world()
        self.n_frames, self.batch = int(n_frames), int(batch_size)
        self.shape = tuple(frame_shape)
        self.per = max_shard(n_frames, self.world)
        self.rounds = (self.per + self.batch - 1) // self.batch
        self.lo, self.hi = shard_bounds(n_frames, self.rank, self.world)
        self.token = str(token)
        self.round_bytes = self.batch * int(np.prod(self.shape))
        size = self.HEADER + max(self.rounds, 1) * self.round_bytes
        # tmpfs reserves nothing at ftruncate: segments larger than what /dev/shm can still hold die with SIGBUS on first touch, not
        # with a Python error (64 MB is the container default).  Every rank of the node sees the same free space, so the check is on
        # the node's SUM (all ranks of a job have segments of one size), the pages are then really reserved (posix_fallocate), and the
        # outcome is collective: one rank that cannot have its segment makes EVERY rank raise — none is left waiting in the barrier below.
        import os

        local = int(os.environ.get("LOCAL_WORLD_SIZE", self.world)) if grouped() else 1
        problem = None
        try:
            vfs = os.statvfs("/dev/shm")
            free_bytes = vfs.f_bavail * vfs.f_frsize
        except OSError:
            free_bytes = None
        if free_bytes is not None and local * size > free_bytes:
            problem = (f"MAUA_FRAME_TRANSPORT=host: the {local} rank(s) of this node need {local} x {size / 2**20:.0f} MiB of shared memory for "
                       f"their frames ({max(self.rounds, 1) * self.batch} per rank) but /dev/shm has {free_bytes / 2**20:.0f} MiB free — enlarge "
                       f"/dev/shm (docker --shm-size) or use the default gather transport")
        self._mine = None
        if problem is None:
            try:
                self._mine = shared_memory.SharedMemory(name=self._name(self.rank), create=True, size=size)
                fd = getattr(self._mine, "_fd", -1)
                if fd is not None and fd >= 0:
                    os.posix_fallocate(fd, 0, size)  # reserve the pages now: ENOSPC here instead of SIGBUS in the frame loop
            except OSError as exc:
                problem = f"MAUA_FRAME_TRANSPORT=host: rank {self.rank} could not reserve its {size / 2**20:.0f} MiB shared-memory segment: {exc}"
        if grouped():
            flag = th.tensor([0 if problem is None else 1], dtype=th.int32, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()) and problem is None:
                problem = "MAUA_FRAME_TRANSPORT=host: another rank could not reserve its shared-memory segment (see its error)"
        if problem is not None:
            if self._mine is not None:
                self._mine.close()
                self._mine.unlink()
                self._mine = None
            raise RuntimeError(problem)
        self._header = np.ndarray((2,), dtype=np.int64, buffer=self._mine.buf)
        self._header[:] = 0
        self._header[1] = _tracker_pid()  # (the owner's resource tracker: see _peer)
        self._frames = th.from_numpy(np.ndarray((max(self.rounds, 1), self.batch) + self.shape, dtype=np.uint8, buffer=self._mine.buf,
                                                offset=self.HEADER))
        self._on_gpu = th.device(device).type == "cuda"
        self._registered = False
        self._copy_stream = None
        if self._on_gpu:
            rc = th.cuda.cudart().cudaHostRegister(self._frames.data_ptr(), self._frames.numel(), 0)
            self._registered = int(rc) == 0  # (an unpinned segment still works: the copy is then staged by the runtime)
            self._copy_stream = th.cuda.Stream(device)
        self.pushed = 0
        self._events = []
        self._dev_ring, self._dev_done = [None] * 3, [None] * 3  # device-side slots between the producer's frame buffer and PCIe
        self._published = 0
        self._lock = threading.Lock()
        self._peers = {}
        if grouped():
            dist.barrier()  # every segment exists before anybody attaches

    def _name(self, rank):
        return f"maua_{self.token}_r{rank}"

    def push(self, k, u8):
        """Round k of this rank's block ([b <= batch, H, W, 3] uint8 on the device, or None): asynchronous copy into the segment."""
        if k != self.pushed:
            raise RuntimeError(f"HostFrameStore.push: round {k} out of order (expected {self.pushed})")
        if u8 is not None:
            dst = self._frames[k, : u8.shape[0]]
            if self._on_gpu:
                # the round leaves the producer's frame buffer inside HBM first (a device-side ring slot, on the producer's stream) and
                # crosses PCIe from there: the producer never waits for a host copy before its next replay (render.render_shard does the
                # same in its single-GPU loop); a slot is re-used once the host copy that read it three rounds ago is done
                producer = th.cuda.current_stream(u8.device)
                slot = k % len(self._dev_ring)
                if self._dev_ring[slot] is None or self._dev_ring[slot].shape[1:] != u8.shape[1:] or self._dev_ring[slot].shape[0] < u8.shape[0]:
                    self._dev_ring[slot] = th.empty((max(self.batch, u8.shape[0]),) + tuple(u8.shape[1:]), dtype=th.uint8, device=u8.device)
                if self._dev_done[slot] is not None:
                    producer.wait_event(self._dev_done[slot])
                held = self._dev_ring[slot][: u8.shape[0]]
                held.copy_(u8, non_blocking=True)
                produced = th.cuda.Event()
                produced.record(producer)
                with th.cuda.stream(self._copy_stream):
                    self._copy_stream.wait_event(produced)
                    dst.copy_(held, non_blocking=True)
                    done = th.cuda.Event()
                    done.record(self._copy_stream)
                self._dev_done[slot] = done
                self._events.append(done)
            else:
                dst.copy_(u8)
                self._events.append(None)
        else:
            self._events.append(None)
        self.pushed += 1
        self.publish(block=False)

    def publish(self, block):
        """Advance the published-round counter over every round whose copy has completed (in order)."""
        with self._lock:
            while self._published < len(self._events):
                ev = self._events[self._published]
                if ev is not None:
                    if block:
                        ev.synchronize()
                    elif not ev.query():
                        break
                self._published += 1
            self._header[0] = self._published

    def finish(self):
        while self.pushed < self.rounds:
            self.push(self.pushed, None)
        self.publish(block=True)

    def _peer(self, p):
        if p == self.rank:
            return self._header, self._frames
        if p not in self._peers:
            np = self._np
            shm = self._shm_mod.SharedMemory(name=self._name(p))
            header = np.ndarray((2,), dtype=np.int64, buffer=shm.buf)
            # Python < 3.13 registers ATTACHED segments with this process's resource tracker as well; the owner unlinks them, so this
            # process's tracker must forget the name — but only if it is a tracker of its OWN (torchrun: one per rank).  Ranks started with
            # multiprocessing share ONE tracker with the owner: un-registering there removes the owner's registration (a KeyError
            # traceback at its unlink, and no clean-up if the owner crashes).  The owner left its tracker's pid in the header.
            if int(header[1]) != _tracker_pid():
                try:
                    from multiprocessing import resource_tracker

                    resource_tracker.unregister(shm._name, "shared_memory")
                except Exception:  # noqa: BLE001 - bookkeeping only
                    pass
            frames = th.from_numpy(np.ndarray((max(self.rounds, 1), self.batch) + self.shape, dtype=np.uint8, buffer=shm.buf,
                                              offset=self.HEADER))
            self._peers[p] = (shm, header, frames)
        return self._peers[p][1], self._peers[p][2]

    def rounds_in_order(self, poll_s=0.0005, own_progress=None, stop=None):
        """Rank 0: yield (first_frame_index, count, uint8 host tensor [count, H, W, 3]) for every round in global frame order
        (rank-major), waiting for each to be published.  ``own_progress()`` is called while waiting on rank 0's OWN rounds (its
        publisher runs on the thread that pushes).  ``stop()`` returning True ends the iteration (the launch loop failed: the reader
        thread must leave before the segments are closed)."""
        import time

        for p in range(self.world):
            lo, hi = shard_bounds(self.n_frames, p, self.world)
            header, frames = self._peer(p)
            k = 0
            while lo + k * self.batch < hi:
                while int(header[0]) <= k:
                    if stop is not None and stop():
                        return
                    if p == self.rank:
                        self.publish(block=False)
                        if own_progress is not None:
                            own_progress()
                    time.sleep(poll_s)
                first = lo + k * self.batch
                count = min(self.batch, hi - first)
                yield first, count, frames[k, :count]
                k += 1

    def close(self):
        """Collective: every rank keeps its segment until rank 0 has read it."""
        if grouped():
            dist.barrier()
        for shm, _, _ in self._peers.values():
            shm.close()
        self._peers = {}
        if self._registered:
            th.cuda.cudart().cudaHostUnregister(self._frames.data_ptr())
            self._registered = False
        self._header = self._frames = None
        try:
            self._mine.close()
            self._mine.unlink()
        except (FileNotFoundError, BufferError):
            pass
