"""Frame rendering loop — mirror of /root/reference/render.py:14-192 ``render(...)`` for MI355X.

Same signature and batch semantics (slice latents / noise / bend modulation / truncation on dim 0, call the generator
with ``input_is_latent=True``), different machinery:

  reference                                             here
  --------------------------------------------------    ------------------------------------------------------------
  pin + H2D of latents and up to 17 noise maps / batch    inputs are uploaded ONCE and stay resident in HBM (288 GB)
  eager generator call, ~200 launches per batch           hipGraph replay per batch (eager only when bends / rewrites /
                                                          randomize_noise make the batch non-capturable, or a tail batch)
  clamp/scale/permute on device, per-FRAME .cpu()         uint8 NHWC written by the last layer's epilogue, one async D2H per BATCH into
  .numpy().astype(uint8), two Python threads + queues     pinned double buffers on a copy stream, ordered sink
  DataParallel replicate/scatter/gather per forward       one process per GPU, contiguous frame shards, RCCL gather of
                                                          uint8 frames to rank 0 (maua_stylegan2_amd/sharding.py)

Sinks: ffmpeg rawvideo pipe (same pixel format / codec arguments as render.py:58-91) when an ``ffmpeg`` binary exists,
otherwise raw rgb24 bytes to ``output_file`` (+ ".rgb24"), or a null sink for benchmarking (``output_file=None``).
"""
import gc
import os
import queue
import shutil
import subprocess
import threading

import numpy as np
import torch as th

from . import _lib, sharding

th.set_grad_enabled(False)


def device_of(generator):
    """Device a generator lives on (StyleGAN2 mirror: its constant / latent input; StyleGAN1's G_style: any parameter)."""
    inp = getattr(getattr(generator, "input", None), "input", None)
    return inp.device if inp is not None else next(generator.parameters()).device


def _output_dims(out_size):
    if out_size == 512:
        return 512, 512
    if out_size == 1024:
        return 1024, 1024
    if out_size == 1920:
        return 1920, 1080
    if out_size == 1080:
        return 1080, 1920
    raise Exception("The only output sizes currently supported are: 512, 1024, 1080, or 1920")


class FrameSink:
    """Ordered consumer of uint8 [H, W, 3] frames."""

    def __init__(self, output_file, width, height, framerate, audio_file=None, offset=0, duration=None,
                 ffmpeg_preset="slow"):
        self.w, self.h = width, height
        self.proc = None
        self.file = None
        self.count = 0
        if output_file is None:
            return
        if shutil.which("ffmpeg") is not None:
            cmd = ["ffmpeg", "-hide_banner", "-y", "-v", "warning", "-f", "rawvideo", "-pix_fmt", "rgb24", "-framerate",
                   f"{framerate}", "-s", f"{width}x{height}", "-i", "pipe:"]
            if audio_file is not None:
                cmd += ["-ss", f"{offset}", "-t", f"{duration}", "-guess_layout_max", "0", "-i", audio_file]
            cmd += ["-r", f"{framerate}", "-vcodec", "libx264", "-pix_fmt", "yuv420p", "-preset", ffmpeg_preset]
            if audio_file is not None:
                cmd += ["-b:a", "320K", "-ac", "2"]
            cmd += [output_file]
            self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE)
        else:
            path = output_file if output_file.endswith(".rgb24") else output_file + ".rgb24"
            audio = f" -ss {offset} -t {duration} -guess_layout_max 0 -i {audio_file}" if audio_file is not None else ""
            mux = " -b:a 320K -ac 2" if audio_file is not None else ""
            target = output_file[:-len(".rgb24")] if output_file.endswith(".rgb24") else output_file
            print(f"ffmpeg binary not found: writing raw rgb24 frames ({width}x{height}) to {path}\n"
                  f"  encode later with: ffmpeg -f rawvideo -pix_fmt rgb24 -framerate {framerate} -s {width}x{height} -i {path}{audio} "
                  f"-r {framerate} -vcodec libx264 -pix_fmt yuv420p -preset {ffmpeg_preset}{mux} {target}")
            self.file = open(path, "wb")

    def write(self, frame):
        """frame: numpy uint8 [H, W, 3]; wide 2048-px outputs are cropped + resized as render.py:98-105."""
        if frame.shape[1] == 2048 or frame.shape[0] == 2048:
            import PIL.Image

            if frame.shape[1] == 2048:
                frame = np.array(PIL.Image.fromarray(frame[:, 112:-112, :]).resize((1920, 1080), PIL.Image.BILINEAR))
            else:
                frame = np.array(PIL.Image.fromarray(frame[112:-112, :, :]).resize((1080, 1920), PIL.Image.BILINEAR))
        assert frame.shape[1] == self.w and frame.shape[0] == self.h, (
            f"generator's output image size does not match specified output size: \n"
            f"got: {frame.shape[1]}x{frame.shape[0]}\t\tshould be {self.w}x{self.h}")
        if self.proc is not None or self.file is not None:
            # the frame's own bytes go to the pipe / file (a view of the pinned staging buffer): no tobytes() copy of 3 MiB per frame
            data = memoryview(np.ascontiguousarray(frame)).cast("B")
            (self.proc.stdin if self.proc is not None else self.file).write(data)
        self.count += 1

    def close(self):
        if self.proc is not None:
            self.proc.stdin.close()
            self.proc.wait()
        if self.file is not None:
            self.file.close()


class SinkWorker:
    """Ordered frame delivery OFF the thread that launches the graphs — the role of the reference's two daemon threads and
    queues (render.py:30-44,94-113: `make_video` / `split_batches`).  The launch thread hands over whole batches
    (``submit(wait, frames, count, release)``: ``wait()`` blocks until the batch's device-to-host copy has landed, ``frames[i]`` are
    its uint8 [H, W, 3] frames in a pinned staging slot, ``release()`` returns the slot to its ring) and never calls
    ``sink.write`` itself: a slow encoder fills the ring and then throttles the producer through the ring's free list, it does
    not sit between two graph launches.  Batches are written in submission order.  An exception of the sink is kept and
    re-raised on the launch thread (next ``submit`` or ``close``)."""

    def __init__(self, sink):
        self.sink = sink
        self.error = None
        self._work = queue.Queue()
        self._thread = threading.Thread(target=self._run, name="maua-sink", daemon=True)
        self._thread.start()

    def _run(self):
        while True:
            item = self._work.get()
            if item is None:
                return
            wait, frames, count, release = item
            try:
                if self.error is None:
                    if wait is not None:
                        wait()
                    for i in range(count):
                        self.sink.write(frames[i])
            except BaseException as exc:  # noqa: BLE001 - handed to the launch thread
                self.error = exc
            finally:
                if release is not None:
                    release()

    def _check(self):
        if self.error is not None:
            exc, self.error = self.error, None
            raise exc

    def submit(self, wait, frames, count, release=None):
        self._check()
        self._work.put((wait, frames, count, release))

    def feed(self, wait, frames, count, release=None):
        """``submit`` for a HELPER thread (the host transport's reader): never raises and never takes the sink's error off the worker —
        the error stays where the launch thread's next ``submit`` / ``close`` finds it.  Returns False once the sink has failed (the
        batch is dropped and its slot released: the helper can stop feeding)."""
        if self.error is not None:
            if release is not None:
                release()
            return False
        self._work.put((wait, frames, count, release))
        return True

    def close(self):
        """Wait until everything submitted has been written (or dropped after an error), then re-raise a sink error."""
        if self._thread.is_alive():
            self._work.put(None)
            self._thread.join()
        self._check()


def frames_to_uint8(images, out=None):
    """[B,3,H,W] fp32 device tensor -> [B,H,W,3] uint8 device tensor (render.py:40-43) via the HIP epilogue."""
    lib = _lib.load()
    images = _lib.require_cuda(images, "images")
    b, c, h, w = images.shape
    if c != 3:
        raise RuntimeError("frames must have 3 channels")
    if out is None:
        out = th.empty((b, h, w, 3), dtype=th.uint8, device=images.device)
    with th.cuda.device(images.device):
        _lib.check(lib.maua_frames_to_u8(images.data_ptr(), out.data_ptr(), b, h, w, _lib.stream_ptr(images.device)),
                   "maua_frames_to_u8")
    return out


def crop_resize_for_delivery(u8, out_size, scratch):
    """The wide-output delivery of reference render.py:97-104 ON THE DEVICE: 2048-px frames of a 1920 / 1080 render are cropped
    (112 px off both ends of the long side) and resized to 1920x1080 / 1080x1920 by maua_crop_resize_u8 = PIL's bilinear resize —
    the reference (and round 2 here) does it per frame on the host with PIL, a few milliseconds each on rank 0's Python thread.
    Other frames pass through.  ``scratch``: dict holding the reusable output buffers (one per input buffer: the lanes' frames are
    in flight concurrently)."""
    b, h, w, _ = u8.shape
    if out_size == 1920 and w == 2048:
        x0, y0, cw, ch, ow, oh = 112, 0, w - 224, h, 1920, 1080
    elif out_size == 1080 and h == 2048:
        x0, y0, cw, ch, ow, oh = 0, 112, w, h - 224, 1080, 1920
    else:
        return u8
    if ow < cw or oh < ch:
        return u8  # not the up-scale the device kernel reproduces: FrameSink.write falls back to PIL on the host
    key = (u8.data_ptr(), b)
    out = scratch.get(key)
    if out is None:
        out = scratch[key] = th.empty((b, oh, ow, 3), dtype=th.uint8, device=u8.device)
    with th.cuda.device(u8.device):
        _lib.check(_lib.load().maua_crop_resize_u8(u8.data_ptr(), out.data_ptr(), b, h, w, x0, y0, cw, ch, ow, oh,
                                                   _lib.stream_ptr(u8.device)), "maua_crop_resize_u8")
    return out


_LANE_STREAMS = {}
class parked_heap:
    """``with parked_heap():`` — no generation-2 collection on the thread that launches the graph replays: everything alive at entry moves
    to the collector's permanent generation (gc.freeze) and comes back at exit.  Re-entrant and shared between threads: the freeze is
    process-global, so the FIRST frame loop in parks the heap and the LAST one out thaws it (an unconditional unfreeze at the end of one
    render thawed the heap under a second render's loop), and a heap the embedding application froze itself (a pre-fork server:
    gc.get_freeze_count() > 0 at entry of the first loop) is left frozen."""

    _lock = threading.Lock()
    _users = 0
    _ours = False

    def __enter__(self):
        cls = parked_heap
        with cls._lock:
            if cls._users == 0:
                cls._ours = gc.get_freeze_count() == 0
                if cls._ours:
                    gc.freeze()
            cls._users += 1
        return self

    def __exit__(self, *exc):
        cls = parked_heap
        with cls._lock:
            cls._users -= 1
            if cls._users == 0 and cls._ours:
                gc.unfreeze()
                cls._ours = False
        return False


_RING_LOCKS = {}  # device index -> lock held by the single-GPU render loop while it uses that device's rings (two renders in two threads)
_PINNED_RING = {}  # device index -> pinned staging slots of the single-GPU render loop (reallocated when the frame shape changes)
_DEVICE_RING = {}  # device index -> their device-side twins (a batch leaves its lane's frame buffer before it crosses PCIe)


def release_rings(device=None):
    """Free the pinned / device staging rings of the single-GPU render loop (6 x batch frames each, kept across renders by default) for
    ``device`` (index or torch.device; default: every device)."""
    index = None if device is None else (th.device(device).index if not isinstance(device, int) else device)
    for ring in (_PINNED_RING, _DEVICE_RING):
        for key in list(ring):
            if index is None or key == index:
                with _RING_LOCKS.setdefault(key, threading.Lock()):
                    del ring[key]


def _lane_stream(dev, k):
    # lane streams are kept per device: the caching allocator pools freed blocks per stream, so fresh streams on
    # every call would strand the staging buffers of the previous render
    stream = _LANE_STREAMS.get((dev.index, k))
    if stream is None:
        stream = _LANE_STREAMS[(dev.index, k)] = th.cuda.Stream(dev)
    return stream


def graph_lanes(generator, batch_size, n_lanes, bends=()):
    """``n_lanes`` captured forwards (GraphLane) of ``batch_size`` frames with uint8 frame output, each on its own stream.
    Without bends they are cached on the generator and reused by every later render (a captured forward reads its inputs
    through a frame source, so nothing about a particular render is baked in); a changed weight drops the cache.  With bends the
    transforms' static operands are part of the graph: captured per call."""
    dev = device_of(generator)
    key = generator.weights_key()
    cache = generator.__dict__.setdefault("_graph_lanes", {})
    lanes = []
    tap = bool(getattr(generator, "tap_float_image", False))  # (parity tests: such lanes also write the fp32 image — other kernels arguments)
    for k in range(n_lanes):
        stream = _lane_stream(dev, k)
        lane = None if bends else cache.get((batch_size, k, tap))
        if lane is not None and lane.weights_key != key:
            lane = None
        if lane is None:
            stream.wait_stream(th.cuda.current_stream(dev))
            with th.cuda.stream(stream):
                lane = generator.capture_graph(batch_size, lane=k, frames_u8=True, bends=bends)
            stream.synchronize()
            if not bends:
                cache[(batch_size, k, tap)] = lane
        lanes.append((stream, lane))
    return lanes


def prepare(generator, batch_size, lanes=3, bends=False):
    """Everything of a render that does not depend on its inputs: weight packing, static buffers and — a captured forward reads
    its inputs through a frame source, so it needs none of them — the graph lanes themselves.  generate() calls this on the
    ranks that do not run the audio front end WHILE rank 0 runs it (multi-GPU jobs were front-end bound: the peers used to start
    loading / packing / capturing only after the scatter).  With bends the graphs are captured per render (the transforms'
    operands are part of them), so only the warm-up forward is done here."""
    if not hasattr(generator, "capture_graph") or (bends and not getattr(generator, "capturable_bends", True)):
        return 0
    if bends:
        dev = device_of(generator)
        zeros = th.zeros(batch_size, generator.n_latent, generator.style_dim, device=dev)
        generator(styles=zeros, noise=None, truncation=1.0, randomize_noise=False, input_is_latent=True)
        return 0
    return len(graph_lanes(generator, batch_size, lanes))


def _sequence_bends(bends, n_frames=None):
    """The render's bends with every modulated transform instantiated ONCE on the modulation of the whole sequence (the reference
    rebuilds it per batch from the batch's slice, render.py:151-158).  Returns (bends, capturable): capturable when every
    transform implements ``run_static`` (audioreactive/bend.py: picks the frame's parameters on the device) and — ``n_frames``
    given — its per-frame table has one row or one row per frame of the sequence: the captured kernel indexes it with
    frame0 + b unchecked, so anything else (a modulation shorter than the frame range, an un-modulated transform built with
    [batch, k] parameters) keeps the eager per-batch path, which slices and validates like the reference."""
    out, capturable = [], True
    for bend in bends:
        transform = bend["transform"](bend["modulation"]) if "modulation" in bend else bend["transform"]
        if not hasattr(transform, "run_static") or not getattr(transform, "capturable", True):
            return None, False
        rows = getattr(transform, "sequence_rows", None)
        if n_frames is not None and rows is not None and rows not in (1, n_frames):
            return None, False
        out.append({"layer": bend["layer"], "transform": transform})
    return out, capturable


def synthesize(generator, latents, noise, batch_size, truncation=1.0, bends=(), rewrites=None, randomize_noise=False,
               use_graph=True, frame_range=None, lanes=3):
    """Generator -> uint8 frames for ``frame_range`` (default: all) of the sequence.  Yields (first_frame_index,
    uint8 device tensor [b, H, W, 3]) per batch, in order, with the producing stream current; the tensor stays valid
    until ``lanes`` further batches have been requested.

    hipGraph path: ``lanes`` graphs of ``batch_size`` frames (same weights, private activations) are replayed round-robin
    on their own streams, so consecutive batches overlap on the device — the small, latency-bound 4^2..32^2 layers and
    the last partial wave of every big launch of one batch run underneath the other batch's MFMA-bound layers.  The
    sequences (latents, noise maps, truncation, bend modulations) stay resident in HBM; a replay moves one frame index."""
    dev = device_of(generator)
    n_total = len(latents)
    lo, hi = frame_range if frame_range is not None else (0, n_total)
    latents = latents.to(dev, th.float32).contiguous()  # resident in HBM for the whole render
    noise = [None if nz is None else nz.to(dev, th.float32).contiguous() for nz in noise]
    if isinstance(truncation, (int, float)):
        truncation = float(truncation)
        # a float != 1 (or a generator that carries a truncation latent) must reach the captured graph as well: it becomes a
        # per-frame truncation sequence filled with the constant (reference models/stylegan2.py:537-543 lerps every batch);
        # 1.0 without a truncation latent is the exact identity
        needs_lerp = truncation != 1.0 or getattr(generator, "truncation_latent", None) is not None
        trunc_t = th.full((n_total,), truncation, dtype=th.float32, device=dev) if needs_lerp else None
    else:
        trunc_t = truncation.to(dev, th.float32).contiguous()
    bends = list(bends or [])
    for bend in bends:
        if "modulation" in bend:
            bend["modulation"] = bend["modulation"].to(dev, th.float32).contiguous()
    # model rewriting (reference render.py:127-131,160-167, whose `.copy()` typo makes it unreachable there): per batch
    # the named parameter is replaced by transform(original weight), transform = rewrite(modulation[batch]).
    rewrites = dict(rewrites or {})
    param_dict = dict(generator.named_parameters())
    original_weights = {}
    for name, (rewrite, modulation) in rewrites.items():
        if name not in param_dict:
            raise KeyError(f"get_rewrites: generator has no parameter {name!r}")
        rewrites[name] = [rewrite, modulation.to(dev, th.float32).contiguous()]
        original_weights[name] = param_dict[name].detach().clone()
    capturable = use_graph and not rewrites and not randomize_noise and hasattr(generator, "capture_graph")
    seq_bends = []
    if capturable and bends:
        seq_bends, capturable = _sequence_bends(bends, n_total) if getattr(generator, "capturable_bends", True) else (None, False)
    n_lanes = max(1, int(lanes)) if capturable else 1
    caller_stream = th.cuda.current_stream(dev)
    lane_state = []  # (stream, GraphLane or None)
    if capturable and hi - lo >= batch_size:
        lane_state = graph_lanes(generator, batch_size, min(n_lanes, (hi - lo) // batch_size), seq_bends)
        n_lanes = len(lane_state)
        for stream, lane in lane_state:
            lane.bind(latents, noise, trunc_t)  # once per render: the pointers of the HBM-resident sequences
            stream.wait_stream(caller_stream)
    else:
        stream = _lane_stream(dev, 0)
        stream.wait_stream(caller_stream)
        lane_state = [(stream, None)]
        n_lanes = 1
    eager_u8 = None

    try:
        k = 0
        for n in range(lo, hi, batch_size):
            m = min(n + batch_size, hi)
            b = m - n
            stream, lane = lane_state[k % n_lanes]
            with th.cuda.stream(stream):
                if lane is not None and b == batch_size:
                    lane.replay(n)
                    yield n, lane.u8  # the frame epilogue is part of the captured forward (fused into the last ToRGB)
                    k += 1
                    continue
                noise_batch = [None if nz is None else (nz if nz.shape[0] == 1 else nz[n:m]) for nz in noise]  # [1, ...] = one map for every frame
                bend_batch = []
                for bend in bends:
                    if "modulation" in bend:
                        transform = bend["transform"](bend["modulation"][n:m])
                        bend_batch.append({"layer": bend["layer"], "transform": transform})
                    else:
                        bend_batch.append({"layer": bend["layer"], "transform": bend["transform"]})
                for name, (rewrite, modulation) in rewrites.items():
                    new_weight = rewrite(modulation[n:m])(original_weights[name]).to(dev, th.float32).contiguous()
                    module = generator
                    *path, leaf = name.split(".")
                    for attr in path:
                        module = getattr(module, attr)
                    setattr(module, leaf, th.nn.Parameter(new_weight, requires_grad=False))
                if n_lanes > 1 or lane is not None:  # the eager tail batch shares lane 0's activations: let every lane drain first
                    for other, _ in lane_state:
                        stream.wait_stream(other)
                images, _ = generator(styles=latents[n:m], noise=noise_batch,
                                      truncation=truncation if trunc_t is None else trunc_t[n:m],
                                      transform_dict_list=bend_batch, randomize_noise=randomize_noise,
                                      input_is_latent=True)
                if eager_u8 is None or eager_u8.shape[0] != b or eager_u8.shape[1:3] != images.shape[2:]:
                    eager_u8 = th.empty((b, images.shape[2], images.shape[3], 3), dtype=th.uint8, device=dev)
                frames_to_uint8(images, eager_u8)
                yield n, eager_u8
            k += 1
    finally:  # also when the consumer stops early (sink error, generator closed)
        for name, w in original_weights.items():  # leave the generator as it was found
            module = generator
            *path, leaf = name.split(".")
            for attr in path:
                module = getattr(module, attr)
            setattr(module, leaf, th.nn.Parameter(w, requires_grad=False))
        for stream, lane in lane_state:
            caller_stream.wait_stream(stream)
            if lane is not None:
                lane.release()  # cached lanes outlive the render: they must not keep its sequences alive (or replay into them)


def render(generator, latents, noise, offset, duration, batch_size, out_size, output_file, audio_file=None,
           truncation=1.0, bends=[], rewrites={}, randomize_noise=False, ffmpeg_preset="slow"):
    """Drop-in for reference render.render (render.py:14-29).  With torch.distributed initialised (one process per
    GPU) every rank renders a contiguous shard of the frames and streams them, batch by batch, to rank 0's ordered sink
    (sharding.FrameStream)."""
    return render_shard(generator, latents, noise, offset, duration, batch_size, out_size, output_file, audio_file,
                        truncation, bends, rewrites, randomize_noise, ffmpeg_preset, None)


def render_shard(generator, latents, noise, offset, duration, batch_size, out_size, output_file, audio_file, truncation,
                 bends, rewrites, randomize_noise, ffmpeg_preset, _shard, transport=None):
    """``render`` with an optional ``_shard = (lo, hi, n_frames)``: set by generate() after sharding.scatter_frames, it says
    that ``latents`` / ``noise`` / ``truncation`` / bend modulations already hold only this rank's block of the frames."""
    width, height = _output_dims(out_size)
    rank, world = sharding.rank_world()
    # multi-GPU frame transport: "gather" (default: RCCL gather of every round into rank 0's HBM, sharding.FrameStream) or "host"
    # (per-rank D2H into shared memory, sharding.HostFrameStore)
    transport = transport or os.environ.get("MAUA_FRAME_TRANSPORT", "gather")
    if transport not in ("gather", "host"):
        raise ValueError(f"unknown frame transport {transport!r} (gather | host)")
    if _shard is None:
        n_frames = len(latents)
        lo, hi = sharding.shard_bounds(n_frames, rank, world)
        frame_range = (lo, hi)
    else:
        lo, hi, n_frames = _shard
        frame_range = (0, hi - lo)
    dev = device_of(generator)
    sink = None
    if rank == 0:
        sink = FrameSink(output_file, width, height, n_frames / duration, audio_file, offset, duration, ffmpeg_preset)

    worker = SinkWorker(sink) if sink is not None else None
    locked = False
    # The frame loop runs on the Python thread that launches every graph replay.  A generation-2 collection walks the ~170 k long-lived
    # objects of a process that has torch imported: 45-90 ms, i.e. 8-15 batches during which no replay is launched (measured: tools/gather_probe.py,
    # and as a 9 % hole in a 150-batch gathered bench region).  Park everything that is alive now in the permanent generation for the
    # duration of the loop: what the loop allocates is then all the collector ever walks.  (No full gc.collect() here: it would cost the same
    # 45-90 ms up front, as much as it saves on a 900-frame job; generate() has just run one, as the reference does.)
    parked = parked_heap()
    parked.__enter__()
    try:
        if not sharding.grouped():
            # pinned staging ring: the D2H of batch k overlaps the replays of the next batches; the sink thread writes a slot and
            # hands it back through `free` (the launch thread blocks here only when the sink is `n_slots` batches behind)
            n_lanes, n_slots = 3, 6
            copy_stream = th.cuda.Stream(dev)
            index = dev.index if dev.index is not None else th.cuda.current_device()  # torch.device("cuda") carries no index
            ring_lock = _RING_LOCKS.setdefault(index, threading.Lock())
            ring_lock.acquire()  # the rings are per device, not per render: a second render on this device (another thread) waits
            locked = True
            pinned = _PINNED_RING.setdefault(index, [None] * n_slots)  # kept across renders: pinning 6 x 25 MB is ~40 ms
            staged = _DEVICE_RING.setdefault(index, [None] * n_slots)
            free = queue.Queue()
            for i in range(n_slots):
                free.put(i)
            resized = {}
            for first, u8 in synthesize(generator, latents, noise, batch_size, truncation, bends, rewrites,
                                        randomize_noise, lanes=n_lanes):
                u8 = crop_resize_for_delivery(u8, out_size, resized)  # 2048-px frames leave the device as 1920x1080 already
                slot = free.get()
                count = u8.shape[0]
                # slots hold a FULL batch; the tail batch of a render uses a prefix (re-pinning per shape cost 2 x 6.5 ms per render)
                if pinned[slot] is None or pinned[slot].shape[1:] != u8.shape[1:] or pinned[slot].shape[0] < max(count, batch_size):
                    shape = (max(count, batch_size),) + tuple(u8.shape[1:])
                    pinned[slot] = th.empty(shape, dtype=th.uint8).pin_memory()
                    staged[slot] = th.empty(shape, dtype=th.uint8, device=dev)
                host, held = pinned[slot][:count], staged[slot][:count]
                # the lane's frame buffer is overwritten by its next replay: the batch moves to a device-side slot on the lane's own
                # stream (25 MB inside HBM: ~20 us) and crosses PCIe from there, so that no lane ever waits for a host copy
                # (waiting for it — 0.5 ms per batch on the lane — cost the render loop the whole gain of the three lanes)
                held.copy_(u8, non_blocking=True)
                produced = th.cuda.Event()
                produced.record(th.cuda.current_stream(dev))
                with th.cuda.stream(copy_stream):
                    copy_stream.wait_event(produced)
                    host.copy_(held, non_blocking=True)
                    copied = th.cuda.Event()
                    copied.record(copy_stream)
                worker.submit(copied.synchronize, host.numpy(), count, lambda s=slot: free.put(s))
        elif transport == "host":
            # every rank copies its rounds to a pinned shared-memory segment over its own PCIe link; rank 0's sink thread reads the
            # segments in global order (sharding.HostFrameStore) — no xGMI traffic, no funnel through rank 0's link
            token = sharding.broadcast_object(f"{os.getpid():x}{int.from_bytes(os.urandom(4), 'little'):08x}" if rank == 0 else None)
            store = None
            reader = None
            k = 0
            resized = {}

            stop_reader = threading.Event()  # set when this rank's launch loop fails: the reader must not outlive the store

            def start_reader():
                def run():
                    # (worker.feed, not submit: a sink error must stay on the worker for the LAUNCH thread — popped here it would end
                    # this thread, clear itself, and the render would return a truncated video without an exception)
                    for _, count, host in store.rounds_in_order(stop=stop_reader.is_set):
                        if not worker.feed(None, host.numpy(), count, None):
                            return

                t = threading.Thread(target=run, name="maua-host-gather", daemon=True)
                t.start()
                return t

            try:
                for first, u8 in synthesize(generator, latents, noise, batch_size, truncation, bends, rewrites,
                                            randomize_noise, frame_range=frame_range):
                    u8 = crop_resize_for_delivery(u8, out_size, resized)
                    if store is None:
                        store = sharding.HostFrameStore(n_frames, batch_size, tuple(u8.shape[1:]), dev, token)
                        if rank == 0:
                            reader = start_reader()
                    store.push(k, u8)
                    k += 1
                if store is None:
                    store = sharding.HostFrameStore(n_frames, batch_size, _stream_frame_shape(generator, out_size), dev, token)
                    if rank == 0:
                        reader = start_reader()
                store.finish()
                if reader is not None:
                    reader.join()
                    worker.close()  # re-raises a sink error on the launch thread
            finally:
                if reader is not None and reader.is_alive():  # the launch loop failed: stop the reader BEFORE its segments go away
                    stop_reader.set()
                    reader.join()
                if store is not None:
                    store.close()
        else:
            # One asynchronous gather per batch-round, issued as soon as the round's frames exist: the transfer of round k
            # runs under the compute of rounds k+1.., rank 0 hands rounds to its sink thread as they land (its own block
            # first — the blocks are contiguous — while the peers' frames accumulate in its HBM store).
            stream = None
            k = 0
            resized = {}

            def deliver(block):
                for _, count, host, release in stream.drain_rounds(block=block):
                    worker.submit(None, host.numpy(), count, release)

            for first, u8 in synthesize(generator, latents, noise, batch_size, truncation, bends, rewrites,
                                        randomize_noise, frame_range=frame_range):
                u8 = crop_resize_for_delivery(u8, out_size, resized)
                if stream is None:  # the frame shape is whatever the generator (and its layer-0 bends) produce
                    stream = sharding.FrameStream(n_frames, batch_size, tuple(u8.shape[1:]), dev)
                stream.push(k, u8)
                k += 1
                if rank == 0:
                    deliver(False)
            if stream is None:  # a rank whose block is empty (more ranks than frames) still takes part in every round
                stream = sharding.FrameStream(n_frames, batch_size, _stream_frame_shape(generator, out_size), dev)
            stream.finish()
            if rank == 0:
                deliver(True)
            else:
                stream.wait_all()
        if worker is not None:
            worker.close()
    finally:  # the encoder process / output file must not outlive a failed render
        parked.__exit__(None, None, None)
        if worker is not None:
            try:
                worker.close()  # (also: every ring slot has been written before the rings are handed to the next render)
            except BaseException:  # noqa: BLE001 - (a second failure while unwinding must not mask the first)
                pass
        if locked:
            ring_lock.release()
        if sink is not None:
            sink.close()
    return sink.count if sink is not None else 0


def _stream_frame_shape(generator, out_size):
    """[H, W, 3] of the frames the generator produces for ``out_size`` (1920 / 1080 render 2048-px-wide / -high frames
    that the sink crops and resizes, render.py:98-105)."""
    side = int(getattr(generator, "size", 0)) or _output_dims(out_size)[0]
    if out_size == 1920:
        return (1080, 1920, 3) if side == 1024 else (side, 2 * side, 3)  # 2048-px frames are resized on the device before they travel
    if out_size == 1080:
        return (1920, 1080, 3) if side == 1024 else (2 * side, side, 3)
    return (side, side, 3)
