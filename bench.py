#!/usr/bin/env python3
"""bench.py — synthesized 1024^2 StyleGAN2 frames/s on N MI355X (BASELINE.json metric), with roofline + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--batches-per-step Q] [--size 1024] [--bends]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over Q (default 15) batches of B (default 8) synthetic frames on every rank.  The per-frame
sequences (latents, noise maps <= 256^2) are resident in HBM; a batch = one 4-byte frame-index write + one replay of the
hipGraph-captured generator forward (style affines, demod, 17 StyledConv, 9 ToRGB; the last layer writes uint8 NHWC frames,
the render.py:40-43 epilogue), batches alternating over `--lanes` graphs on their own streams exactly as render.synthesize
does.  Frames are independent, so ranks shard them with no data-path collective (weak scaling: Q * B frames per rank per step).
Weights are random-init (seeded numpy streams) of the real 1024^2 architecture; arithmetic is fp32 end to end.

Rank 0 prints ONE JSON line.  `roofline` is the single launch with the largest device time (measured live with HIP events on
the launch stream): `achieved` / `frac` count the flops the matrix cores EXECUTED (the Winograd forms issue a fraction of the
direct form's multiplies), the SURVEY-8d direct-conv figure sits under `roofline.algorithmic`; `roofline_time_dominant` is the
kernel instance with the largest share of the serial forward (all its launches of a batch), `conv_kernel_instances` the same
for every conv instance; `roofline_upfirdn2d` is the standalone upfirdn2d op on the Blur-after-
up-conv shape that carries 49 % of the path's upfirdn2d bytes, which BASELINE.json's metric names.
`side_configs` times BASELINE configs 2 (256^2 generator) and 5 (1024^2 with per-frame Translate + Zoom network bends,
captured) on this GPU.  `cpu_baseline` times oracle/ (the CPU restatement pinned to the reference by tests/golden) on the
host cores (BASELINE.md §4 legs).  `frame_check` compares the frames of the last replay of every lane with the eager
(un-captured) forward of the same frames, bit for bit.
"""
import argparse
import gc
import json
import os
import sys
import time

# one hardware queue per stream (3 graph lanes + copy stream + default stream; the HIP default is 4): see maua_stylegan2_amd/__init__.py.
# Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle: invalid argument otherwise)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32
# conv mode (ModulatedConv2d.conv_mode) -> (multiplies the matrix cores execute / direct-form multiplies, description)
EXECUTED = {
    0: (1.0, "direct implicit GEMM"),
    1: (1.0, "polyphase transposed conv: exactly the reference's multiplies"),
    2: (2.0 / 3.0, "winograd F(2,3) along x: executed MFMA flops = 2/3 algorithmic"),
    3: (0.5, "winograd F(4,3) along x: executed MFMA flops = 1/2 algorithmic"),
    4: (30.0 / 36.0, "polyphase transposed conv with F(2,2) on the even x-phase: executed MFMA flops = 30/36 algorithmic"),
    5: (1.0 / 3.0, "2-D winograd F(2x4,3x3): executed MFMA flops = 1/3 algorithmic"),
    6: (25.0 / 36.0, "polyphase transposed conv with F(2,2) on both axes: executed MFMA flops = 25/36 algorithmic"),
}
POOL = 64  # distinct batches of the HBM-resident sequence (POOL * B frames), cycled through


def pmc_traffic_table():
    """Newest profiles/r*_pmc_traffic.json (written by tools/make_profiles.py from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of THIS bench command): {kernel instance: {read_bytes, write_bytes, dispatches, avg_us, batch}} or None."""
    import glob

    paths = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")))
    if not paths:
        return None, None
    with open(paths[-1]) as f:
        return json.load(f), os.path.relpath(paths[-1], REPO)


def conv_flops_per_frame(size):
    """2*Cin*Cout*k^2*H_in*W_in per modulated conv."""
    from maua_stylegan2_amd.seeding import channels_for

    ch = channels_for(2)
    total = 2 * 512 * 512 * 9 * 16 + 2 * 512 * 3 * 16
    cin, res = 512, 4
    while res < size:
        res *= 2
        cout = ch[res]
        total += 2 * cin * cout * 9 * (res // 2) ** 2 + 2 * cout * cout * 9 * res * res + 2 * cout * 3 * res * res
        cin = cout
    return total


def time_calls(fn, iters, stream_ptr):
    """Average device time (ms) of `fn` over `iters` back-to-back launches (median of three such loops), HIP events on the launch stream."""
    from maua_stylegan2_amd import _lib

    for _ in range(3):
        fn()
    loops = []
    for _ in range(3):  # median of three loops: a loop that starts in a lower clock state (the chip is power-managed and the
        e0, e1 = _lib.HipEvent(), _lib.HipEvent()  # isolated launches follow host-side gaps) does not become the layer's figure
        e0.record(stream_ptr)
        for _ in range(iters):
            fn()
        e1.record(stream_ptr)
        loops.append(e0.elapsed_ms(e1) / iters)
    return sorted(loops)[1]


INSTANCES = {}  # bench row name -> rocprofv3 kernel instance name (filled by layer_breakdown)
FUSED_OVERLAP = {}  # breakdown row of a fused up-sampling layer -> tiles launched / tiles of the plain tiling (its executed-flops factor)
FUSED_GRID = {}     # ... -> threads of its launch (the fused layers share one kernel instance: the PMC table keys their launches by grid)


def layer_breakdown(g, batch, noise_batch, stream):
    """Per-kernel-family device time of one forward (eager launches on `stream`, HIP events).  ``noise_batch``: per layer a
    [batch or 1, 1, r, r] map (the first frames of the bench's sequences / the checkpoint buffers)."""
    from maua_stylegan2_amd import _lib

    sp = stream.cuda_stream
    rows = []
    bufs = lambda name, shape: g._buf(batch, "bench." + name, shape)  # noqa: E731
    info = g._table(batch)
    s = g._buf(batch, "styles", (batch, info["s_total"]))
    d = g._buf(batch, "demod", (info["d_total"],))
    ent = info["entries"]

    def demod_of(e):
        return d[e["d_off"]: e["d_off"] + batch * e["cout"]].view(batch, e["cout"])

    li = 0
    x = g._buf(batch, "const", (batch, 512, 4, 4))
    fuse1 = dict(module=g.to_rgb1, s_off=ent[1]["s_off"], skip=None, out=bufs("rgb1", (batch, 3, 4, 4)), store=True)
    out = g._buf(batch, "conv1", (batch, 512, 4, 4))
    if getattr(g.conv1, "last_path", "") == "const":  # (what the captured forward ran: conv1 as T s on the constant input, csrc/constconv.hip)
        rows.append(("conv1+to_rgb1 (constant input: y = T s, tail + partial ToRGB sums in the same launch; plane sum)", "modconv",
                     time_calls(lambda: g._run_const_conv(s, ent[0]["s_off"], demod_of(ent[0]), noise_batch[0], bufs, dict(fuse1), None, batch), 20, sp),
                     2 * 512 * 512 * 9 * 16 * batch + 2 * 512 * 3 * 16 * batch, 0))
        INSTANCES[rows[-1][0]] = "const_conv_kernel"
        fuse1 = None
    else:
        g.conv1.run(x, s, ent[0]["s_off"], demod_of(ent[0]), noise_batch[0], bufs, "c1", rgb=fuse1)
    if fuse1 is None:
        pass
    elif fuse1.get("done"):  # the low-resolution entry: convolution slabs, slab sum + tail + per-group ToRGB sums, plane sum (three launches)
        rows.append(("conv1+to_rgb1 (slabs, reduce + tail + partial ToRGB sums, plane sum)", "modconv",
                     time_calls(lambda: g.conv1.run(x, s, ent[0]["s_off"], demod_of(ent[0]), noise_batch[0], bufs, "c1", rgb=dict(fuse1)), 20, sp),
                     2 * 512 * 512 * 9 * 16 * batch + 2 * 512 * 3 * 16 * batch, 0))
    else:
        rows.append(("conv1", "modconv", time_calls(lambda: g.conv1.run(x, s, ent[0]["s_off"], demod_of(ent[0]), noise_batch[0], bufs, "c1"), 20, sp),
                     2 * 512 * 512 * 9 * 16 * batch, 0))
        rows.append(("to_rgb1", "torgb", time_calls(lambda: g.to_rgb1.run(out, s, ent[1]["s_off"], None, bufs("rgb1", (batch, 3, 4, 4))), 20, sp),
                     2 * 512 * 3 * 16 * batch, 4 * batch * (512 + 3) * 16))
    li = 2
    image = g._buf(batch, "rgb1", (batch, 3, 4, 4))
    # the style fold as Generator._forward_device plans it (no bends here): a producer stores its map multiplied by the consumer's styles
    # (post_off), the consumer then runs the kernel instance without the style multiplies (prescaled) — the launches timed below are the
    # instances the captured forward replays (the operand VALUES do not matter for the time)
    posted = False
    for n in range(g.log_size - 2):
        up, plain, rgb = g.convs[2 * n], g.convs[2 * n + 1], g.to_rgbs[n]
        cin, cout = up.conv.in_channel, up.conv.out_channel
        h = out.shape[2]
        e_up, e_pl, e_rgb = ent[li], ent[li + 1], ent[li + 2]
        nz1, nz2 = noise_batch[2 * n + 1], noise_batch[2 * n + 2]
        xin = out
        pre_up = posted
        post_up = e_pl["s_off"] if g.style_fold and plain.accepts_prescaled(2 * h, 2 * h) else None
        t_all = time_calls(lambda: up.run(xin, s, e_up["s_off"], demod_of(e_up), nz1, bufs, f"u{n}", prescaled=pre_up, post_off=post_up), 10, sp)
        posted = up.posted
        blur_bytes = 4 * batch * cout * ((2 * h + 1) ** 2 + (2 * h) ** 2)
        if getattr(up, "last_path", "pair") == "fused":
            # transposed conv + blur + noise + bias + activation as ONE kernel (maua_upconv_blur_f32) + its seam pass: a single row.  Its
            # overlapped tiling launches tiles_x * 64 / 2W times the columns and (H / 8 + 1) / (H / 8) times the rows of the plain kernel.
            tiles_x = (2 * h + 59) // 60  # (a tile keeps 60 of its 64 raw columns)
            FUSED_OVERLAP[f"convs.{2*n}.upconv+blur+noise+act (one kernel)"] = (tiles_x * 64.0 / (2 * h)) * ((h // 8 + 1) / (h // 8))
            rows.append((f"convs.{2*n}.upconv+blur+noise+act (one kernel)", "modconv_up_fused", t_all, 2 * cin * cout * 9 * h * h * batch,
                         4 * batch * (cin * h * h + cout * (2 * h) ** 2)))
            # (maua_upconv_blur_f32's kernel instance; the seam pass is up2d_seam_kernel)
            INSTANCES[rows[-1][0]] = f"modconv_up2d_kernel<8, 2, {'true' if pre_up else 'false'}, 32>"
            # workgroups = images x vertical segments x tile columns x 32-channel tiles; the segment count from the seam workspace the library
            # asks for ((segments - 1) x 6 rows of 2W floats per image and channel + 4)
            n_seg = (_lib.load().maua_upconv_blur_ws_floats(batch, cin, cout, h, h) - 4) // (batch * cout * 12 * h) + 1
            FUSED_GRID[rows[-1][0]] = batch * n_seg * tiles_x * (cout // 32) * 256
        elif getattr(up, "last_path", "pair") == "lowres":
            # the low-resolution entry (maua_upconv_blur_lowres_f32): polyphase convolution -> split-K slabs, then slab sum + blur + noise + act
            rows.append((f"convs.{2*n}.upconv+blur+noise+act (slabs, reduce + blur + tail)", "modconv_up", t_all, 2 * cin * cout * 9 * h * h * batch, 0))
            INSTANCES[rows[-1][0]] = _lib.last_modconv_instance()
        else:
            # transposed conv alone and blur tail alone (they are separate launches inside StyledConv.run)
            raw = bufs(f"raw{n}", (batch, cout, 2 * h + 1, 2 * h + 1))
            n_ws = _lib.load().maua_modconv_ws_floats(batch, cin, cout, h, h, up.conv.conv_mode(h, h))
            ws = g._buf(batch, "bench.ws", (max(n_ws, 1),)) if n_ws else None
            t_up = time_calls(lambda: up.conv.run(xin, s, e_up["s_off"], demod_of(e_up), raw, ws, prescaled=pre_up), 10, sp)
            rows.append((f"convs.{2*n}.upconv", "modconv_up", t_up, 2 * cin * cout * 9 * h * h * batch, 0))
            INSTANCES[rows[-1][0]] = _lib.last_modconv_instance()
            rows.append((f"convs.{2*n}.blur+noise+act", "upfirdn2d_tail", max(t_all - t_up, 1e-6), 16 * 2 * batch * cout * (2 * h) ** 2, blur_bytes))
        mid = g._buf(batch, f"convs.{2*n}", (batch, cout, 2 * h, 2 * h))
        img_in = image
        rgb_buf = bufs(f"rgb{n}", (batch, 3, 2 * h, 2 * h))
        is_last = n == g.log_size - 3
        fuse = dict(module=rgb, s_off=e_rgb["s_off"], skip=img_in, out=rgb_buf, store=not is_last)
        pre_pl = posted
        nxt = None if is_last else g.convs[2 * n + 2]
        post_pl = ent[li + 3]["s_off"] if (nxt is not None and g.style_fold and nxt.accepts_prescaled(2 * h, 2 * h)) else None
        plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}", rgb=fuse, prescaled=pre_pl, post_off=post_pl)
        posted = plain.posted
        fused = bool(fuse.get("done"))  # the generator folds ToRGB into the conv epilogue for <= 64-channel layers
        conv_flops = 2 * cout * cout * 9 * (2 * h) ** 2 * batch
        rgb_flops = 2 * cout * 3 * (2 * h) ** 2 * batch
        if fused:
            t_pl = time_calls(lambda: plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}", rgb=dict(fuse), prescaled=pre_pl,
                                                post_off=post_pl), 10, sp)
            # <= 64 channels: one launch; wider layers: the conv leaves per-tile partial ToRGB sums and a 3*m_tiles-plane pass adds them
            label = ("(fused)" if cout <= 64 else "(slabs, reduce + tail + partial ToRGB sums, plane sum)" if getattr(plain, "last_path", "") == "lowres"
                     else "(fused: partial sums + plane sum)")
            rows.append((f"convs.{2*n+1}+to_rgbs.{n} {label}", "modconv", t_pl, conv_flops + rgb_flops, 0))
        else:
            t_pl = time_calls(lambda: plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}", prescaled=pre_pl), 10, sp)
            rows.append((f"convs.{2*n+1}", "modconv", t_pl, conv_flops, 0))
        INSTANCES[rows[-1][0]] = _lib.last_modconv_instance()
        out = g._buf(batch, f"convs.{2*n+1}", (batch, cout, 2 * h, 2 * h))
        o2 = out
        if not fused:
            t_rgb = time_calls(lambda: rgb.run(o2, s, e_rgb["s_off"], img_in, rgb_buf), 10, sp)
            rows.append((f"to_rgbs.{n}", "torgb", t_rgb, rgb_flops, 4 * batch * (cout + 3 + 1) * (2 * h) ** 2))
        image = g._buf(batch, f"rgbs.{n}", (batch, 3, 2 * h, 2 * h))
        li += 3
    return rows


def cpu_baseline(size, budget_s=90.0):
    """Oracle (kind "port") on the host cores, the legs of BASELINE.md §4 on a bounded sample (~90 s): the benched 1024^2 generator
    at batch 1 (1 warm-up + 3 repetitions: the headline `value`), config 1 (256^2, 16 fixed latents, batch 8: 2 repetitions) and the
    1024^2 generator once at batch 8 while the budget lasts (the oracle is batch-parallel torch CPU code: that leg says what batching
    buys on the host)."""
    from maua_stylegan2_amd import seeding
    from oracle import stylegan2_oracle as so

    torch.set_grad_enabled(False)
    t_start = time.perf_counter()

    def leg(sz, n_frames, batch, reps, warm):
        sd = seeding.seeded_state_dict(sz, seed=0)
        n_latent = 2 * (sz.bit_length() - 1) - 2
        lat = seeding.seeded_latents(n_frames, n_latent, seed=1)
        noise = seeding.seeded_noise(n_frames, sz, seed=2)

        def once():
            t0 = time.perf_counter()
            for i in range(0, n_frames, batch):
                so.generator_forward(sd, lat[i:i + batch], [nz[i:i + batch] for nz in noise])
            return n_frames / (time.perf_counter() - t0)

        for _ in range(warm):
            once()
        rates = []
        for _ in range(reps):
            rates.append(once())
            if time.perf_counter() - t_start > budget_s:
                break
        return {"frames_per_s_mean": sum(rates) / len(rates), "frames_per_s_min": min(rates), "repetitions": len(rates),
                "frames_per_repetition": n_frames, "batch": batch, "size": sz}

    # (on the 128 host threads of the GPU box the oracle runs ~0.22 frames/s at 1024^2 and ~1 frame/s at 256^2: the legs are ordered
    # so that the headline leg always gets its three repetitions inside the budget)
    legs = {}
    if size != 256:
        legs[f"{size}_b1"] = leg(size, 1, 1, 3, 1)
    legs["config1_256_b8"] = leg(256, 16, 8, 2 if size != 256 else 3, 0 if size != 256 else 1)
    if size != 256 and time.perf_counter() - t_start < budget_s * 0.7:
        legs[f"{size}_b8"] = leg(size, 8, 8, 1, 0)
    head = legs[f"{size}_b1"] if size != 256 else legs["config1_256_b8"]
    cpu_model, topo = "unknown", {}
    try:
        with open("/proc/cpuinfo") as f:
            text = f.read()
        cpu_model = next((ln.split(":", 1)[1].strip() for ln in text.splitlines() if ln.startswith("model name")), cpu_model)
        cores, sockets, logical = set(), set(), 0
        for block in text.split("\n\n"):
            kv = {ln.split(":", 1)[0].strip(): ln.split(":", 1)[1].strip() for ln in block.splitlines() if ":" in ln}
            if "processor" in kv:
                logical += 1
                sockets.add(kv.get("physical id", "0"))
                cores.add((kv.get("physical id", "0"), kv.get("core id", kv["processor"])))
        topo = {"sockets": len(sockets), "physical_cores": len(cores), "logical_cpus": logical,
                "smt": (logical // max(len(cores), 1)) if cores else None}
    except (OSError, IndexError):
        pass
    threads = torch.get_num_threads()
    return {"value": head["frames_per_s_mean"], "unit": "frames/s", "cores": threads, "cpu_model": cpu_model,
            "cores_definition": "intra-op threads torch used for the oracle (torch.get_num_threads()); host topology under `host`",
            "host": topo, "kind": "port",
            "sample": (f"oracle/stylegan2_oracle.py (torch CPU fp32): {head['repetitions']} x {head['frames_per_repetition']} frame(s) of "
                       f"{size}x{size} at batch {head['batch']} after 1 warm-up (value = mean); legs = the other BASELINE.md §4 cases"),
            "legs": legs, "seconds": time.perf_counter() - t_start}


class Workload:
    """A generator + HBM-resident synthetic sequences + captured graph lanes, stepped as render.synthesize does."""

    def __init__(self, size, batch, n_lanes, dev, rank, use_dist, bends=False):
        from maua_stylegan2_amd import render, seeding
        from maua_stylegan2_amd.models.stylegan2 import Generator

        self.size, self.B, self.dev = size, batch, dev
        g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
        if rank == 0:  # only rank 0 holds the "checkpoint" (generate_audiovisual.load_generator): the other ranks keep their own random
            g.load_state_dict(seeding.seeded_state_dict(size, seed=0))  # initialisation until the flat broadcast below overwrites it
        self.g = g = g.to(dev).eval()
        self.collectives = {"broadcast_bytes": 0, "scatter_bytes": 0}
        # synthetic sequences resident in HBM: POOL * B frames of latents; noise maps for scales <= 256 are per-frame
        # (audio-reactive in the default plugin), 512/1024 use the checkpoint buffers (get_noise -> None)
        self.n_frames = n = POOL * batch
        sizes = seeding.noise_sizes(size)

        def sequences(r):  # rank r's block of the job's per-frame inputs
            gen = torch.Generator(device=dev)
            gen.manual_seed(1000 + r)
            lat = torch.randn(n, g.n_latent, 512, device=dev, generator=gen)
            return gen, lat, [torch.randn(n, 1, side, side, device=dev, generator=gen) if side <= 256 else None for side in sizes]

        if use_dist:
            # weights: rank 0 is the source of truth - ONE flat broadcast; per-frame inputs: rank 0 "ran the front end" for
            # the whole job and every rank receives only its contiguous block (sharding.scatter_frames, one dist.scatter per sequence),
            # exactly the hand-over of generate(); a group of one rank issues the same collectives (sharding.grouped)
            import torch.distributed as dist

            from maua_stylegan2_amd import sharding

            world = dist.get_world_size()
            sharding.broadcast_module(g)
            self.collectives["broadcast_bytes"] = sum(t.numel() * t.element_size() for t in list(g.parameters()) + list(g.buffers()))
            gen = torch.Generator(device=dev)
            gen.manual_seed(1000 + rank)
            full_lat, full_noise = None, [None] * len(sizes)
            if rank == 0:
                blocks = [sequences(r) for r in range(world)]
                gen = blocks[0][0]
                full_lat = torch.cat([b[1] for b in blocks])
                full_noise = [None if blocks[0][2][i] is None else torch.cat([b[2][i] for b in blocks]) for i in range(len(sizes))]
                del blocks
            self.latents = sharding.scatter_frames(full_lat, world * n, device=dev)
            self.noise = [sharding.scatter_frames(nz, world * n, device=dev) for nz in full_noise]
            assert self.latents.shape[0] == n and [nz is None for nz in self.noise] == [side > 256 for side in sizes]
            self.collectives["scatter_bytes"] = sum(t.numel() * 4 for t in [self.latents] + [nz for nz in self.noise if nz is not None])
            del full_lat, full_noise
        else:
            gen, self.latents, self.noise = sequences(rank)
        self.bend_spec = []
        if bends:  # BASELINE config 5: a modulated Translate at layer id 4 and a modulated Zoom at layer id 5 (16 x 16 features)
            from maua_stylegan2_amd.audioreactive import bend

            h = w = 16
            saw = (torch.arange(n, device=dev, dtype=torch.float32) % 180) / 180.0 * w
            translation = torch.stack([saw, torch.zeros_like(saw)], 1)
            zoom = 1.0 + 0.25 * torch.sin(torch.arange(n, device=dev, dtype=torch.float32) / 30.0) ** 2
            tnoise = 0.2 * torch.randn(1, 1, h, 5 * w, device=dev, generator=gen)
            self.bend_spec = [
                {"layer": 4, "modulation": translation, "transform": lambda b: bend.Translate(b, h, w, tnoise)},
                {"layer": 5, "modulation": zoom, "transform": lambda b: bend.Zoom(b, h, w)},
            ]
        seq_bends, ok = render._sequence_bends(self.bend_spec)
        assert ok
        self.lanes = render.graph_lanes(g, batch, n_lanes, seq_bends)
        for _, lane in self.lanes:
            lane.bind(self.latents, self.noise, None)
        self.n_lanes = len(self.lanes)
        self.count = 0  # batches issued so far
        self.last = [None] * self.n_lanes  # first frame of the last replay of every lane

    def batch(self):
        """Issue the next batch; returns (stream, lane)."""
        k = self.count % self.n_lanes
        stream, lane = self.lanes[k]
        frame0 = (self.count % POOL) * self.B
        lane.replay(frame0, stream.cuda_stream)
        self.last[k] = frame0
        self.count += 1
        return stream, lane

    def sync(self):
        for stream, _ in self.lanes:
            stream.synchronize()

    def eager_frames(self, frame0):
        """uint8 frames of [frame0, frame0 + B) through the eager (un-captured) forward of the same generator."""
        from maua_stylegan2_amd import render

        n, m = frame0, frame0 + self.B
        bend_batch = [{"layer": bd["layer"], "transform": bd["transform"](bd["modulation"][n:m])} for bd in self.bend_spec]
        img, _ = self.g(styles=self.latents[n:m], noise=[None if nz is None else nz[n:m] for nz in self.noise], truncation=1.0,
                        transform_dict_list=bend_batch, randomize_noise=False, input_is_latent=True)
        return render.frames_to_uint8(img)

    def check_frames(self):
        """The last replay of every lane against the eager forward of the same frames."""
        self.sync()
        worst = 0
        for k, (stream, lane) in enumerate(self.lanes):
            if self.last[k] is None:
                continue
            got = lane.u8.clone()
            want = self.eager_frames(self.last[k])
            torch.cuda.synchronize(self.dev)
            worst = max(worst, int((got.int() - want.int()).abs().max()))
        return worst


def time_region(wl, steps, bps, mode, use_dist, world):
    """Time ``steps`` steps of ``bps`` batches.  mode "synth": replay + uint8 epilogue only.  "gathered" (N > 1): every batch's
    frames also travel to rank 0 through sharding.FrameStream (one asynchronous RCCL gather per batch, as render() issues them);
    the clock stops when the last round has landed in rank 0's HBM — SURVEY.md 8d's definition of the metric ("uint8 frames
    gathered to rank 0").  "pcie": every batch's frames also go to the host exactly as render() sends them (device-side ring slot on
    the lane's stream, then the pinned ring on a copy stream; null sink)."""
    import torch.distributed as dist

    dev, B, size = wl.dev, wl.B, wl.size
    fs = None
    if mode == "gathered":
        from maua_stylegan2_amd import sharding

        # two streams of SEG rounds used alternately: rank 0's HBM store stays at 2 x world x SEG x B frames whatever --steps is (one
        # stream for the whole region would be world x steps x 15 x 8 frames: 90 GB on rank 0 at 8 GPUs and 30 steps)
        SEG = 60  # (a multiple of every lane count up to 6: a slot of a stream is always written on the same lane's stream)
        fs = [sharding.FrameStream(world * SEG * B, B, (size, size, 3), dev) for _ in range(2)]
    n_slots = 6
    pinned = [torch.empty((B, size, size, 3), dtype=torch.uint8).pin_memory() for _ in range(n_slots)] if mode == "pcie" else None
    staged = [torch.empty((B, size, size, 3), dtype=torch.uint8, device=dev) for _ in range(n_slots)] if mode == "pcie" else None
    copy_stream = torch.cuda.Stream(dev) if mode == "pcie" else None
    copied = [None] * n_slots
    wl.sync()
    # as render() does before its frame loop: no generation-2 collection (45-90 ms on this heap: tools/gather_probe.py) on the thread that
    # launches the replays; the survivors of one collection are parked in the permanent generation
    from maua_stylegan2_amd.render import parked_heap

    gc.collect()  # (generate() runs one right before render(), as the reference does: generate_audiovisual.py:194-205)
    parked = parked_heap().__enter__()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    k = 0
    for _ in range(steps):
        for _ in range(bps):
            stream, lane = wl.batch()
            if fs is not None:
                seg, r = divmod(k, SEG)
                if r == 0 and seg >= 2:
                    fs[seg % 2].reset()  # (its gathers were issued 60 .. 120 rounds ago: nothing left to wait for in practice)
                with torch.cuda.stream(stream):
                    fs[seg % 2].push(r, lane.u8)
            if mode == "pcie":
                slot = k % n_slots
                if copied[slot] is not None:
                    copied[slot].synchronize()  # the host consumed this slot (null sink) before it is overwritten
                with torch.cuda.stream(stream):
                    staged[slot].copy_(lane.u8, non_blocking=True)  # the batch leaves the lane's frame buffer inside HBM (render.py)
                produced = torch.cuda.Event()
                produced.record(stream)
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(produced)
                    pinned[slot].copy_(staged[slot], non_blocking=True)
                    copied[slot] = torch.cuda.Event()
                    copied[slot].record(copy_stream)
            k += 1
    if fs is not None:
        for f in fs:
            f.wait_all()
    if copy_stream is not None:
        copy_stream.synchronize()
    wl.sync()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    parked.__exit__(None, None, None)  # (the region parked the heap for itself: hand it back)
    if use_dist:
        dist.barrier()
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if fs is not None and k:
        # evidence for the `rccl` block (outside the clock): what the gathers moved, and whether the LAST round of every rank arrived in rank
        # 0's store intact (byte sums of the round on its producer against the sums of rank 0's slots)
        seg, r = divmod(k - 1, SEG)
        f = fs[seg % 2]
        mine = f.mine[r * B: (r + 1) * B].long().sum().reshape(1)
        if use_dist and dist.get_backend() != "nccl":
            mine = mine.cpu()  # (the gloo debug backend: small bookkeeping tensors travel from the host)
        sums = [mine.clone() for _ in range(world)]
        if use_dist:
            dist.all_gather(sums, mine)
        landed = [int(f.store[p, r * B: (r + 1) * B].long().sum()) for p in range(world)] if f.store is not None else None
        GATHER_EVIDENCE.update(
            rounds_per_rank=k, gather_calls_per_rank=k if f.grouped else 0, bytes_per_round_per_rank=B * size * size * 3,
            bytes_into_rank0=k * B * size * size * 3 * world if f.grouped else 0,
            transport="dist.gather(async_op=True) per batch-round (sharding.FrameStream)" if f.grouped else "local copy (no process group)",
            last_round_byte_sums_on_producers=[int(x) for x in sums], last_round_byte_sums_in_rank0_store=landed,
            payload_check=None if landed is None else ("ok" if landed == [int(x) for x in sums] else "FAILED"))
    return dt


GATHER_EVIDENCE = {}


def rccl_block(dev, rank, world, backend, wl):
    """Who took part in the job and what the collectives moved — so that a SCALE record proves N ranks on N devices.
    Every rank reports its process, host, device (index, name, UUID, PCI address); rank 0 also checks that the flat weight broadcast left
    every rank with the same parameters (fp64 checksums)."""
    import socket

    import torch.distributed as dist

    props = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "pid": os.getpid(), "host": socket.gethostname(), "device_index": dev.index, "device_name": props.name,
          "device_uuid": str(getattr(props, "uuid", "")) or None,
          "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
          "hbm_gib": round(props.total_memory / 2 ** 30, 1)}
    ranks = [None] * world
    dist.all_gather_object(ranks, me)
    check = torch.stack([t.detach().double().sum() for t in wl.g.parameters()]).sum().reshape(1)
    if backend != "nccl":
        check = check.cpu()
    sums = [check.clone() for _ in range(world)]
    dist.all_gather(sums, check)
    sums = [float(x) for x in sums]
    try:
        version = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:  # noqa: BLE001
        version = None
    uuids = [r["device_uuid"] or r["pci"] for r in ranks]
    return {"backend": backend, "backend_is": "RCCL (torch.distributed 'nccl' on ROCm)" if backend == "nccl" else backend,
            "rccl_version": version, "world_size": world, "ranks": ranks, "distinct_devices": len(set(zip((r["host"] for r in ranks), uuids))),
            "weights": {"collective": "dist.broadcast of one flat fp32 buffer (sharding.broadcast_module); ranks != 0 started from their own random init",
                        "bytes": wl.collectives["broadcast_bytes"], "param_checksums_equal_after_broadcast": len(set(sums)) == 1},
            "inputs": {"collective": "dist.scatter per sequence (sharding.scatter_frames): rank 0 produced every rank's latents / noise block",
                       "bytes_received_per_rank": wl.collectives["scatter_bytes"]},
            "frames": dict(GATHER_EVIDENCE)}


def self_launch(n, backend):
    """`python bench.py --gpus N` without a rendezvous in the environment: start N ranks of this same command line through
    torch.distributed.run (one process per GPU, 127.0.0.1 rendezvous on a free port), pass rank 0's JSON line through on stdout and
    return the launcher's exit code.  The reference's equivalent is DataParallel inside one process (generate_audiovisual.py:54-55).
    With the `nccl` backend (RCCL) every rank needs its own device: fewer visible devices than ranks is ONE JSON error line, rc 1,
    before anything is started."""
    import socket
    import subprocess

    if backend == "nccl" and not os.environ.get("MAUA_BENCH_RENDEZVOUS_ONLY"):
        have = torch.cuda.device_count()
        if have < n:
            print(json.dumps({"error": f"bench.py --gpus {n}: {have} GPU(s) visible to this process; the nccl (RCCL) backend needs one device per "
                                       "rank (MAUA_DIST_BACKEND=gloo runs the ranks on a shared device, for debugging the code path only)",
                              "n_gpus": n, "visible_gpus": have, "value": None}))
            return 1
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--self-launch"]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))  # (torchrun would pin it to 1: the cpu_baseline leg at N = 1 uses the host cores)
    print("bench.py: launching", " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def rendezvous_only(args, backend, rank, world):
    """MAUA_BENCH_RENDEZVOUS_ONLY=1 (tests on a box without a GPU): the ranks the launcher started meet on the process group, rank 0 prints
    who arrived as one JSON line, nothing touches a device."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else backend)
    ranks = [None] * world
    dist.all_gather_object(ranks, {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "pid": os.getpid()})
    dist.barrier()
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": world, "world_size": dist.get_world_size(), "steps": args.steps, "warmup": args.warmup,
                          "ranks": ranks}))
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30, help="timed steps (default: ~3.3 s of device time at 15 x 8 frames/step)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="frames per batch = per graph replay (reference default --batch 8)")
    ap.add_argument("--batches-per-step", type=int, default=15,
                    help="batches per rank per step: a step is long enough (~0.11 s) that a 20-step run times > 2 s")
    ap.add_argument("--size", type=int, default=1024, help="generator resolution: 1024 (BASELINE configs 3-5) or 256 (configs 1-2)")
    ap.add_argument("--bends", action="store_true",
                    help="BASELINE config 5 workload: per-frame Translate (layer 4) + Zoom (layer 5) network bends inside the captured forward")
    ap.add_argument("--lanes", type=int, default=3,
                    help="hipGraphs replayed round-robin on their own streams (consecutive batches overlap on the device, "
                         "as render.synthesize does); 1 = strictly serial batches")
    ap.add_argument("--force-gather", action="store_true",
                    help="debug: run the N > 1 timed region (per-batch FrameStream push + landing wait) on one GPU as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--no-side-configs", action="store_true", help="skip the config 2 / config 5 side measurements")
    ap.add_argument("--no-pcie-side", action="store_true", help="skip the host-inclusive side measurement (copy accounting runs)")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B switch of an EXPERIMENTS build (tools/build_exp.sh, pass it with --lib): maua_tuning_set(KEY, VALUE) before "
                         "the graphs are captured; the product library has no such entry")
    ap.add_argument("--lib", default=None, help="A/B / ablation switch: load this build of libmaua_hip.so instead of the in-tree one")
    ap.add_argument("--no-style-fold", action="store_true",
                    help="A/B switch: every convolution multiplies its input by its styles itself (round 5's form) instead of reading a map its producer pre-multiplied")
    ap.add_argument("--no-lowres-fusion", action="store_true",
                    help="A/B switch: the 4^2 .. 32^2 layers as convolution, reduce + tail, blur / ToRGB (separate launches) instead of the low-resolution entries")
    ap.add_argument("--no-small-winograd", action="store_true",
                    help="A/B switch: the 8^2 / 16^2 plain layers on the direct kernel instead of Winograd F(2,3) along x")
    ap.add_argument("--no-partial-rgb", action="store_true", help="A/B switch: ToRGB of the >= 128-channel layers as a separate pass over the feature map")
    ap.add_argument("--wino2d-min-cout", type=int, default=None,
                    help="A/B switch: override ModulatedConv2d.winograd2d_min_cout (smallest layer that runs the 2-D Winograd kernel)")
    ap.add_argument("--fused-blur-min-width", type=int, default=None,
                    help="A/B switch: override StyledConv.fused_blur_min_width (narrowest up-sampling layer input that runs transposed conv + blur + "
                         "noise + activation as ONE kernel; a huge value = always the two-launch path)")
    ap.add_argument("--up2d-min-cout", type=int, default=None,
                    help="A/B switch: override ModulatedConv2d.upwino2d_min_cout (smallest transposed layer on the 2-D F(2,2) kernel)")
    ap.add_argument("--self-launch", action="store_true",
                    help="start the ranks through torch.distributed.run from this process also when --gpus is 1 (--gpus N > 1 without a "
                         "rendezvous in the environment always does)")
    args = ap.parse_args()

    # MAUA_DIST_BACKEND=gloo (debug): run the N > 1 code path with several ranks SHARING one GPU — RCCL wants one GPU per rank, gloo moves
    # device tensors for the collectives this path uses; the rate measured that way is meaningless, the code path is the real one
    backend = os.environ.get("MAUA_DIST_BACKEND", "nccl")
    if "RANK" not in os.environ and (args.gpus > 1 or args.self_launch):
        # `python bench.py --gpus N` (the shape the driver uses at N = 1): this process becomes the launcher of N ranks, one per GPU
        sys.exit(self_launch(args.gpus, backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print(json.dumps({"error": f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", "n_gpus": args.gpus, "world_size": world}))
        sys.exit(2)
    if os.environ.get("MAUA_BENCH_RENDEZVOUS_ONLY"):
        sys.exit(rendezvous_only(args, backend, rank, world))
    device_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ  # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    torch.set_grad_enabled(False)

    from maua_stylegan2_amd import _lib, seeding
    from maua_stylegan2_amd.models.stylegan2 import ModulatedConv2d, StyledConv

    if args.no_partial_rgb:
        StyledConv.partial_rgb_fusion = False
    if args.no_lowres_fusion:
        StyledConv.lowres_fusion = False
    if args.no_small_winograd:
        ModulatedConv2d.winograd_small_min_cout = 1 << 30
    if args.no_style_fold:
        from maua_stylegan2_amd.models.stylegan2 import Generator

        Generator.style_fold = False
    if args.wino2d_min_cout is not None:
        ModulatedConv2d.winograd2d_min_cout = args.wino2d_min_cout
    if args.up2d_min_cout is not None:
        ModulatedConv2d.upwino2d_min_cout = args.up2d_min_cout
    if args.fused_blur_min_width is not None:
        StyledConv.fused_blur_min_width = args.fused_blur_min_width

    if args.lib:
        _lib.LIB_PATH = os.path.abspath(args.lib)
    lib = _lib.load()
    for kv in args.tuning:
        key, value = kv.split("=")
        try:
            tuning_set = lib.maua_tuning_set
        except AttributeError:
            sys.exit("--tuning needs an experiments build of the library (tools/build_exp.sh <name>; bench.py --lib tools/bin/libmaua_<name>.so)")
        tuning_set(int(key), int(value))
    size, B, bps = args.size, args.batch, max(1, args.batches_per_step)
    n_lanes = max(1, args.lanes)

    wl = Workload(size, B, n_lanes, dev, rank, use_dist, bends=args.bends)
    g = wl.g
    for _ in range(args.warmup * bps):
        wl.batch()
    wl.sync()
    # the headline region: EXACTLY --steps steps.  One GPU: synthesis with the frames left in HBM (the PCIe-inclusive rate
    # is reported next to it).  Several GPUs: the frames of every batch are gathered to rank 0 inside the timed region.
    elapsed = time_region(wl, args.steps, bps, "gathered" if (world > 1 or args.force_gather) else "synth", use_dist, world)
    frame_err = wl.check_frames()
    extra = {}
    side = max(1, min(args.steps, 8))
    if world > 1 or args.force_gather:
        extra["frames_per_sec_synth_only"] = world * side * bps * B / time_region(wl, side, bps, "synth", use_dist, world)
    if world == 1 and not args.no_pcie_side:
        regions = sorted(side * bps * B / time_region(wl, side, bps, "pcie", use_dist, world) for _ in range(3))
        extra["frames_per_sec_pcie_inclusive"] = regions[1]  # median of three regions (they scatter by 5 % run to run on one box)
        extra["frames_per_sec_pcie_inclusive_regions"] = regions
        extra["pcie_inclusive_note"] = (f"median of three regions of {side} steps; every batch moves to a device-side ring slot on its lane's stream and from there to "
                                        "pinned host memory on a copy stream (6 slots), as render() does; null sink (no encoder)")

    rccl = rccl_block(dev, rank, world, backend, wl) if use_dist else None  # (collective: every rank calls it)
    result = None
    if rank == 0:
        frames = world * args.steps * bps * B
        fps = frames / elapsed
        result = {
            "metric": "1024^2 frames/sec (whole job)" if size == 1024 else f"{size}^2 frames/sec (whole job)",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"StyleGAN2-{size} generator (config {'5' if args.bends else ('3' if size == 1024 else '2')} shape: random-init, "
                                   f"channel_multiplier 2), a step = {bps} batches x {B} frames per GPU, hipGraph per batch "
                                   f"({wl.n_lanes} graphs round-robin on {wl.n_lanes} streams), latents + per-frame noise <=256^2 read from "
                                   f"HBM-resident sequences through a device-side frame index, uint8 NHWC frames written by the last layer"
                                   + (", per-frame Translate@layer4 + Zoom@layer5 network bends inside the captured forward" if args.bends else ""),
                       "frames_per_step_per_gpu": bps * B, "batch": B, "batches_per_step": bps, "lanes": wl.n_lanes,
                       "parallelism": f"frame-shard x{world}"},
            "frames_per_sec_per_gpu": fps / world,
            "ms_per_batch": 1000.0 * elapsed / (args.steps * bps),
            "timed_region_s": elapsed,
            "value_definition": ("uint8 frames of every batch gathered to rank 0's HBM (one async RCCL gather per batch, "
                                 "sharding.FrameStream) inside the timed region" if (world > 1 or args.force_gather) else
                                 "uint8 frames left in HBM (see frames_per_sec_pcie_inclusive for the host-inclusive rate)"),
            **extra,
            "conv_tflops_sustained": conv_flops_per_frame(size) * fps / world / 1e12,
            "frame_check": {"max_abs_grey_level_diff_graph_vs_eager": frame_err,
                            "what": "uint8 frames of the last replay of every lane vs the eager (un-captured) forward of the same frames"},
            "device": _lib.device_info(),
        }
        if rccl is not None:
            result["rccl"] = rccl
            if rccl["frames"].get("payload_check") == "FAILED" or not rccl["weights"]["param_checksums_equal_after_broadcast"]:
                print("bench.py: a collective delivered wrong bytes (see the rccl block)", file=sys.stderr)
                result["rccl"]["FAILED"] = True
        if frame_err != 0:
            print(f"bench.py: captured frames differ from the eager forward by {frame_err} grey levels", file=sys.stderr)
            result["frame_check"]["FAILED"] = True
    # ---- side configurations (rank 0's GPU only, after the headline region; N == 1 runs only)
    if rank == 0 and world == 1 and not args.no_side_configs and size == 1024 and not args.bends:
        sides = {}
        for name, (sz, bends) in {"config5_1024_bends": (1024, True), "config2_256": (256, False)}.items():
            try:
                w2 = Workload(sz, B, n_lanes, dev, rank, False, bends=bends)
                for _ in range(2 * n_lanes):
                    w2.batch()
                w2.sync()
                st = 4 if sz == 1024 else 8
                dt = time_region(w2, st, bps, "synth", False, 1)
                sides[name] = {"frames_per_s": st * bps * B / dt, "timed_region_s": dt, "frames": st * bps * B,
                               "graph_vs_eager_max_diff": w2.check_frames(), "lanes": w2.n_lanes}
                del w2
                torch.cuda.empty_cache()
            except Exception as exc:  # a side measurement must not take the headline down
                sides[name] = {"error": repr(exc)}
        # SIDE MEASUREMENT, never the headline: conv layers on the bf16 matrix cores with split-bf16 products
        # (csrc/modconv_sbf16.hip: a b ~= a_h b_h + a_h b_l + a_l b_h, fp32 accumulate) — the plain >= 128-channel layers (mode 7,
        # convs.5 / 7 / 9 / 11) and the transposed layers from 32^2 inputs up (mode 8, convs.6 / 8 / 10 / 12 / 14) instead of the fp32
        # Winograd / F(2,2) kernels; same machinery otherwise.  Reported with its error against the fp32 path.
        for key, plain_min, up_min in (("split_bf16_plain_layers", 128, 1 << 30), ("split_bf16_plain_and_transposed_layers", 128, 32)):
            try:
                ModulatedConv2d.split_bf16_min_cout, ModulatedConv2d.split_bf16_up_min_cout = plain_min, up_min
                w3 = Workload(size, B, n_lanes, dev, rank, False)
                for _ in range(2 * n_lanes):
                    w3.batch()
                w3.sync()
                dt = time_region(w3, 4, bps, "synth", False, 1)
                nb = [getattr(w3.g.noises, f"noise_{i}") if nz is None else nz[:B] for i, nz in enumerate(w3.noise)]
                modes = {}
                for i, c in enumerate(w3.g.convs):
                    res = 4 * 2 ** ((i + 1) // 2) if i % 2 else 4 * 2 ** (i // 2)
                    modes[f"convs.{i}"] = c.conv.conv_mode(res, res)
                img7, _ = w3.g(styles=w3.latents[:B], noise=nb, truncation=1.0, randomize_noise=False, input_is_latent=True)
                ModulatedConv2d.split_bf16_min_cout = ModulatedConv2d.split_bf16_up_min_cout = 1 << 30
                img5, _ = w3.g(styles=w3.latents[:B], noise=nb, truncation=1.0, randomize_noise=False, input_is_latent=True)
                sides[key] = {
                    "frames_per_s": 4 * bps * B / dt, "timed_region_s": dt, "frames": 4 * bps * B, "lanes": w3.n_lanes,
                    "relative_to_fp32_headline": 4 * bps * B / dt / (fps / world),
                    "layers_on_the_bf16_matrix_cores": [k for k, m in modes.items() if m in (7, 8)],
                    "max_abs_image_diff_vs_fp32_path": float((img7 - img5).abs().max()), "image_std": float(img5.std()),
                    "graph_vs_eager_max_diff": w3.check_frames() if False else None,
                    "dtype": "bf16 x 3 split products, fp32 accumulation (NOT the headline dtype)",
                    "what": "side measurement only: direct / polyphase 3x3 convolutions on v_mfma_f32_32x32x16_bf16"}
                del sides[key]["graph_vs_eager_max_diff"]
                del w3
                torch.cuda.empty_cache()
            except Exception as exc:
                sides[key] = {"error": repr(exc)}
            finally:
                ModulatedConv2d.split_bf16_min_cout = ModulatedConv2d.split_bf16_up_min_cout = 1 << 30
        # --stylegan1 (models/stylegan1.py G_style at 1024 px, random init): frames through render.synthesize's captured lanes
        try:
            from maua_stylegan2_amd import render
            from maua_stylegan2_amd.models import stylegan1 as sg1

            torch.manual_seed(7)
            g1 = sg1.G_style(output_size=1024, checkpoint=None, network_resolution=1024).to(dev).eval()
            n1 = 12 * B
            lat1 = torch.randn(n1, 18, 512, device=dev)
            noise1 = [torch.randn(n1, 1, *getattr(g1, f"noise_{i}").shape[2:], device=dev) if getattr(g1, f"noise_{i}").shape[-1] <= 256 else None
                      for i in range(len(g1.g_synthesis.blocks))]

            def run_sg1():
                last = None
                for _, u8 in render.synthesize(g1, lat1, noise1, B, truncation=0.7, lanes=n_lanes):
                    last = u8
                torch.cuda.synchronize(dev)
                return last

            run_sg1()  # captures the lanes
            t0 = time.perf_counter()
            run_sg1()
            dt = time.perf_counter() - t0
            sides["stylegan1_1024"] = {"frames_per_s": n1 / dt, "timed_region_s": dt, "frames": n1, "lanes": n_lanes,
                                       "what": "G_style (random init, 1024 px network) through render.synthesize: captured forward per batch, "
                                               "per-frame noise <= 256^2, truncation 0.7; the lanes copy their batch's inputs per replay (DESIGN 7)"}
            del g1
            torch.cuda.empty_cache()
        except Exception as exc:
            sides["stylegan1_1024"] = {"error": repr(exc)}
        if "frames_per_s" in sides.get("config5_1024_bends", {}):
            sides["config5_1024_bends"]["relative_to_plain"] = sides["config5_1024_bends"]["frames_per_s"] / (fps / world)
        result["side_configs"] = sides
        result["side_configs_note"] = ("same machinery as the headline (captured forward, 3 lanes, HBM-resident sequences): BASELINE config 5's "
                                       "workload = modulated Translate at layer id 4 + modulated Zoom at layer id 5 read per frame on the "
                                       "device; config 2's generator = StyleGAN2-256 with every noise scale per-frame")
    if rank == 0:
        if not args.no_breakdown and not args.bends:
            # the isolated launches below are timed in the chip's loaded state: ~0.4 s of the headline workload right in front of them
            # (after the side configurations the device has idled through their set-up)
            for _ in range(4 * bps):
                wl.batch()
            wl.sync()
        stream = wl.lanes[0][0]
        sp = stream.cuda_stream
        with torch.cuda.stream(stream):
            # ---- roofline legs (rank 0, N==1 style measurements on this rank's stream)
            if not args.no_breakdown and not args.bends:
                noise_batch = [getattr(g.noises, f"noise_{i}") if nz is None else nz[:B] for i, nz in enumerate(wl.noise)]
                g._forward_device(wl.latents[:B].contiguous(), noise_batch, None, None, [])  # populate the eager activations
                rows = layer_breakdown(g, B, noise_batch, stream)
                total_ms = sum(r[2] for r in rows)
                fam = {}
                for name, family, ms, flops, byts in rows:
                    fam.setdefault(family, [0.0, 0, 0])
                    fam[family][0] += ms
                    fam[family][1] += flops
                    fam[family][2] += byts
                result["kernel_families_note"] = (
                    "isolated eager launches on one stream (HIP events), summed per family; `share` is the share of THAT sum "
                    f"({total_ms:.3f} ms per batch) — the timed batches overlap {wl.n_lanes} graph lanes, so the sum exceeds ms_per_batch")
                result["kernel_families"] = {k: {"ms_isolated": v[0], "share": v[0] / total_ms,
                                                 "tflops": v[1] / v[0] / 1e9, "gbs": v[2] / v[0] / 1e6}
                                             for k, v in fam.items()}
                result["layers"] = [{"name": n, "ms": ms, "tflops": fl / ms / 1e9, "gbs": by / ms / 1e6} for n, _, ms, fl, by in rows]
                # ---- roofline of the conv kernels.  `frac` is what the matrix cores EXECUTED over the fp32-MFMA peak (the Winograd /
                # polyphase forms issue 1/3 .. 25/36 of the direct form's multiplies); the SURVEY-8d figure in algorithmic direct-conv
                # flops sits under `algorithmic` (its ratio to the peak can exceed 1 and is not an occupancy).
                def mode_of(row_name):
                    base = row_name.split("+")[0].split(" ")[0]
                    n_conv = int(base.split(".")[1]) if base.startswith("convs.") else None
                    if n_conv is None:
                        return 0
                    if "upconv" in base:
                        res = 4 * 2 ** (n_conv // 2)
                    else:
                        res = 4 * 2 ** ((n_conv + 1) // 2)
                    return g.convs[n_conv].conv.conv_mode(res, res)

                def roof(name, ms, algo_flops, ratio, algo_text):
                    ach = algo_flops * ratio / ms / 1e9
                    return {"kernel": name, "bound": "mfma", "achieved": ach, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                            "frac": ach / MFMA_F32_PEAK_TFLOPS, "traffic": None, "launch_ms": ms,
                            "achieved_definition": ("USEFUL flops the matrix cores executed (algorithmic direct-conv flops x executed_ratio of the "
                                                    "layer's algorithm; redundant tiles of an overlapped tiling NOT counted) / launch duration"),
                            "executed_ratio": ratio, "algorithm": algo_text,
                            "algorithmic": {"flops": algo_flops, "achieved": algo_flops / ms / 1e9,
                                            "frac": algo_flops / ms / 1e9 / MFMA_F32_PEAK_TFLOPS,
                                            "note": "direct-conv flops as SURVEY 8d / BASELINE.md 3.1 count them; can exceed the peak"}}

                def executed(row_name):
                    """(USEFUL executed / algorithmic matrix flops, description) of a breakdown row: the multiplies the layer's algorithm needs.  The
                    fused up-sampling layers launch FUSED_OVERLAP x as many tiles (overlapped tiling: 60 of 64 columns kept, one extra tile row);
                    those redundant products keep the matrix cores busy but are not throughput — they are reported beside, never inside, `frac`."""
                    if "constant input" in row_name:  # conv1 as y = T s (csrc/constconv.hip): fp32 VALU, 1/9 of the direct form's multiplies
                        return (0.0, "y = T s on the vector units: no matrix-core products")
                    if "up2d" in (INSTANCES.get(row_name) or ""):  # (the low-resolution entry runs the 16-wide layer on the F(2,2)^2 kernel)
                        return EXECUTED[6]
                    return EXECUTED.get(mode_of(row_name), (1.0, "direct form"))

                conv_rows = [r for r in rows if r[1].startswith("modconv")]
                # (a) the single launch with the largest device time
                dom = max(rows, key=lambda r: r[2])
                if dom[1].startswith("modconv"):
                    inst = INSTANCES.get(dom[0])
                    ratio, algo = executed(dom[0])
                    result["roofline"] = roof(f"{(inst or 'modconv').split('<')[0]} ({dom[0]})", dom[2], dom[3], ratio, algo)
                    result["roofline"]["kernel_instance"] = inst
                    result["roofline"]["what"] = "the single launch with the largest device time (isolated, HIP events on the launch stream)"
                    table, table_path = pmc_traffic_table()
                    rec = (table or {}).get("kernels", {}).get(inst) if inst else None
                    if rec is not None and rec.get("dispatches_per_step") != 1 and dom[0] in FUSED_GRID:
                        rec = rec.get("by_grid", {}).get(str(FUSED_GRID[dom[0]]))  # this launch among the launches of its instance
                    if rec is not None and table.get("batch") == B and table.get("size") == size and rec.get("dispatches_per_step") == 1:
                        # HBM bytes of this launch from the PMC passes of the SAME bench command (tools/profile_round.sh ->
                        # tools/make_profiles.py): FETCH_SIZE x2 (guide's gfx950 correction, calibrated on a known-size copy) +
                        # WRITE_SIZE.  Only used when that instance is launched once per batch, i.e. the average IS this launch.
                        result["roofline"]["traffic"] = rec["read_bytes"] + rec["write_bytes"]
                        result["roofline"]["traffic_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {table_path}"
                    if dom[0] in FUSED_OVERLAP:
                        # the fused up-sampling layer: `frac` = USEFUL products only; what the matrix cores executed INCLUDING the redundant
                        # tiles of its overlapped tiling (PMC-verifiable: SQ_INSTS_VALU_MFMA_MOPS_F32) is a utilisation, reported beside it
                        result["roofline"]["redundant_tile_factor"] = FUSED_OVERLAP[dom[0]]
                        result["roofline"]["matrix_core_utilisation_including_redundant_tiles"] = result["roofline"]["frac"] * FUSED_OVERLAP[dom[0]]
                        result["roofline"]["frac_without_redundant_tiles"] = result["roofline"]["frac"]  # (the name rounds 4-5 used for this figure)
                        # input map + activated output map + noise map + packed weight read / written once
                        result["roofline"]["algorithmic_bytes"] = dom[4] + B * size * size * 4 + 21 * 64 * 32 * 4
                        result["roofline"]["hbm_frac_of_launch"] = result["roofline"]["algorithmic_bytes"] / (dom[2] * 1e-3) / (HBM_PEAK_GBS * 1e9)
                        result["roofline"]["what"] += ("; this launch is the whole up-sampling StyledConv (transposed conv + blur + noise + bias + leaky ReLU, "
                                                       "maua_upconv_blur_f32): its blur epilogue is fp32 VALU work inside a matrix kernel")
                    if size == 1024 and dom[0].startswith("convs.15"):
                        # input + skip image + noise map + packed (Winograd-domain) weights, read once; uint8 frames written once
                        result["roofline"]["algorithmic_bytes"] = (B * 32 * size * size * 4 + B * 3 * (size // 2) ** 2 * 4
                                                                   + size * size * 4 + 24 * 32 * 32 * 4 + B * 3 * size * size)
                        result["roofline"]["hbm_frac_of_launch"] = (result["roofline"]["algorithmic_bytes"] / (dom[2] * 1e-3)
                                                                    / (HBM_PEAK_GBS * 1e9))
                else:
                    ach = dom[4] / dom[2] / 1e6
                    result["roofline"] = {"kernel": dom[0], "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                                          "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "launch_ms": dom[2]}
                # (b) the kernel INSTANCE with the largest share of the serial forward (all its launches of one batch together)
                by_inst = {}
                for name, family, ms, flops, byts in conv_rows:
                    inst = INSTANCES.get(name) or family
                    ratio = executed(name)[0]
                    acc = by_inst.setdefault(inst, {"ms": 0.0, "algo": 0.0, "exec": 0.0, "layers": []})
                    acc["ms"] += ms
                    acc["algo"] += flops
                    acc["exec"] += flops * ratio
                    acc["layers"].append(name)
                if by_inst:
                    inst, acc = max(by_inst.items(), key=lambda kv: kv[1]["ms"])
                    result["roofline_time_dominant"] = {
                        "kernel_instance": inst, "layers": acc["layers"], "ms_per_batch": acc["ms"], "share_of_serial_forward": acc["ms"] / total_ms,
                        "bound": "mfma", "achieved": acc["exec"] / acc["ms"] / 1e9, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": acc["exec"] / acc["ms"] / 1e9 / MFMA_F32_PEAK_TFLOPS,
                        "algorithmic": {"achieved": acc["algo"] / acc["ms"] / 1e9, "frac": acc["algo"] / acc["ms"] / 1e9 / MFMA_F32_PEAK_TFLOPS},
                        "what": "all launches of this template instance in one forward (isolated launch times summed); achieved = executed flops"}
                    # the whole forward: matrix flops EXECUTED per batch over the headline's time per batch (three lanes overlapped), as a
                    # fraction of the fp32-MFMA peak — what the chip's matrix cores are busy with, end to end
                    exec_per_batch = sum(v["exec"] for v in by_inst.values())
                    result["whole_forward_executed_tflops"] = exec_per_batch / (result["ms_per_batch"] * 1e-3) / 1e12
                    result["whole_forward_executed_frac"] = result["whole_forward_executed_tflops"] / MFMA_F32_PEAK_TFLOPS
                    result["whole_forward_executed_note"] = ("sum over the conv launches of (direct-conv flops x executed ratio of their algorithm; the "
                                                             "redundant tiles of the fused up-sampling layers not counted) per batch / ms_per_batch of the "
                                                             "timed region / 157.3 TFLOP/s")
                    redundant = sum(r[3] * executed(r[0])[0] * (FUSED_OVERLAP[r[0]] - 1.0) for r in conv_rows if r[0] in FUSED_OVERLAP)
                    result["whole_forward_matrix_core_utilisation"] = ((exec_per_batch + redundant) / (result["ms_per_batch"] * 1e-3) / 1e12
                                                                       / MFMA_F32_PEAK_TFLOPS)
                    result["conv_kernel_instances"] = {
                        k: {"ms_per_batch": v["ms"], "share_of_serial_forward": v["ms"] / total_ms, "executed_tflops": v["exec"] / v["ms"] / 1e9,
                            "executed_frac": v["exec"] / v["ms"] / 1e9 / MFMA_F32_PEAK_TFLOPS, "algorithmic_tflops": v["algo"] / v["ms"] / 1e9,
                            "layers": v["layers"]} for k, v in sorted(by_inst.items(), key=lambda kv: -kv[1]["ms"])}
            # standalone upfirdn2d on the largest Blur shape of this generator (the op BASELINE.json's metric names)
            r_out = size
            xin = torch.randn(B, 32 if size == 1024 else 128, r_out + 1, r_out + 1, device=dev)
            kern = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)).to(dev)
            yout = torch.empty(xin.shape[0] * xin.shape[1], r_out, r_out, 1, device=dev)
            major = xin.shape[0] * xin.shape[1]

            def launch_fir():
                lib.maua_upfirdn2d_f32(xin.data_ptr(), kern.data_ptr(), yout.data_ptr(), major, r_out + 1, r_out + 1, 1,
                                       4, 4, 1, 1, 1, 1, 1, 1, 1, 1, sp)

            ms = time_calls(launch_fir, 20, sp)
            byts = 4 * major * ((r_out + 1) ** 2 + r_out ** 2)
            ach = byts / ms / 1e6
            # HBM bytes per launch from the newest PMC passes (FETCH_SIZE x2 correction, WRITE_SIZE exact), scaled to this launch's planes
            table, table_path = pmc_traffic_table()
            fir_name = "fir_tile_kernel<4, 4, 4, false, 24>"
            kernels = (table or {}).get("kernels", {})
            # (older tables: the instance had no strip-height argument; rounds 1-3: the plain op was a kernel of its own)
            rec = kernels.get(fir_name) or kernels.get("fir_tile_kernel<4, 4, 4, false>") or kernels.get("fir_strip_kernel<4, 4, 4>")
            traffic = None
            if rec is not None and table.get("size") == size and rec.get("planes"):
                traffic = (rec["read_bytes"] + rec["write_bytes"]) * major / rec["planes"]  # scaled to this launch's planes
            result["roofline_upfirdn2d"] = {"kernel": fir_name, "bound": "hbm", "achieved": ach,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                            "traffic": traffic, "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {table_path}" if traffic else None), "launch_ms": ms, "algorithmic_bytes": byts,
                                            "shape": f"[{xin.shape[0]},{xin.shape[1]},{r_out+1},{r_out+1}] -> {r_out}^2"}
            # measured copy ceiling of this box (SURVEY.md 8d asks for "vs spec" and "vs measured copy"): a device-to-device copy of
            # the same number of bytes (read n/2, write n/2) on the same stream
            src = torch.empty(byts // 8, dtype=torch.float32, device=dev).normal_()
            dst = torch.empty_like(src)
            ms_copy = time_calls(lambda: dst.copy_(src), 20, sp)
            copy_gbs = 2 * src.numel() * 4 / ms_copy / 1e6
            result["roofline_upfirdn2d"]["measured_copy_gbs"] = copy_gbs
            result["roofline_upfirdn2d"]["frac_of_measured_copy"] = ach / copy_gbs
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(size)
        print(json.dumps(result))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
