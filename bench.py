#!/usr/bin/env python3
"""bench.py — synthesized 1024^2 StyleGAN2 frames/s on N MI355X (BASELINE.json metric), with roofline + CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--size 1024]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of B synthetic frames on every rank: copy the batch's latents
(already resident in HBM) into the graph's static inputs, replay the hipGraph-captured generator forward
(style affines, demod, 17 StyledConv, 9 ToRGB; the last ToRGB writes uint8 NHWC frames, render.py:40-43 epilogue).  Frames
are independent, so ranks shard them with no data-path collective (weak scaling: B frames per rank per step).
Weights are random-init (seeded numpy streams) of the real 1024^2 architecture; arithmetic is fp32 end to end.

Rank 0 prints ONE JSON line.  `roofline` is the dominant kernel of the frame (largest share of device time, measured
live with HIP events on the launch stream); `roofline_upfirdn2d` is the standalone upfirdn2d op on the Blur-after-
up-conv shape that carries 49 % of the path's upfirdn2d bytes, which BASELINE.json's metric names.
`cpu_baseline` times oracle/ (the CPU restatement pinned to the reference by tests/golden) on the host cores.
"""
import argparse
import json
import os
import sys
import time

# one hardware queue per stream (3 graph lanes + copy stream + default stream; the HIP default is 4): see maua_stylegan2_amd/__init__.py.
# Must be in the environment before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
MFMA_F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32


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
    """Average device time (ms) of `fn` over `iters` back-to-back launches, HIP events on the launch stream."""
    from maua_stylegan2_amd import _lib

    for _ in range(3):
        fn()
    e0, e1 = _lib.HipEvent(), _lib.HipEvent()
    e0.record(stream_ptr)
    for _ in range(iters):
        fn()
    e1.record(stream_ptr)
    return e0.elapsed_ms(e1) / iters


INSTANCES = {}  # bench row name -> rocprofv3 kernel instance name (filled by layer_breakdown)


def layer_breakdown(g, batch, static, stream):
    """Per-kernel-family device time of one forward (eager launches on `stream`, HIP events)."""
    from maua_stylegan2_amd import _lib
    from maua_stylegan2_amd.seeding import channels_for

    sp = stream.cuda_stream
    dev = static["latents"].device
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
    layers = [("conv1", g.conv1, x, 0)]
    rows.append(("conv1", "modconv", time_calls(lambda: g.conv1.run(x, s, ent[0]["s_off"], demod_of(ent[0]), static["noise"][0], bufs, "c1"), 20, sp),
                 2 * 512 * 512 * 9 * 16 * batch, 0))
    out = g._buf(batch, "conv1", (batch, 512, 4, 4))
    rows.append(("to_rgb1", "torgb", time_calls(lambda: g.to_rgb1.run(out, s, ent[1]["s_off"], None, bufs("rgb1", (batch, 3, 4, 4))), 20, sp),
                 2 * 512 * 3 * 16 * batch, 4 * batch * (512 + 3) * 16))
    li = 2
    image = g._buf(batch, "rgb1", (batch, 3, 4, 4))
    for n in range(g.log_size - 2):
        up, plain, rgb = g.convs[2 * n], g.convs[2 * n + 1], g.to_rgbs[n]
        cin, cout = up.conv.in_channel, up.conv.out_channel
        h = out.shape[2]
        e_up, e_pl, e_rgb = ent[li], ent[li + 1], ent[li + 2]
        nz1, nz2 = static["noise"][2 * n + 1], static["noise"][2 * n + 2]
        xin = out
        # transposed conv alone and blur tail alone (they are separate launches inside StyledConv.run)
        raw = bufs(f"raw{n}", (batch, cout, 2 * h + 1, 2 * h + 1))
        n_ws = _lib.load().maua_modconv_ws_floats(batch, cin, cout, h, h, up.conv.conv_mode(h, h))
        ws = g._buf(batch, "bench.ws", (max(n_ws, 1),)) if n_ws else None
        t_up = time_calls(lambda: up.conv.run(xin, s, e_up["s_off"], demod_of(e_up), raw, ws), 10, sp)
        rows.append((f"convs.{2*n}.upconv", "modconv_up", t_up, 2 * cin * cout * 9 * h * h * batch, 0))
        INSTANCES[rows[-1][0]] = _lib.last_modconv_instance()
        t_all = time_calls(lambda: up.run(xin, s, e_up["s_off"], demod_of(e_up), nz1, bufs, f"u{n}"), 10, sp)
        blur_bytes = 4 * batch * cout * ((2 * h + 1) ** 2 + (2 * h) ** 2)
        rows.append((f"convs.{2*n}.blur+noise+act", "upfirdn2d_tail", max(t_all - t_up, 1e-6), 16 * 2 * batch * cout * (2 * h) ** 2, blur_bytes))
        mid = g._buf(batch, f"convs.{2*n}", (batch, cout, 2 * h, 2 * h))
        img_in = image
        rgb_buf = bufs(f"rgb{n}", (batch, 3, 2 * h, 2 * h))
        is_last = n == g.log_size - 3
        fuse = dict(module=rgb, s_off=e_rgb["s_off"], skip=img_in, out=rgb_buf, store=not is_last)
        plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}", rgb=fuse)
        fused = bool(fuse.get("done"))  # the generator folds ToRGB into the conv epilogue for <= 64-channel layers
        conv_flops = 2 * cout * cout * 9 * (2 * h) ** 2 * batch
        rgb_flops = 2 * cout * 3 * (2 * h) ** 2 * batch
        if fused:
            t_pl = time_calls(lambda: plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}", rgb=dict(fuse)), 10, sp)
            # <= 64 channels: one launch; wider layers: the conv leaves per-tile partial ToRGB sums and a 3*m_tiles-plane pass adds them
            label = "(fused)" if cout <= 64 else "(fused: partial sums + plane sum)"
            rows.append((f"convs.{2*n+1}+to_rgbs.{n} {label}", "modconv", t_pl, conv_flops + rgb_flops, 0))
        else:
            t_pl = time_calls(lambda: plain.run(mid, s, e_pl["s_off"], demod_of(e_pl), nz2, bufs, f"p{n}"), 10, sp)
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


def cpu_baseline(size, max_seconds=25.0):
    """Oracle (kind "port") on the host cores: bounded sample of 1024^2 frames, batch 1."""
    from maua_stylegan2_amd import seeding
    from oracle import stylegan2_oracle as so

    torch.set_grad_enabled(False)
    sd = seeding.seeded_state_dict(size, seed=0)
    n_latent = 2 * (size.bit_length() - 1) - 2
    lat = seeding.seeded_latents(1, n_latent, seed=1)
    noise = seeding.seeded_noise(1, size, seed=2)
    so.generator_forward(sd, lat, noise)  # warm-up
    t0, n = time.perf_counter(), 0
    while n < 3 or (time.perf_counter() - t0 < max_seconds and n < 64):
        so.generator_forward(sd, lat, noise)
        n += 1
        if time.perf_counter() - t0 > max_seconds:
            break
    dt = time.perf_counter() - t0
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), cpu_model)
    except OSError:
        pass
    return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model, "kind": "port",
            "sample": f"{n} frames of {size}x{size}, batch 1, oracle/stylegan2_oracle.py (torch CPU fp32)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: ~2.5 s of device time at 8 frames/step)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="frames per rank per step (reference default --batch 8)")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--lanes", type=int, default=3,
                    help="hipGraphs replayed round-robin on their own streams (consecutive steps overlap on the device, "
                         "as render.synthesize does); 1 = strictly serial steps")
    ap.add_argument("--force-gather", action="store_true",
                    help="debug: run the N > 1 timed region (per-step FrameStream push + landing wait) on one GPU as well")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-breakdown", action="store_true")
    ap.add_argument("--tuning", action="append", default=[], metavar="KEY=VALUE",
                    help="A/B switch of an EXPERIMENTS build (tools/build_exp.sh, pass it with --lib): maua_tuning_set(KEY, VALUE) before "
                         "the graphs are captured; the product library has no such entry")
    ap.add_argument("--lib", default=None, help="A/B / ablation switch: load this build of libmaua_hip.so instead of the in-tree one")
    ap.add_argument("--no-partial-rgb", action="store_true", help="A/B switch: ToRGB of the >= 128-channel layers as a separate pass over the feature map")
    ap.add_argument("--wino2d-min-cout", type=int, default=None,
                    help="A/B switch: override ModulatedConv2d.winograd2d_min_cout (smallest layer that runs the 2-D Winograd kernel)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        print("bench.py --gpus N>1 must be launched through torch.distributed.run (one rank per GPU)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ  # launched by torch.distributed.run
    if use_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    torch.set_grad_enabled(False)

    from maua_stylegan2_amd import _lib, seeding
    from maua_stylegan2_amd.models.stylegan2 import Generator, ModulatedConv2d, StyledConv

    if args.no_partial_rgb:
        StyledConv.partial_rgb_fusion = False
    if args.wino2d_min_cout is not None:
        ModulatedConv2d.winograd2d_min_cout = args.wino2d_min_cout

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
    size, B = args.size, args.batch
    g = Generator(size, 512, 8, channel_multiplier=2, constant_input=True)
    g.load_state_dict(seeding.seeded_state_dict(size, seed=0))
    g = g.to(dev).eval()
    if use_dist:  # weights: rank 0 is the source of truth, one RCCL broadcast per tensor
        for t in list(g.parameters()) + list(g.buffers()):
            dist.broadcast(t.data, 0)

    # synthetic inputs resident in HBM: a pool of latents for every step of this rank; noise maps for scales <= 256
    # are per-frame (audio-reactive in the default plugin), 512/1024 use the checkpoint buffers (get_noise -> None).
    n_steps = args.steps + args.warmup
    POOL = 64  # distinct latent batches resident in HBM, cycled through (every step still refreshes the graph's inputs)
    lat_pool = seeding.seeded_latents(POOL * B, g.n_latent, seed=100 + rank).to(dev)
    sizes = seeding.noise_sizes(size)
    noise_shapes = [(r, r) if r <= 256 else None for r in sizes]
    noise_pool = [torch.from_numpy(seeding.seeded_array(200 + rank, f"n{i}", (B, 1, r, r))).to(dev) if r <= 256 else None
                  for i, r in enumerate(sizes)]

    n_lanes = max(1, args.lanes)
    lanes = []
    for lane_id in range(n_lanes):
        lane_stream = torch.cuda.Stream(dev)
        with torch.cuda.stream(lane_stream):
            lane_graph, lane_static = g.capture_graph(B, noise_shapes, lane=lane_id, frames_u8=True)
            for dst, src in zip(lane_static["noise"], noise_pool):
                if src is not None:
                    dst.copy_(src)
        lane_stream.synchronize()
        # (the uint8 NHWC frame epilogue of render.py:40-43 is part of the captured forward: fused into the last ToRGB)
        lanes.append({"stream": lane_stream, "graph": lane_graph, "static": lane_static, "u8": lane_static["u8"]})
    stream, graph, static = lanes[0]["stream"], lanes[0]["graph"], lanes[0]["static"]
    with torch.cuda.stream(stream):
        sp = stream.cuda_stream

        def step(i):
            # refresh the graph's static inputs from the HBM-resident sequence, exactly as render.synthesize does;
            # step i runs on lane i % n_lanes, so it overlaps with the previous step on the device
            lane = lanes[i % n_lanes]
            with torch.cuda.stream(lane["stream"]):
                sp_ = lane["stream"].cuda_stream
                j = i % POOL
                lane["static"]["latents"].copy_(lat_pool[j * B:(j + 1) * B], non_blocking=True)
                for dst, src in zip(lane["static"]["noise"], noise_pool):
                    if src is not None:
                        dst.copy_(src, non_blocking=True)
                lane["graph"].replay(sp_)

        def sync_lanes():
            for lane in lanes:
                lane["stream"].synchronize()

        def run_region(first_step, count, mode):
            """Time ``count`` steps.  mode "synth": replay + uint8 epilogue only.  "gathered" (N > 1): every step's frames
            also travel to rank 0 through sharding.FrameStream (one asynchronous RCCL gather per step, as render() issues
            them); the clock stops when the last round has landed in rank 0's HBM — SURVEY.md 8d's definition of the metric
            ("uint8 frames gathered to rank 0").  "pcie": every step's frames also go to the host through the pinned
            staging ring on a copy stream exactly as render() does (null sink)."""
            fs = None
            if mode == "gathered":
                from maua_stylegan2_amd import sharding

                fs = sharding.FrameStream(world * count * B, B, (size, size, 3), dev)
            n_slots = 3
            pinned = [torch.empty((B, size, size, 3), dtype=torch.uint8).pin_memory() for _ in range(n_slots)] if mode == "pcie" else None
            copy_stream = torch.cuda.Stream(dev) if mode == "pcie" else None
            copied = [None] * n_slots
            sync_lanes()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(count):
                i = first_step + k
                step(i)
                lane = lanes[i % n_lanes]
                if fs is not None:
                    with torch.cuda.stream(lane["stream"]):
                        fs.push(k, lane["u8"])
                if mode == "pcie":
                    slot = k % n_slots
                    if copied[slot] is not None:
                        copied[slot].synchronize()  # the host consumed this slot (null sink) before it is overwritten
                    produced = torch.cuda.Event()
                    produced.record(lane["stream"])
                    with torch.cuda.stream(copy_stream):
                        copy_stream.wait_event(produced)
                        pinned[slot].copy_(lane["u8"], non_blocking=True)
                        copied[slot] = torch.cuda.Event()
                        copied[slot].record(copy_stream)
                    lane["stream"].wait_event(copied[slot])  # the producer must not overwrite u8 before the copy read it
            if fs is not None:
                fs.wait_all()
            if copy_stream is not None:
                copy_stream.synchronize()
            sync_lanes()
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            if use_dist:
                dist.barrier()
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt

        for i in range(args.warmup):
            step(i)
        sync_lanes()
        # the headline region: EXACTLY --steps steps.  One GPU: synthesis with the frames left in HBM (the PCIe-inclusive rate
        # is reported next to it).  Several GPUs: the frames of every step are gathered to rank 0 inside the timed region.
        elapsed = run_region(args.warmup, args.steps, "gathered" if (world > 1 or args.force_gather) else "synth")
        extra = {}
        side = max(3, min(args.steps, 120))
        if world > 1:
            extra["frames_per_sec_synth_only"] = world * side * B / run_region(n_steps, side, "synth")
        else:
            extra["frames_per_sec_pcie_inclusive"] = side * B / run_region(n_steps, side, "pcie")
            extra["pcie_inclusive_note"] = (f"{side} steps; uint8 frames copied to pinned host memory through a 3-slot staging "
                                            "ring on a copy stream, as render() does; null sink (no encoder)")
        checksum = int(sum(int(lane["u8"].sum().item()) for lane in lanes))

        result = None
        if rank == 0:
            fps = world * args.steps * B / elapsed
            ms_per_step = 1000.0 * elapsed / args.steps
            result = {
                "metric": "1024^2 frames/sec (whole job)" if size == 1024 else f"{size}^2 frames/sec (whole job)",
                "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"StyleGAN2-{size} generator (config 3 shape: random-init, channel_multiplier 2), "
                                       f"{B} frames/step/GPU, hipGraph per batch ({n_lanes} graphs round-robin on {n_lanes} streams), "
                                       f"per-frame noise <=256^2, uint8 NHWC frames written by the last layer",
                           "frames_per_step_per_gpu": B, "lanes": n_lanes, "parallelism": f"frame-shard x{world}"},
                "frames_per_sec_per_gpu": fps / world,
                "timed_region_s": elapsed,
                "value_definition": ("uint8 frames of every step gathered to rank 0's HBM (one async RCCL gather per step, "
                                     "sharding.FrameStream) inside the timed region" if world > 1 else
                                     "uint8 frames left in HBM (see frames_per_sec_pcie_inclusive for the host-inclusive rate)"),
                **extra,
                "conv_tflops_sustained": conv_flops_per_frame(size) * fps / world / 1e12,
                "frame_checksum": checksum,
                "device": _lib.device_info(),
            }
            # ---- roofline legs (rank 0, N==1 style measurements on this rank's stream)
            if not args.no_breakdown:
                rows = layer_breakdown(g, B, static, stream)
                total_ms = sum(r[2] for r in rows)
                fam = {}
                for name, family, ms, flops, byts in rows:
                    fam.setdefault(family, [0.0, 0, 0])
                    fam[family][0] += ms
                    fam[family][1] += flops
                    fam[family][2] += byts
                result["kernel_families_note"] = (
                    "isolated eager launches on one stream (HIP events), summed per family; `share` is the share of THAT sum "
                    f"({total_ms:.3f} ms) — the timed steps overlap {n_lanes} graph lanes, so the sum exceeds ms_per_step")
                result["kernel_families"] = {k: {"ms_isolated": v[0], "share": v[0] / total_ms,
                                                 "tflops": v[1] / v[0] / 1e9, "gbs": v[2] / v[0] / 1e6}
                                             for k, v in fam.items()}
                result["layers"] = [{"name": n, "ms": ms, "tflops": fl / ms / 1e9, "gbs": by / ms / 1e6} for n, _, ms, fl, by in rows]
                # dominant kernel instance = the single launch with the largest time
                dom = max(rows, key=lambda r: r[2])
                if dom[1].startswith("modconv"):
                    ach = dom[3] / dom[2] / 1e9
                    inst = INSTANCES.get(dom[0])
                    result["roofline"] = {"kernel": f"{(inst or 'modconv').split('<')[0]} ({dom[0]})", "bound": "mfma", "achieved": ach,
                                          "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFLOPS,
                                          "traffic": None, "launch_ms": dom[2]}
                    # `achieved` counts ALGORITHMIC flops (direct 3x3).  Plain layers run through Winograd
                    # F(2,3) along x, which issues 4 instead of 6 multiplies per output pair: say what the matrix cores
                    # actually executed as well, so that the fraction is not mistaken for MFMA occupancy.
                    base = dom[0].split("+")[0].split(" ")[0]
                    if base.startswith("convs.") and "upconv" not in base:
                        n_conv = int(base.split(".")[1])
                        res = 4 * 2 ** ((n_conv + 1) // 2)
                        mode = g.convs[n_conv].conv.conv_mode(res, res)
                        if mode in (2, 3, 5):
                            ratio = {2: 2.0 / 3.0, 3: 0.5, 5: 1.0 / 3.0}[mode]
                            result["roofline"]["algorithm"] = {
                                2: "winograd F(2,3) along x: executed MFMA flops = 2/3 algorithmic",
                                3: "winograd F(4,3) along x: executed MFMA flops = 1/2 algorithmic (frac counts algorithmic "
                                   "direct-conv flops as SURVEY 8d defines them and can therefore exceed 1)",
                                5: "2-D winograd F(2x4,3x3): executed MFMA flops = 1/3 algorithmic (frac counts algorithmic "
                                   "direct-conv flops as SURVEY 8d defines them and can therefore exceed 1; executed_frac is what the "
                                   "matrix cores did)"}[mode]
                            result["roofline"]["executed"] = ach * ratio
                            result["roofline"]["executed_frac"] = ach * ratio / MFMA_F32_PEAK_TFLOPS
                    result["roofline"]["kernel_instance"] = inst
                    table, table_path = pmc_traffic_table()
                    rec = (table or {}).get("kernels", {}).get(inst) if inst else None
                    if rec is not None and table.get("batch") == B and table.get("size") == size and rec.get("dispatches_per_step") == 1:
                        # HBM bytes of this launch from the PMC passes of the SAME bench command (tools/profile_round.sh ->
                        # tools/make_profiles.py): FETCH_SIZE x2 (guide's gfx950 correction, calibrated on a known-size copy) +
                        # WRITE_SIZE.  Only used when that instance is launched once per step, i.e. the average IS this launch.
                        result["roofline"]["traffic"] = rec["read_bytes"] + rec["write_bytes"]
                        result["roofline"]["traffic_source"] = f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {table_path}"
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
            # standalone upfirdn2d on the 1024-res Blur shape (the op BASELINE.json's metric names)
            r_out = size
            xin = torch.randn(B, 32 if size == 1024 else 64, r_out + 1, r_out + 1, device=dev)
            kern = torch.from_numpy(seeding.fir_kernel_2d((1, 3, 3, 1), 4.0)).to(dev)
            yout = torch.empty(xin.shape[0] * xin.shape[1], r_out, r_out, 1, device=dev)
            major = xin.shape[0] * xin.shape[1]

            def launch_fir():
                lib.maua_upfirdn2d_f32(xin.data_ptr(), kern.data_ptr(), yout.data_ptr(), major, r_out + 1, r_out + 1, 1,
                                       4, 4, 1, 1, 1, 1, 1, 1, 1, 1, sp)

            ms = time_calls(launch_fir, 20, sp)
            byts = 4 * major * ((r_out + 1) ** 2 + r_out ** 2)
            ach = byts / ms / 1e6
            # HBM bytes per launch from the PMC passes of profiles/r01_pmc_upfirdn2d.md (FETCH_SIZE x2 correction,
            # WRITE_SIZE exact), scaled to this launch's plane count; measured at 256 planes.
            table, table_path = pmc_traffic_table()
            rec = (table or {}).get("kernels", {}).get("fir_strip_kernel<4, 4, 4, false, false>")
            traffic = None
            if rec is not None and table.get("size") == size and rec.get("planes"):
                traffic = (rec["read_bytes"] + rec["write_bytes"]) * major / rec["planes"]  # scaled to this launch's planes
            result["roofline_upfirdn2d"] = {"kernel": "fir_strip_kernel<4,4,4,false,false>", "bound": "hbm", "achieved": ach,
                                            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                            "traffic": traffic, "traffic_source": (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, {table_path}" if traffic else None), "launch_ms": ms, "algorithmic_bytes": byts,
                                            "shape": f"[{xin.shape[0]},{xin.shape[1]},{r_out+1},{r_out+1}] -> {r_out}^2"}
            # measured copy ceiling of this box (SURVEY.md 8d asks for "vs spec" and "vs measured copy"): a device-to-device copy of
            # the same number of bytes (read n/2, write n/2) on the same stream
            with torch.cuda.stream(torch.cuda.ExternalStream(sp, device=dev)):
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
