#!/bin/bash
# Round 5 evidence run (through gpurun): the round-end profile set of tools/profile_round.sh, then the round's extras — the one-rank RCCL
# run and the 8-rank gloo dry run with their `rccl` blocks, the sanitizer legs, the fused-layer probes and the A/B of the fused layer in bench.py.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd "$R"
bash tools/profile_round.sh > gpurun_out/profile_round_stdout.log 2>&1
O=$R/gpurun_out/prof_final
X=$R/gpurun_out/r5_extras
rm -rf "$X"; mkdir -p "$X"
COMMON="--no-cpu-baseline --no-breakdown --no-side-configs --no-pcie-side"
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --force-gather $COMMON > "$X/bench_rccl_1rank_force_gather.json" 2> "$X/bench_rccl_1rank.err"; echo "rccl-1 rc=$?" )
( MAUA_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 8 --steps 2 --warmup 1 --lanes 1 $COMMON > "$X/bench_8ranks_gloo_one_gpu.json" 2> "$X/bench_8ranks_gloo.err"; echo "gloo-8 rc=$?" )
bash tools/asan_run.sh > "$X/asan_stdout.log" 2>&1; cp -r gpurun_out/asan "$X/asan" 2>/dev/null; rm -rf gpurun_out/asan
( timeout 300 python tools/fuse_probe.py --lib maua_stylegan2_amd/csrc/libmaua_hip.so --exact --rounds 3 > "$X/fuse_probe_exact.json" 2> "$X/fuse_probe_exact.err"; echo "fuse exact rc=$?" )
# (the segment-length override only exists in the experiments build)
( timeout 300 python tools/fuse_probe.py --lib tools/bin/libmaua_fuse.so --exact --rounds 5 --sweep 9,10,4,5,6,8,3,16 > "$X/fuse_probe_segment_sweep.json" 2> "$X/fuse_probe_segment_sweep.err"; echo "fuse sweep rc=$?" )
( timeout 200 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 tools/gather_probe.py 2>/dev/null | grep "^{" > "$X/gather_probe.jsonl"; echo "gather probe rc=$?" )
( timeout 300 python tools/fuse_probe.py --lib tools/bin/libmaua_fuse.so --rounds 3 > "$X/fuse_probe_halo_free.json" 2> "$X/fuse_probe_halo_free.err"; echo "fuse halo-free rc=$?" )
ROUNDS=3 WIDTHS="256 512 100000" bash tools/ab_fused.sh > "$X/bench_ab_fused_layer.txt" 2>&1
( timeout 200 python tools/microbench.py fir > "$X/microbench_fir.json" 2> "$X/microbench_fir.err"; echo "microbench fir rc=$?" )
cat "$X/bench_ab_fused_layer.txt"
python - <<PY
import json
for f in ("bench_rccl_1rank_force_gather", "bench_8ranks_gloo_one_gpu"):
    try:
        d = json.loads(open("$X/%s.json" % f).read().strip().splitlines()[-1]); r = d.get("rccl", {})
        print(f, "value", round(d["value"], 1), "n_gpus", d["n_gpus"], "frame_check", d["frame_check"]["max_abs_grey_level_diff_graph_vs_eager"], "| rccl:", r.get("backend"),
              r.get("rccl_version"), "world", r.get("world_size"), "weights_ok", r.get("weights", {}).get("param_checksums_equal_after_broadcast"), "payload", r.get("frames", {}).get("payload_check"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
grep -A3 "torch-free driver" "$X/asan/summary.txt" | head -12; tail -3 "$O/pytest_gpu.log"; head -c 300 "$O/bench_default.json"; echo; du -sh "$O" "$X"
