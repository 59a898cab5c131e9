#!/bin/bash
# Round 5, GPU call 1: the new multi-rank / known-answer tests, the one-rank RCCL run with the rccl evidence block, the 8-rank gloo dry run on
# the shared device, the sanitizer legs, and a short default bench (sanity of the bench refactor + reference numbers for later A/B).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5a
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_world8_gpu.py tests/test_signal_gpu.py tests/test_two_rank_gpu.py "tests/test_render_gpu.py::test_render_rank_shards_on_device_equal_single_rank" "tests/test_render_gpu.py::test_whole_workload_wav_to_frames_vs_oracle" -q -m gpu -s -p no:cacheprovider > "$O/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_new.log" )
tail -3 "$O/pytest_new.log"
COMMON="--no-cpu-baseline --no-breakdown --no-side-configs --no-pcie-side"
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 1 --force-gather $COMMON > "$O/bench_rccl_1rank_force_gather.json" 2> "$O/bench_rccl_1rank.err"; echo "rccl-1 rc=$?" )
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 tools/rccl_probe.py > "$O/rccl_probe.log" 2>&1; echo "probe rc=$?" )
( MAUA_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29515 bench.py --gpus 8 --steps 2 --warmup 1 --lanes 1 $COMMON > "$O/bench_8ranks_gloo_one_gpu.json" 2> "$O/bench_8ranks_gloo.err"; echo "gloo-8 rc=$?" )
bash tools/asan_run.sh > "$O/asan_stdout.log" 2>&1; cp -r gpurun_out/asan "$O/asan" 2>/dev/null; rm -rf gpurun_out/asan
( timeout 600 python bench.py --steps 10 --no-cpu-baseline > "$O/bench_short.json" 2> "$O/bench_short.err"; echo "bench rc=$?" )
head -c 400 "$O/bench_short.json"; echo
( timeout 600 python tools/fuse_probe.py --lib tools/bin/libmaua_fuse.so > "$O/fuse_probe.json" 2> "$O/fuse_probe.err"; echo "fuse rc=$?"; cat "$O/fuse_probe.json"; tail -3 "$O/fuse_probe.err" )
python - <<PY
import json
for f in ("bench_rccl_1rank_force_gather", "bench_8ranks_gloo_one_gpu"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        r = d.get("rccl", {})
        print(f, "value", round(d["value"], 1), "n_gpus", d["n_gpus"], "frame_check", d["frame_check"]["max_abs_grey_level_diff_graph_vs_eager"],
              "| rccl:", r.get("backend"), r.get("rccl_version"), "world", r.get("world_size"), "devices", r.get("distinct_devices"),
              "weights_ok", r.get("weights", {}).get("param_checksums_equal_after_broadcast"), "payload", r.get("frames", {}).get("payload_check"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
tail -30 "$O/asan/summary.txt" 2>/dev/null
# one wave per SIMD for the >= 64-channel 2-D Winograd kernel (VERDICT r4 item 4 i): parity of the variant build, then alternating A/B
( MAUA_TEST_LIB=tools/bin/libmaua_w1.so timeout 600 python -m pytest tests/test_layers_gpu.py -q -m gpu -k "winograd2d or torgb or styled" -p no:cacheprovider 2>&1 | tail -3 ) > "$O/w1_parity.log" 2>&1; tail -2 "$O/w1_parity.log"
for r in 1 2 3; do
  for v in fuse w1; do
    timeout 300 python tools/microbench.py conv --lib tools/bin/libmaua_$v.so --iters 20 > "$O/w1_ab_${v}_$r.json" 2> "$O/w1_ab_${v}_$r.err" || echo "microbench $v $r failed"
  done
done
python - <<PY
import json, glob
for v in ("fuse", "w1"):
    rows = {}
    for f in sorted(glob.glob("$O/w1_ab_%s_*.json" % v)):
        try:
            d = json.loads(open(f).read().strip().splitlines()[-1])
        except Exception as e:
            print(f, "unreadable", e); continue
        for k, r in d.items():
            rows.setdefault(k, []).append(round(r["ms"], 4))
    print(v, {k: (min(x), r) for k, x in rows.items() for r in [x]})
PY
du -sh "$O"
