#!/bin/bash
# Round 6, last evidence call: sanitizer legs, the bench of record with the round's own PMC traffic table, the self-launched multi-rank runs
# (one nccl rank with the gathered region; eight gloo ranks sharing the device), end-to-end config 3.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_final
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
bash tools/asan_run.sh > "$O/asan_stdout.log" 2>&1; cp gpurun_out/asan/summary.txt "$O/asan_summary.txt" 2>/dev/null; cp gpurun_out/asan/host.log "$O/asan_host.log" 2>/dev/null; rm -rf gpurun_out/asan
head -12 "$O/asan_summary.txt"; grep -c " ok$" "$O/asan_summary.txt"
( timeout 900 python bench.py --steps 20 --warmup 5 > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$?" )
COMMON="--no-cpu-baseline --no-breakdown --no-side-configs --no-pcie-side"
( timeout 600 python bench.py --gpus 1 --self-launch --force-gather --steps 5 --warmup 1 $COMMON > "$O/bench_rccl_1rank_self_launched.json" 2> "$O/bench_rccl_1rank.err"; echo "rccl-1 rc=$?" )
( MAUA_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 2 --warmup 1 --lanes 1 $COMMON > "$O/bench_8ranks_gloo_self_launched.json" 2> "$O/bench_8ranks_gloo.err"; echo "gloo-8 rc=$?" )
( timeout 600 python tools/e2e_config3.py --repeat 6 > "$O/e2e_config3.txt" 2>&1; echo "e2e rc=$?" )
grep "E2E run\|preprocessing took\|rendered" "$O/e2e_config3.txt"
python - <<PY
import json
for f in ("bench_default", "bench_rccl_1rank_self_launched", "bench_8ranks_gloo_self_launched"):
    try:
        lines = [l for l in open("$O/%s.json" % f).read().strip().splitlines() if l.startswith("{")]
        d = json.loads(lines[-1])
        r = d.get("rccl", {})
        print(f, "value", round(d["value"], 1), "n_gpus", d["n_gpus"], "frame_check", d["frame_check"]["max_abs_grey_level_diff_graph_vs_eager"],
              "| roofline", (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("traffic"),
              "| rccl:", r.get("backend"), "world", r.get("world_size"), "weights_ok", r.get("weights", {}).get("param_checksums_equal_after_broadcast"),
              "payload", r.get("frames", {}).get("payload_check"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
