#!/bin/bash
# Round 6, GPU call 5: isolated layer rows of the 4^2 .. 32^2 block at batch 8 / 16 / 24 / 32 (is the block latency-bound, i.e. nearly batch-independent?)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6e
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
for b in 8 16 24 32; do
  timeout 600 python bench.py --batch $b --lanes 1 --steps 2 --warmup 1 --batches-per-step 4 --no-cpu-baseline --no-side-configs --no-pcie-side > "$O/b$b.json" 2> "$O/b$b.err"; echo "rc=$?"
  python - <<PY
import json
p=json.loads(open("$O/b$b.json").read().strip().splitlines()[-1])
print("batch $b:", round(p["value"],1), "frames/s")
tot=0
for r in p["layers"]:
    n=r["name"]
    small = n.startswith(("conv1","to_rgb1","convs.0","convs.1","convs.2","convs.3","convs.4","to_rgbs.0","to_rgbs.1")) and not n.startswith(("convs.10","convs.11","convs.12","convs.13","convs.14","convs.15"))
    if small: tot+=r["ms"]
    print("  %-55s %.4f ms %s" % (n, r["ms"], "*" if small else ""))
print("  small block: %.4f ms = %.4f per 8 frames" % (tot, tot*8/$b))
PY
done
