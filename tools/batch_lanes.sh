mkdir -p gpurun_out/bl
for cfg in "8 3" "16 2" "16 3" "12 3" "8 4" "4 4"; do
  set -- $cfg
  python bench.py --batch $1 --lanes $2 --steps 10 --warmup 2 --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > gpurun_out/bl/b$1_l$2.json 2> gpurun_out/bl/b$1_l$2.err
  python - <<PY
import json
try:
    p=json.loads(open("gpurun_out/bl/b$1_l$2.json").read().strip().splitlines()[-1])
    print("batch $1 lanes $2:", round(p["value"],1), "frames/s", p.get("frame_check"))
except Exception as e:
    print("batch $1 lanes $2: failed", e)
PY
done
