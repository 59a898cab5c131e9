#!/bin/bash
# Shader clock and package power while bench.py runs with 1 and with 3 graph lanes (rocm-smi sampled every 0.25 s): tools/clock_probe.sh <outdir>
out=$1; mkdir -p $out
for lanes in 1 3; do
  python bench.py --steps 80 --warmup 3 --lanes $lanes --no-cpu-baseline --no-side-configs --no-breakdown --no-pcie-side > $out/bench_l$lanes.json 2> $out/bench_l$lanes.err &
  pid=$!
  sleep 2
  : > $out/smi_l$lanes.txt
  while kill -0 $pid 2>/dev/null; do
    /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Average Graphics Package Power\|Socket" >> $out/smi_l$lanes.txt
    sleep 0.25
  done
  wait $pid
  python - <<PY
import json,re
p=json.loads(open("$out/bench_l$lanes.json").read().strip().splitlines()[-1])
t=open("$out/smi_l$lanes.txt").read()
clk=[int(x) for x in re.findall(r"\((\d+)Mhz\)", t)]
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", t)]
busy=[c for c, w in zip(clk, pw) if w > 900]; bw=[w for w in pw if w > 900]   # samples taken while the timed region ran
s=lambda a: (min(a), sorted(a)[len(a)//2], max(a)) if a else None
print("lanes $lanes: %.1f frames/s; under load (%d samples): sclk min/median/max %s MHz, socket power %s W" % (p["value"], len(busy), s(busy), s(bw)))
PY
done
tail -4 $out/smi_l3.txt
