#!/bin/bash
# Experiments build of libmaua_hip.so: every source compiled with -DMAUA_EXPERIMENTS (ablation masks, tile-shape switches, the
# DBG kernel instantiations and the `maua_tuning_set` entry, none of which exist in the product library) plus any extra -D flags.
# usage: tools/build_exp.sh <name> [flags...]   -> tools/bin/libmaua_<name>.so   (use with bench.py --lib / tools/*.py --lib)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin/$name
objs=""
for src in maua_stylegan2_amd/csrc/*.hip; do
  o=tools/bin/$name/$(basename ${src%.hip}).o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMAUA_EXPERIMENTS "$@" -c $src -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libmaua_$name.so $objs
echo tools/bin/libmaua_$name.so
