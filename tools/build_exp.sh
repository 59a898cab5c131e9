#!/bin/bash
# Experimental variant of libmaua_hip.so: modconv.hip compiled with extra -D flags, other objects reused.
# usage: tools/build_exp.sh <name> <flags...>   -> tools/bin/libmaua_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c maua_stylegan2_amd/csrc/modconv.hip -o tools/bin/modconv_$name.o
objs=$(ls maua_stylegan2_amd/csrc/*.o | grep -v modconv.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libmaua_$name.so tools/bin/modconv_$name.o $objs
echo tools/bin/libmaua_$name.so
