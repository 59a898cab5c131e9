#!/bin/bash
# Compile-time ablation builds of modconv_w2d_kernel: tools/ab/libmaua_abl<mask>.so for every mask given (see MAUA_W2D_ABL in
# csrc/modconv_w2d.hip).  Runtime switches inside the loop disturb the MFMA stream they are meant to measure.
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/ab
objs=$(ls maua_stylegan2_amd/csrc/*.o | grep -v modconv_w2d.o)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DMAUA_W2D_ABL=$m -c maua_stylegan2_amd/csrc/modconv_w2d.hip -o tools/ab/w2d_abl$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libmaua_abl$m.so tools/ab/w2d_abl$m.o $objs
  rm tools/ab/w2d_abl$m.o
  echo tools/ab/libmaua_abl$m.so
done
