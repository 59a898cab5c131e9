#!/bin/bash
# Times every tools/ab/libmaua_abl*.so (and the in-tree library) on the w2d layers; PAD=<bytes> limits a CU to one workgroup
cd "$(dirname "$0")/.."
for lib in maua_stylegan2_amd/csrc/libmaua_hip.so $(ls tools/ab/libmaua_abl*.so | sort -V); do
  MAUA_W2D_LDS_PAD=${PAD:-0} python tools/microbench.py conv fused --iters 20 --wino2d-min-cout 32 --lib $lib 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin); print('$(basename $lib .so)'.ljust(16), {k:round(v['ms'],3) for k,v in r.items() if 'w2d' in v.get('kernel','')})"
done
