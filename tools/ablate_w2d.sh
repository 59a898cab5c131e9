#!/bin/bash
# Ablation of modconv_w2d_kernel on plain and fused layers (w2d-debug bits: 1 no MFMA, 2 no DMA after the first chunk, 4 no feature
# stores, 8 no epilogue, 16 one K chunk only).  LIB=<path> measures another build.
cd "$(dirname "$0")/.."
for d in ${DBG:-0 1 2 3}; do
  python tools/microbench.py conv fused --iters 20 --wino2d-min-cout 32 --w2d-debug $d ${LIB:+--lib $LIB} 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin); print('dbg=$d', {k:round(v['ms'],3) for k,v in r.items() if 'w2d' in v.get('kernel','')})"
done
