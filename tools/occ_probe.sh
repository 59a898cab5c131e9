#!/bin/bash
# Occupancy probe for modconv_w2d_kernel: with MAUA_W2D_LDS_PAD the dynamic LDS request grows past 80 KB, so a CU holds one workgroup
# instead of two.  If the two co-resident workgroups overlapped their MFMA and transform phases, one per CU would take ~2x as long.
cd "$(dirname "$0")/.."
for pad in 0 24000; do
  MAUA_W2D_LDS_PAD=$pad python tools/microbench.py conv fused --iters 20 --wino2d-min-cout 32 2>/dev/null | python -c "
import json,sys; r=json.load(sys.stdin); print('pad=$pad', {k:round(v['ms'],3) for k,v in r.items() if 'w2d' in v.get('kernel','')})"
done
