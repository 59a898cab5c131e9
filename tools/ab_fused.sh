#!/bin/bash
# A/B of the fused up-sampling layer inside bench.py: fused_blur_min_width in $WIDTHS (100000 = never), $ROUNDS alternating rounds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in $(seq 1 ${ROUNDS:-3}); do for w in ${WIDTHS:-512 256 100000}; do
  python bench.py --steps ${STEPS:-12} --no-cpu-baseline --no-side-configs --no-pcie-side --no-breakdown --fused-blur-min-width $w 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused_blur_min_width $w round $r:', round(d['value'],1), 'frames/s', round(d['ms_per_batch'],4), 'ms/batch, frame_check', d['frame_check']['max_abs_grey_level_diff_graph_vs_eager'])"
done; done
