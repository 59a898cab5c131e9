#!/bin/bash
# Round 6, final tree: determinism soak (tools/soak_conv.py) + the GPU suite without the caching allocator
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_soak
rm -rf "$O"; mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
timeout 1200 python tools/soak_conv.py > "$O/soak.txt" 2>&1; echo "soak rc=$?" >> "$O/soak.txt"
tail -6 "$O/soak.txt"
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 2400 python -m pytest tests -q -m gpu -x -p no:cacheprovider --deselect tests/test_render_gpu.py::test_stylegan1_captured_forward_equals_eager --deselect tests/test_render_gpu.py::test_stylegan1_through_generate_and_render_vs_oracle > "$O/pytest_nocache.log" 2>&1; echo "rc=$?" >> "$O/pytest_nocache.log"
tail -4 "$O/pytest_nocache.log"
