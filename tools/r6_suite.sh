#!/bin/bash
# the whole GPU suite (+ smoke) on the current tree, then the same suite without the caching allocator
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6_suite
rm -rf "$O"; mkdir -p "$O"
cd "$R"; export TMPDIR=/tmp
( timeout 2700 python -m pytest tests -q -m gpu -p no:cacheprovider > "$O/pytest_gpu.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_gpu.log" )
grep -v "^E  \|^    " "$O/pytest_gpu.log" | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 2400 python -m pytest tests -q -m gpu -x -p no:cacheprovider --deselect tests/test_render_gpu.py::test_stylegan1_captured_forward_equals_eager --deselect tests/test_render_gpu.py::test_stylegan1_through_generate_and_render_vs_oracle > "$O/pytest_nocache.log" 2>&1; echo "rc=$?" >> "$O/pytest_nocache.log"
tail -4 "$O/pytest_nocache.log"
