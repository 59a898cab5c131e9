#!/bin/bash
# Round 6, GPU call 2: the canary (red-zone) suite, the new host-side tests, e2e timing of config 3.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6b
rm -rf "$O"; mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_canary_gpu.py -q -m gpu -p no:cacheprovider > "$O/pytest_canary.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_canary.log" )
grep -n "passed\|failed\|rc=\|^FAILED\|^E  " "$O/pytest_canary.log" | head -60
( timeout 900 python -m pytest tests/test_signal_gpu.py "tests/test_render_gpu.py::test_generator_is_kept_across_generate_calls_and_follows_the_checkpoint_file" "tests/test_render_gpu.py::test_generate_end_to_end_default_plugin" "tests/test_render_gpu.py::test_config3_900_frames_through_generate_vs_oracle" -q -m gpu -p no:cacheprovider > "$O/pytest_misc.log" 2>&1; echo "pytest rc=$?" >> "$O/pytest_misc.log" )
tail -15 "$O/pytest_misc.log"
( timeout 600 python tools/e2e_config3.py --repeat 5 > "$O/e2e_config3.txt" 2>&1; echo "e2e rc=$?" )
grep -n "E2E run\|preprocessing took\|rendered" "$O/e2e_config3.txt"
