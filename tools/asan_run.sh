#!/bin/bash
# Sanitizer runs of the native library on the GPU box.  Builds come from
# `python -m maua_stylegan2_amd.build --asan-host` / `--asan` (hipcc cross-compiles them without a GPU; they travel with the snapshot).
#   leg 1  host ASAN + UBSan (launchers, argument checks, host-side tables; device code uninstrumented): the ops / property suites with the
#          clang ASAN runtime pre-loaded into python and MAUA_TEST_LIB pointing at the build (tests/conftest.py).
#   (device ASAN — gfx950:xnack+ code objects — ran in round 5 on the FIR / bias-act / tail kernels, profiles/r05_asan.txt; the GPU pool
#   refuses XNACK-on runs since round 6, so those legs are gone from this script.  The matrix-core kernels, which device ASAN refused to launch
#   anyway, are covered by the red-zone suite tests/test_canary_gpu.py.)
# Output: gpurun_out/asan/{host,device}.log + summary.txt
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/asan
mkdir -p "$O"
cd "$R"
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1:protect_shadow_gap=0:abort_on_error=0:halt_on_error=0:detect_odr_violation=0:log_path=$O/asan_report
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=0:log_path=$O/ubsan_report
TESTS="tests/test_ops_gpu.py tests/test_property_gpu.py"
{
  echo "== leg 1: host ASAN + UBSan build, $TESTS"
  LD_PRELOAD=$RT MAUA_TEST_LIB=maua_stylegan2_amd/csrc/san/libmaua_hip_hostasan.so timeout 900 python -m pytest $TESTS -q -m gpu -x -p no:cacheprovider 2>&1 | tail -15
  echo "rc=${PIPESTATUS[0]}"
} > "$O/host.log" 2>&1
# leg 3: the torch-free driver (tools/asan_driver.cpp: the two native ops + the fused blur tail on tile-edge shapes against the C oracle),
# linked against the host-sanitized library
RTDIR=$(dirname "$RT")
{
  echo "== leg 3: tools/bin/asan_driver_host (host ASAN + UBSan library, no python in the process)"
  LD_LIBRARY_PATH=$RTDIR:${LD_LIBRARY_PATH:-} timeout 300 tools/bin/asan_driver_host 2>&1 | tail -60
  echo "rc=${PIPESTATUS[0]}"
} > "$O/driver_host.log" 2>&1
{
  echo "sanitizer runs on $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null), $(date -u +%FT%TZ)"
  echo "--- host ASAN + UBSan"; tail -4 "$O/host.log"
  echo "--- torch-free driver, host ASAN + UBSan library"; tail -50 "$O/driver_host.log"
  echo "--- sanitizer report files:"; ls "$O" | grep -c "_report" ; for f in "$O"/*_report*; do [ -f "$f" ] && { echo "## $f"; head -40 "$f"; }; done
} > "$O/summary.txt" 2>&1
cat "$O/summary.txt" | head -120
