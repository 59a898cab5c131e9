#!/bin/bash
# Builds the torch-free sanitizer drivers (tools/asan_driver.cpp) against the sanitized libraries of
# `python -m maua_stylegan2_amd.build --asan-host` / `--asan`:  tools/bin/asan_driver_host, tools/bin/asan_driver_device.
# hipcc cross-compiles them here; they travel with the gpurun snapshot and are run by tools/asan_run.sh (legs 3 / 4).
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/bin
SAN=maua_stylegan2_amd/csrc/san
RP='-Wl,-rpath,$ORIGIN/../../maua_stylegan2_amd/csrc/san'
gcc -O1 -g -fPIC -c oracle/c/ops_ref.c -o tools/bin/ops_ref_driver.o
HOSTF="--offload-arch=gfx950 -fsanitize=address -shared-libasan -fno-gpu-sanitize -fsanitize=undefined -fno-sanitize=vptr"
DEVF="--offload-arch=gfx950:xnack+ -fsanitize=address -shared-libasan"
/opt/rocm/bin/hipcc $HOSTF -O1 -g -std=c++17 -c tools/asan_driver.cpp -o tools/bin/asan_driver_host.o
/opt/rocm/bin/hipcc $HOSTF tools/bin/asan_driver_host.o tools/bin/ops_ref_driver.o -L$SAN -lmaua_hip_hostasan $RP -o tools/bin/asan_driver_host
if [ -f $SAN/libmaua_hip_asan.so ]; then  # (device ASAN: gfx950:xnack+ — the GPU pool refuses such runs since round 6; built only where it can run)
  /opt/rocm/bin/hipcc $DEVF -O1 -g -std=c++17 -c tools/asan_driver.cpp -o tools/bin/asan_driver_device.o
  /opt/rocm/bin/hipcc $DEVF tools/bin/asan_driver_device.o tools/bin/ops_ref_driver.o -L$SAN -lmaua_hip_asan $RP -o tools/bin/asan_driver_device
fi
rm -f tools/bin/ops_ref_driver.o tools/bin/asan_driver_host.o tools/bin/asan_driver_device.o
ls -la tools/bin/asan_driver_host*
