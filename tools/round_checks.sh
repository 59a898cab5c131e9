#!/bin/bash
# Round-end robustness / end-to-end checks on the GPU box (through gpurun): generate() end to end (config 3, counting sink), the gathered
# region of bench.py on one GPU, the determinism / NaN-poison soak of every conv mode, and the GPU suite with the caching allocator off
# (freed memory is really released: a dangling pointer inside a captured graph faults).  Outputs under gpurun_out/r4j/.
mkdir -p gpurun_out/r4j
python tools/e2e_config3.py --repeat 3 --stage-times > gpurun_out/r4j/e2e_config3.txt 2>&1; grep "rendered\|E2E\|preprocessing" gpurun_out/r4j/e2e_config3.txt
python bench.py --steps 12 --no-cpu-baseline --no-side-configs --no-breakdown --force-gather > gpurun_out/r4j/bench_force_gather.json 2> gpurun_out/r4j/bench_force_gather.err
python -c "
import json;p=json.load(open('gpurun_out/r4j/bench_force_gather.json'));print('force-gather', p['value'], 'synth-only', p['frames_per_sec_synth_only'])"
python tools/soak_conv.py > gpurun_out/r4j/soak.txt 2>&1; tail -3 gpurun_out/r4j/soak.txt
PYTORCH_NO_CUDA_MEMORY_CACHING=1 python -m pytest tests -q -m gpu -x --deselect tests/test_render_gpu.py::test_stylegan1_captured_forward_equals_eager --deselect tests/test_render_gpu.py::test_stylegan1_through_generate_and_render_vs_oracle > gpurun_out/r4j/pytest_nocache.log 2>&1; tail -2 gpurun_out/r4j/pytest_nocache.log
