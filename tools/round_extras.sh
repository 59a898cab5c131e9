#!/bin/bash
# Round-end extras on the GPU box (through gpurun): the measured maxima of the whole-frame parity tests (pytest -s), bench.py with its side
# configurations, PMC of the split-bf16 side kernel.  Outputs under gpurun_out/r4i/.
mkdir -p gpurun_out/r4i
python -m pytest tests/test_render_gpu.py -q -m gpu -s -k "bench_configuration or config5_bends_1024 or config3_900 or captured_bends_equal" 2>&1 | grep "^\[\|passed\|failed" > gpurun_out/r4i/parity_full_frames.txt
python -m pytest tests/test_layers_gpu.py -q -m gpu -s -k "split_bf16" 2>&1 | grep "^\[\|^\.\[\|passed\|failed" >> gpurun_out/r4i/parity_full_frames.txt
cat gpurun_out/r4i/parity_full_frames.txt
python bench.py --steps 20 --no-cpu-baseline > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err
python -c "
import json;p=json.load(open('gpurun_out/r4i/bench.json'));print(p['value']);print(json.dumps(p['side_configs'],indent=1)[:3000]);print(p['roofline'].get('traffic'))"
bash tools/pmc_microbench.sh gpurun_out/r4i/pmc_sb sbf16 --split-bf16-min-cout 128 | grep "sbf16_kernel" | grep "GRBM\|MFMA_BUSY\|WAIT_ANY\|WAVE_CYCLES\|WAIT_INST_ANY"
