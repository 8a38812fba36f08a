#!/bin/bash
set -u
mkdir -p gpurun_out
CLOVER_HALF=f16 timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q --deselect tests/test_kernels_gpu.py::test_grouped_weight_gradients_shape_fitted_tiles 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6k_f16_kernels.log; echo "kernels rc=$?"
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r6k_f16_kernels.log | cut -c1-200 | tail -40
CLOVER_HALF=f16 timeout 1500 python -m pytest tests/test_step_gpu.py -m gpu -q -s -k "full_size or mid_ or step_losses_and_grads" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6k_f16_step.log; echo "step rc=$?"
grep -n "loss errors\|passed\|failed\|^FAILED\|^ERROR\|Error" gpurun_out/r6k_f16_step.log | cut -c1-600 | tail -30
for d in bf16 f16; do python bench.py --dtype $d --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'], d['losses'], d['grad_norm'], d['dtype'])"; done | tee gpurun_out/r6k_bench.txt
