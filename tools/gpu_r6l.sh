#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -s --deselect tests/test_kernels_gpu.py::test_grouped_weight_gradients_shape_fitted_tiles 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6l_pytest.log; echo "pytest rc=$?"
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r6l_pytest.log | cut -c1-300 | tail -30
grep -n "loss errors" gpurun_out/r6l_pytest.log | cut -c1-420
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses'], d['grad_norm'], d['dtype'])" | tee gpurun_out/r6l_bench.txt
