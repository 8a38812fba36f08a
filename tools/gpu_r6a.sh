#!/bin/bash
# Round-6 pass: full GPU test suite (all failures listed), then same-box A/B against the round-5 tree.
set -u
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --deselect tests/test_kernels_gpu.py::test_grouped_weight_gradients_shape_fitted_tiles 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6a_pytest.log; echo "pytest rc=$?"
grep -n "passed\|failed\|^FAILED\|^ERROR" gpurun_out/r6a_pytest.log | cut -c1-300
bash tools/ab_trees.sh ${1:-3} 2>&1 | tee gpurun_out/r6a_ab.txt
