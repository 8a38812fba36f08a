#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_one_kernel or fused_mlp or fused_ln_linear" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6c_pytest.log; echo "pytest rc=$?"
tail -30 gpurun_out/r6c_pytest.log | cut -c1-300
