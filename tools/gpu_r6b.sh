#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py -m gpu -q -s -k "mid or bench_shapes or gemm_nt_epilogues or linear_and_mlp_on_the_hip or full_size" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/r6b_pytest.log; echo "pytest rc=$?"
grep -n "loss errors\|grad rel errors\|passed\|failed\|FAILED\|ERROR" gpurun_out/r6b_pytest.log | cut -c1-1500
bash tools/ab_trees.sh 3 2>&1 | tee gpurun_out/r6b_ab.txt
