#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_one_kernel or fused_mlp" 2>&1 | tail -2
python tools/probes/mlp_fused_bench.py 2>&1 | grep "one-kernel\|round-5" | tee gpurun_out/r6e_mlp_bench.txt
CLV_FMLP_ABL=2 python tools/probes/mlp_fused_bench.py 2>&1 | grep "one-kernel" | sed 's/^/no act,dpre stores: /' | tee -a gpurun_out/r6e_mlp_bench.txt
