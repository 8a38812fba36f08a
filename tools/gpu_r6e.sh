#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "mlp_one_kernel or fused_mlp" 2>&1 | tail -2
for nw in 12 8; do CLV_FMLP_NW=$nw python tools/probes/mlp_fused_bench.py 2>&1 | grep "one-kernel" | sed "s/^/NW=$nw: /"; done | tee gpurun_out/r6e_mlp_bench.txt
CLV_FMLP_ABL=1 python tools/probes/mlp_fused_bench.py 2>&1 | grep "one-kernel" | sed 's/^/no-gelu: /' | tee -a gpurun_out/r6e_mlp_bench.txt
