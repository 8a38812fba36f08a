#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nt" > gpurun_out/r4b_t1.log 2>&1; echo "t1 rc=$?"; tail -3 gpurun_out/r4b_t1.log
timeout 1200 python tools/probes/gemm_sweep.py > gpurun_out/r4b_sweep.log 2>&1; cat gpurun_out/r4b_sweep.log
