#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1200 python tools/probes/gemm_sweep.py quick coldw > gpurun_out/r4c_sweep_cold.log 2>&1; cat gpurun_out/r4c_sweep_cold.log
