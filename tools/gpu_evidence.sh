#!/bin/bash
# Round evidence pass on one MI355X box: full GPU test suite, default bench + rocprof stats + PMC passes, the GEMM sweep with
# cold weights, and the GEMM lab's stage anatomy.   CLOVER_COMMIT=<sha> bash tools/gpu_evidence.sh <tag>
set -u
TAG=${1:-r04}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -5 | tee gpurun_out/${TAG}_pytest_gpu.txt
bash tools/gpu_profile.sh $TAG
timeout 900 python tools/probes/gemm_sweep.py coldw > gpurun_out/${TAG}_gemm_sweep_coldw.txt 2>&1; tail -40 gpurun_out/${TAG}_gemm_sweep_coldw.txt | cut -c1-160
{ echo "=== default planner"; timeout 300 tools/probes/bin/gemm_lab 1
  for t in ws128c8 ws64p2 64x128w4; do echo "=== CLV_GEMM_TILE=$t"; CLV_GEMM_TILE=$t timeout 300 tools/probes/bin/gemm_lab 1; done; } > gpurun_out/${TAG}_gemm_lab_stage_anatomy.txt 2>&1
