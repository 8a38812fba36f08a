#!/bin/bash
# Round evidence pass on one MI355X box: full GPU test suite, default bench + rocprof stats + PMC passes (tools/gpu_profile.sh),
# the per-stage weight-gradient traffic probe.   CLOVER_COMMIT=<sha> bash tools/gpu_evidence.sh <tag>
set -u
TAG=${1:-r05}
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -8 | tee gpurun_out/${TAG}_pytest_gpu.txt
bash tools/gpu_profile.sh $TAG
