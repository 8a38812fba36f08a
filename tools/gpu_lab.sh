#!/bin/bash
set -u
mkdir -p gpurun_out
for cfg in "64x128w4 1" "64x128w4 0" "ws64p2 1" "ws64p2 0" "ws64p2 4" "ws128c8 1" "ws128c8 2" "ws128c8 0"; do
  set -- $cfg
  echo "=== CLV_GEMM_TILE=$1 SPLITK=$2"
  CLV_GEMM_TILE=$1 CLV_GEMM_SPLITK=$2 timeout 300 tools/probes/bin/gemm_lab 1 | cut -c1-62
done 2>&1 | tee gpurun_out/gemm_lab.log
