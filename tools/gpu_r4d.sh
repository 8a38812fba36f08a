#!/bin/bash
set -u
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "0 0" "1 0" "2 0" "2 1" "3 0"; do
  set -- $cfg
  r=$(CLV_GEMM_WS=$1 CLV_GEMM_SPLITK=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "WS=$1 SPLITK=$2 rep$rep: $r"
done; done 2>&1 | tee gpurun_out/r4d_ab.log
