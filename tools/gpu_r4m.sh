#!/bin/bash
set -u
CLV_WGRAD_WS=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad or first_touch or linear" 2>&1 | tail -3
for rep in 1 2; do for v in 0 1; do
  r=$(CLV_WGRAD_WS=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "WGRAD_WS=$v rep$rep: $r"
done; done
