#!/bin/bash
set -u
for g in 2048 1280 1024 2560 1536; do
  r=$(CLV_LNV_GRID=$g python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "CLV_LNV_GRID=$g: $r"
done
