#!/bin/bash
set -u
for rep in 1 2; do for v in 0 1; do
  r=$(CLOVER_FC2_OWN=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
  echo "FC2_OWN=$v rep$rep: $r"
done; done
