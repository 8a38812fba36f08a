#!/bin/bash
# A/B of whole library builds on the default bench: bash tools/gpu_ab_lib.sh libA.so libB.so ... (two rounds; extra env via ENVV="K=V")
set -u
for rep in 1 2; do for l in "$@"; do
  r=$(env CLOVER_LIB_PATH=$PWD/$l ${ENVV:-X=1} python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses']['loss'], d['grad_norm'])")
  echo "$l rep$rep: $r"
done; done
