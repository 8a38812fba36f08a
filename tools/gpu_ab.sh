#!/bin/bash
# A/B of one environment switch on the default bench: bash tools/gpu_ab.sh VAR val1 val2 ... (two rounds each)
set -u
VAR=$1; shift
for rep in 1 2; do for v in "$@"; do
  r=$(env $VAR=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses']['loss'], d['grad_norm'])")
  echo "$VAR=$v rep$rep: $r"
done; done
