#!/bin/bash
# Same-box A/B of an environment switch on the default bench (pairs/s): tools/ab_bench.sh VAR A_VALUE B_VALUE [extra bench args]
VAR=$1; A=$2; B=$3; shift 3
for rep in $(seq 1 ${ABREPS:-2}); do
  for v in "$A" "$B"; do
    r=$(env $VAR=$v python bench.py --steps ${ABSTEPS:-30} --warmup 10 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")
    echo "$VAR=$v rep$rep: $r"
  done
done
