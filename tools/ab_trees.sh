#!/bin/bash
# Same-box A/B of the default bench between this tree and a copy of an older one (tools/probes/bin/old_tree, built in the
# build container from the round-start commit): bash tools/ab_trees.sh [reps] [bench args]
set -u
REPS=${1:-2}; shift || true
for rep in $(seq $REPS); do for t in old new; do
  if [ $t = old ]; then B=tools/probes/bin/old_tree/bench.py; else B=bench.py; fi
  r=$(python $B --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses']['loss'], d['grad_norm'])")
  echo "$t rep$rep: $r"
done; done
