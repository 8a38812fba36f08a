#!/bin/bash
# GPU time per phase of the graphed step (events at the phase boundaries), loss section eager and as a hipGraph.
set -u
for lg in 0 1; do
  CLOVER_LOSS_GRAPH=$lg python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing --phases ${BENCH_ARGS:-} 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('LOSS_GRAPH=$lg', d['value'], d['ms_per_step'], d['phases_ms'])"
done
