#!/bin/bash
# Same-box cost of the data-parallel structure at one rank: the default step against CLOVER_FORCE_COLLECTIVES=1 (1-rank RCCL
# group: the four-graph cut, the packed wire copies, the bucket all-reduces), with the phase clock.  bash tools/gpu_dp_cost.sh [reps]
REPS=${1:-2}
one() { python bench.py --steps 40 --warmup 10 --phases --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'], 'exposed', d['exposed_comm_ms'])"; }
for r in $(seq $REPS); do
  echo "plain  rep$r: $(one)"
  echo "forced rep$r: $(CLOVER_FORCE_COLLECTIVES=1 one)"
done
