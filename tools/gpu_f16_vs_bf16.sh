#!/bin/bash
# Same-box step time of the two builds and of the loss scaler's three modes (phase clock): bash tools/gpu_f16_vs_bf16.sh [reps]
REPS=${1:-2}
one() { python bench.py --steps 40 --warmup 10 --phases --no-cpu-baseline --no-kernel-timing "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'])"; }
for r in $(seq $REPS); do
  echo "bf16                rep$r: $(one --dtype bf16)"
  echo "f16 dynamic scaler  rep$r: $(one --dtype f16)"
  echo "f16 static 1024     rep$r: $(one --dtype f16 --loss-scale 1024)"
  echo "f16 no scaler       rep$r: $(one --dtype f16 --loss-scale none)"
  echo "bf16 dynamic scaler rep$r: $(one --dtype bf16 --loss-scale dynamic)"
done
