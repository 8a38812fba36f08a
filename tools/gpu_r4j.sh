#!/bin/bash
set -u
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "rccl or finetune_engine" 2>&1 | tail -4
for rep in 1 2; do
  for cfg in "0 1" "1 0" "1 1"; do set -- $cfg
  r=$(CLOVER_FORCE_COLLECTIVES=$1 CLOVER_PACK_IN_GRAPH=$2 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('exposed_comm_ms'))")
  echo "FORCE_COLLECTIVES=$1 PACK_IN_GRAPH=$2 rep$rep: $r"
done; done
