#!/bin/bash
# What the data-parallel structure's cost at one rank consists of: plain | forced 1-rank RCCL with the 16-bit wire (pack
# kernels in the backward graphs) | forced with an fp32 wire (no pack kernels, norm / AdamW on the slabs).
one() { python bench.py --steps 40 --warmup 10 --phases --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms'])"; }
for r in 1 2; do
  echo "plain               rep$r: $(one)"
  echo "forced, 16-bit wire rep$r: $(CLOVER_FORCE_COLLECTIVES=1 one)"
  echo "forced, fp32 wire   rep$r: $(CLOVER_FORCE_COLLECTIVES=1 CLOVER_BF16_ALLREDUCE=0 one)"
done
