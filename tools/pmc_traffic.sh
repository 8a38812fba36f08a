#!/bin/bash
# HBM traffic of the step's kernels from the L2 memory-side counters: one rocprofv3 --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, eager step so every kernel is a dispatch.
set -u
export TMPDIR=/tmp
R=$PWD
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c; mkdir -p $R/gpurun_out/pmc_$c
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- \
     python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS:-} > $R/gpurun_out/pmc_$c/stdout.log 2>&1)
  echo "$c rc=$?"
  find gpurun_out/pmc_$c -name '*kernel_trace.csv' -delete
done
python3 tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
head -c 1500 gpurun_out/pmc_traffic.json
find gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE -name '*counter_collection.csv' -size +20M -delete
