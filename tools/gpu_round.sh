#!/bin/bash
# One GPU-box pass: parity tests, default bench line, rocprof kernel stats of the same command.
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" 
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
cat gpurun_out/bench_n1.json
export TMPDIR=/tmp
R=$PWD
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err; echo "rocprof rc=$?"
cd $R
find gpurun_out/prof -name '*kernel_stats.csv' | head
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && head -12 "$f" | cut -c1-160
find gpurun_out/prof -name '*kernel_trace.csv' -delete
find gpurun_out/prof -name '*.db' -delete
cat gpurun_out/bench_prof.json | cut -c1-600
