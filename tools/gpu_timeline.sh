#!/bin/bash
# Kernel trace of a short default bench run -> tools/probes/timeline.py summary (busy / idle / alone / overlapped per kernel).
set -u
TAG=${1:-r04}
R=$PWD
export TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/tl; mkdir -p /tmp/tl
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- python3 $R/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS:-} > /tmp/tl/bench.json 2> /tmp/tl/bench.err); echo "rocprof rc=$?"
f=$(find /tmp/tl -name '*kernel_trace.csv' | head -1)
python3 tools/probes/timeline.py "$f" optim_prep_kernel gpurun_out/${TAG}_timeline_step.txt > gpurun_out/${TAG}_timeline.txt 2>&1
head -75 gpurun_out/${TAG}_timeline.txt | cut -c1-130
