#!/bin/bash
# rocprofv3 kernel stats of the default bench -> gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-cur}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out; rm -rf /tmp/ks; mkdir -p /tmp/ks
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 50 --warmup 10 ${BENCH_ARGS:-} > $R/gpurun_out/${TAG}_stats_bench.json 2> /tmp/ks/err.log)
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/${TAG}_stats_bench.json | cut -c1-200
