#!/bin/bash
# Per-stage time + L2-memory-side traffic of the grouped weight-gradient launches (tools/probes/wgrad_traffic.py):
# one un-profiled timed run, then one rocprofv3 --pmc pass per counter; summary -> gpurun_out/<tag>_wgrad_traffic.json
set -u
TAG=${1:-cur}
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
python3 tools/probes/wgrad_traffic.py > gpurun_out/${TAG}_wgrad_timed.txt 2>&1
cat gpurun_out/${TAG}_wgrad_timed.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/wt_$c; mkdir -p /tmp/wt_$c
  (cd /tmp && TIMED=0 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/wt_$c -o p -- \
     python3 $R/tools/probes/wgrad_traffic.py > /tmp/wt_$c/stdout.log 2>&1)
  echo "$c rc=$?"
done
python3 tools/probes/wgrad_traffic_sum.py /tmp/wt_FETCH_SIZE /tmp/wt_WRITE_SIZE /tmp/wt_FETCH_SIZE/stdout.log > gpurun_out/${TAG}_wgrad_traffic.json
cat gpurun_out/${TAG}_wgrad_traffic.json
