#!/bin/bash
# usage: run_traced.sh <pattern> <python script> [args]  -- kernel-trace a probe and print per-grid medians
pat=$1; shift
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o t -- python3 $R/"$@" > $R/gpurun_out/trace/stdout.log 2>&1
cd $R
f=$(find gpurun_out/trace -name '*kernel_trace.csv' | head -1)
python tools/probes/trace_summary.py $f "$pat" ${TRACE_AGG:+--agg}
rm -f $f
