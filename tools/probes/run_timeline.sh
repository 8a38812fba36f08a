#!/bin/bash
# usage: run_timeline.sh [bench args] -- kernel-trace bench.py, write gpurun_out/step_timeline.txt
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/trace; mkdir -p $R/gpurun_out/trace
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -o t -- python3 $R/bench.py "$@" > $R/gpurun_out/trace/stdout.log 2>&1
cd $R
tail -1 gpurun_out/trace/stdout.log | cut -c1-300
f=$(find gpurun_out/trace -name '*kernel_trace.csv' | head -1)
python tools/probes/step_timeline.py $f gpurun_out/step_timeline.txt
python tools/probes/steady_trace.py $f 8 70 > gpurun_out/steady.txt
head -50 gpurun_out/step_timeline.txt
rm -f $f
