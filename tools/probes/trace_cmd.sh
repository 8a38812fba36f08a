export TMPDIR=/tmp; R=$PWD; rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1)
f=$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)
python tools/probes/steady_trace.py $f 8 70 > gpurun_out/steady_r03b.txt 2>&1; python tools/probes/step_sequence.py $f > gpurun_out/step_sequence.txt 2>&1
rm -rf gpurun_out/prof
head -8 gpurun_out/steady_r03b.txt; wc -l gpurun_out/step_sequence.txt
