export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/gtrace; mkdir -p $R/gpurun_out/gtrace
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gtrace -o t -- python3 $R/tools/probes/gemm_bench.py "$@" > $R/gpurun_out/gtrace/stdout.log 2>&1
cd $R
f=$(find gpurun_out/gtrace -name '*kernel_trace.csv' | head -1)
python tools/probes/trace_summary.py $f > gpurun_out/gemm_trace.txt
rm -f $f
cat gpurun_out/gemm_trace.txt
