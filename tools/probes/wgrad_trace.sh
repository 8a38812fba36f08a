export TMPDIR=/tmp
R=$PWD
L="fc1 s0|fc2 s0|qkv s1|fc1 s1|fc2 s1|qkv s2|proj s2|fc1 s2|fc2 s2|proj s3|qkv s3|fc1 s3|fc2 s3|qkv fu|fc1 fu|bert proj|bert fc1"
for cfg in "$@"; do
  rm -rf $R/gpurun_out/wtrace; mkdir -p $R/gpurun_out/wtrace
  (cd /tmp && env $cfg rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/wtrace -o t -- python3 $R/tools/probes/wgrad_shapes.py > /dev/null 2>&1)
  f=$(find gpurun_out/wtrace -name '*kernel_trace.csv' | head -1)
  echo "== $cfg"; echo -n "  main: "; python tools/probes/trace_runs.py $f wgrad_ "$L"; echo -n "  fold: "; python tools/probes/trace_runs.py $f fold_partials
done
