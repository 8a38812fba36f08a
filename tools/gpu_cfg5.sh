#!/bin/bash
# BASELINE config 5's per-GPU workload (VideoSwin-B, 32 frames, B = 32 per GPU, fp8 forward GEMMs) as a workload: bench
# line (fp8 and bf16 on the same box), rocprofv3 kernel stats of the fp8 command, HBM-traffic + MFMA counter passes.
#   CLOVER_COMMIT=<sha> bash tools/gpu_cfg5.sh <tag>     -> gpurun_out/<tag>_cfg5_*   (copy into profiles/ to commit)
set -u
TAG=${1:-r04}
export CLOVER_COMMIT=${CLOVER_COMMIT:-unknown}
ARGS="--variant B --frames 32 --batch ${CFG5_BATCH:-32}"
mkdir -p gpurun_out
for dt in fp8 bf16; do
  timeout 900 python bench.py $ARGS --dtype $dt --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_cfg5_bench_$dt.json 2> gpurun_out/${TAG}_cfg5_bench_$dt.err; echo "bench $dt rc=$?"
  tail -1 gpurun_out/${TAG}_cfg5_bench_$dt.json | cut -c1-300
done
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof5; mkdir -p $R/gpurun_out/prof5
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof5 -o bench -- python3 $R/bench.py $ARGS --dtype fp8 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/${TAG}_cfg5_prof.json 2> $R/gpurun_out/${TAG}_cfg5_prof.err); echo "rocprof rc=$?"
f=$(find gpurun_out/prof5 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_cfg5_kernel_stats_fp8.csv && head -12 "$f" | cut -c1-150
echo "{\"commit\": \"$CLOVER_COMMIT\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py $ARGS --dtype fp8 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing\"}" > gpurun_out/${TAG}_cfg5_kernel_stats_fp8.meta.json
rm -rf gpurun_out/prof5
export BENCH_ARGS="$ARGS --dtype fp8"
bash tools/pmc_traffic.sh > gpurun_out/${TAG}_cfg5_pmc_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_cfg5_pmc_traffic.json
bash tools/pmc_mfma.sh > gpurun_out/${TAG}_cfg5_pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma.json gpurun_out/${TAG}_cfg5_pmc_mfma.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mfma
