#!/bin/bash
# Round-6 evidence on ONE box: full GPU suite, default bench + rocprofv3 stats + PMC passes (tools/gpu_profile.sh), kernel
# timeline, same-box A/B against the round-5 tree, the data-parallel structure's cost at one rank, the bf16 build's bench
# line, config 5's per-GPU shapes in fp16 / bf16 / fp8.    CLOVER_COMMIT=<sha> bash tools/gpu_r6final.sh
set -u
TAG=r06
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -6 > gpurun_out/${TAG}_pytest_gpu.txt
tail -2 gpurun_out/${TAG}_pytest_gpu.txt
bash tools/gpu_profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1; tail -8 gpurun_out/${TAG}_profile.log
bash tools/gpu_timeline.sh $TAG > /dev/null 2>&1; head -6 gpurun_out/${TAG}_timeline.txt
bash tools/ab_trees.sh 3 > gpurun_out/${TAG}_ab_trees.txt 2>&1; cat gpurun_out/${TAG}_ab_trees.txt
bash tools/gpu_dp_cost.sh 2 > gpurun_out/${TAG}_dp_cost.txt 2>&1; cat gpurun_out/${TAG}_dp_cost.txt
timeout 600 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${TAG}_bench_n1_bf16.json 2> /dev/null; tail -1 gpurun_out/${TAG}_bench_n1_bf16.json | cut -c1-260
for dt in f16 bf16 fp8; do
  timeout 900 python bench.py --variant B --frames 32 --batch 32 --dtype $dt --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_cfg5_bench_$dt.json 2> gpurun_out/${TAG}_cfg5_bench_$dt.err; echo "cfg5 $dt rc=$?"
  tail -1 gpurun_out/${TAG}_cfg5_bench_$dt.json | cut -c1-260
done
