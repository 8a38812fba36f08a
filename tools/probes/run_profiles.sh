set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/pytest_gpu.log | tail -2
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
cut -c1-300 gpurun_out/bench_n1.json
export TMPDIR=/tmp
R=$PWD
rm -rf gpurun_out/prof
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py > $R/gpurun_out/bench_prof.json 2> $R/gpurun_out/bench_prof.err; echo "rocprof rc=$?"
cd $R
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1); cp $f gpurun_out/kernel_stats.csv; head -8 gpurun_out/kernel_stats.csv | cut -c1-150
find gpurun_out/prof -name '*kernel_trace.csv' -delete; find gpurun_out/prof -name '*.db' -delete
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1; tail -3 gpurun_out/pmc_traffic.log | cut -c1-200
bash tools/pmc_mfma.sh > gpurun_out/pmc_mfma.log 2>&1; tail -3 gpurun_out/pmc_mfma.log | cut -c1-200
