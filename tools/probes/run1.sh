set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err; echo "bench rc=$?"
cut -c1-400 gpurun_out/bench_n1.json
timeout 900 bash tools/probes/run_timeline.sh --steps 12 --warmup 3 --no-kernel-timing
