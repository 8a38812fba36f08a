set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-kernel-timing > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"
cut -c1-330 gpurun_out/bench_a.json
timeout 600 python bench.py --no-kernel-timing > gpurun_out/bench_b.json 2> gpurun_out/bench_b.err
cut -c1-330 gpurun_out/bench_b.json
