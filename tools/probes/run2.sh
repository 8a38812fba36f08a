set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nt or transpose_batch or adamw" > gpurun_out/pytest_k.log 2>&1; echo "pytest-k rc=$?"
tail -5 gpurun_out/pytest_k.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q > gpurun_out/pytest_e.log 2>&1; echo "pytest-e rc=$?"
tail -5 gpurun_out/pytest_e.log
timeout 600 python tools/probes/gemm_bench.py > gpurun_out/gemm_bench.txt 2>&1; echo "bench rc=$?"
cat gpurun_out/gemm_bench.txt
