set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm_nt or transpose_batch" > gpurun_out/pytest_k.log 2>&1; echo "pytest-k rc=$?"
tail -5 gpurun_out/pytest_k.log
bash tools/probes/run_gemm_trace.sh "$@" > /dev/null 2>&1
python tools/probes/gemm_trace_table.py gpurun_out/gemm_trace.txt
