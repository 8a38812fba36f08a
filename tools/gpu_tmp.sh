set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "deferred_table or window_attention" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_engine_gpu.py tests/test_step_gpu.py -x -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm version\|^Hostname\|^Librccl" | tail -2
bash tools/gpu_ab.sh CLOVER_DBIAS_SUM_AUX 0 1
