set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "window_attention or batchnorm" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_step_gpu.py -x -q 2>&1 | tail -4
bash tools/gpu_ab.sh CLOVER_DEFER_DBIAS 0 1
