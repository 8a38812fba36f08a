python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "wgrad" 2>&1 | tail -2
echo "== new"; python tools/probes/wgrad_bench.py 2>/dev/null | cut -c1-120
echo "== old"; CLV_WGRAD_OLD=1 python tools/probes/wgrad_bench.py 2>/dev/null | cut -c1-120
