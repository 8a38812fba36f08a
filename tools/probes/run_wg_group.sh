set -u
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "grouped or wgrad" 2>&1 | tail -2
for b in 0 1; do echo "BIG=$b"; CLV_WGRAD_BIG=$b python tools/probes/wgrad_group.py 2>&1 | grep problems; done
for t in 12 24 48 96; do echo "BIG target $t"; CLV_WGRAD_BIG_TARGET=$t python tools/probes/wgrad_group.py 2>&1 | grep problems; done
