set -u
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|^E " gpurun_out/pytest_gpu.log | tail -5
bash tools/probes/ab.sh "$1" "$2"
