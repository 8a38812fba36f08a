set -u
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -k "batchnorm or grouped_weight or mlm_decoder or heads" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_step_gpu.py -x -q -k "ablation or step_losses" 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -6
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('stale'), d['roofline'].get('traffic_source'), d.get('library_gemm_calls'))"
