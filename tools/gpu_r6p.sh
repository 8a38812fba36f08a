#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6p_engine.log
tail -6 gpurun_out/r6p_engine.log
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "weight_grad or wgrad or fold" 2>&1 | tail -3
bash tools/ab_trees.sh 2
