#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "first_touch" 2>&1 | tail -3
bash tools/ab_trees.sh 3 2>&1 | tee gpurun_out/r6m_ab.txt
