#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r6n_engine.log
timeout 600 python -m pytest tests/test_step_gpu.py tests/test_bf16_build_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r6n_step.log
timeout 600 python bench.py --steps 30 --warmup 5 > gpurun_out/r6n_bench.json 2> gpurun_out/r6n_bench.err
tail -5 gpurun_out/r6n_engine.log; tail -3 gpurun_out/r6n_step.log; cut -c1-600 gpurun_out/r6n_bench.json
