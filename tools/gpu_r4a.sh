#!/bin/bash
# round-4 first GPU pass: new / changed tests, default bench line, library-vs-own GEMM table, config-5 workload profile
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "geometry or first_touch" > gpurun_out/r4a_t1.log 2>&1; echo "t1 rc=$?"; tail -3 gpurun_out/r4a_t1.log
timeout 1500 python -m pytest tests/test_step_gpu.py tests/test_parity_gpu.py -m gpu -x -q -s -k "checkpoint or one_frame" > gpurun_out/r4a_t2.log 2>&1; echo "t2 rc=$?"; tail -3 gpurun_out/r4a_t2.log
timeout 2400 python -m pytest tests/test_fp8_gpu.py tests/test_step_gpu.py -m gpu -q -s -k "fp8_step_losses or full_size_step" > gpurun_out/r4a_t3.log 2>&1; echo "t3 rc=$?"; grep -E "errors|passed|failed|Error" gpurun_out/r4a_t3.log | cut -c1-900
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err; echo "bench rc=$?"; tail -1 gpurun_out/r4a_bench.json | cut -c1-250
timeout 600 python tools/probes/gemm_tiles.py > gpurun_out/r4a_gemm_tiles.log 2>&1; cat gpurun_out/r4a_gemm_tiles.log
bash tools/gpu_cfg5.sh r04a
