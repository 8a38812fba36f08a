#!/bin/bash
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "mlm_decoder or focal or gemm_nt" > gpurun_out/r4f_t1.log 2>&1; echo "t1 rc=$?"; tail -8 gpurun_out/r4f_t1.log
timeout 900 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "own_decoder or graph_capture or first_touch_grad" > gpurun_out/r4f_t2.log 2>&1; echo "t2 rc=$?"; tail -8 gpurun_out/r4f_t2.log
for rep in 1 2; do for v in 0 1; do
  r=$(CLOVER_OWN_DECODER=$v python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['losses'])")
  echo "OWN_DECODER=$v rep$rep: $r"
done; done
