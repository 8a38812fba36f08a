#!/bin/bash
set -u
mkdir -p gpurun_out
python tools/probes/mlp_fused_bench.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tee gpurun_out/r6d_mlp_bench.txt
for g in 128 512; do CLV_FMLP_GRID=$g python tools/probes/mlp_fused_bench.py 2>&1 | grep "one-kernel" | sed "s/^/grid $g: /"; done | tee -a gpurun_out/r6d_mlp_bench.txt
bash tools/gpu_stats.sh r6d
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r6d_kernel_stats.csv')))
for r in rows[:45]:
    print(f"{r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:80]:80s} calls={r['Calls']:>6s} tot_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f}")
PY
