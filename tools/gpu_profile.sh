#!/bin/bash
# One GPU-box pass for the round's evidence: default bench line, rocprofv3 kernel stats of the same command, the two
# HBM-traffic counter passes and the MFMA / wait counter pass (each its own run, --kernel-trace only: see tools/pmc_*.sh).
#   CLOVER_COMMIT=<sha> bash tools/gpu_profile.sh <tag>      -> gpurun_out/<tag>_*  (copy into profiles/ to commit)
set -u
TAG=${1:-r03}
export CLOVER_COMMIT=${CLOVER_COMMIT:-unknown}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/${TAG}_bench_n1.json 2> gpurun_out/${TAG}_bench_n1.err; echo "bench rc=$?"
tail -1 gpurun_out/${TAG}_bench_n1.json | cut -c1-400
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof; mkdir -p $R/gpurun_out/prof
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_prof.json 2> $R/gpurun_out/${TAG}_bench_prof.err); echo "rocprof rc=$?"
f=$(find gpurun_out/prof -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${TAG}_kernel_stats_bench_default_final.csv && head -8 "$f" | cut -c1-150
echo "{\"commit\": \"$CLOVER_COMMIT\", \"csrc_sha16\": \"$(python3 tools/csrc_hash.py)\", \"command\": \"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline\"}" > gpurun_out/${TAG}_kernel_stats_bench_default_final.meta.json
find gpurun_out/prof -name '*kernel_trace.csv' -delete; find gpurun_out/prof -name '*.db' -delete
bash tools/pmc_traffic.sh > gpurun_out/${TAG}_pmc_traffic.log 2>&1; cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
bash tools/pmc_mfma.sh > gpurun_out/${TAG}_pmc_mfma.log 2>&1; cp gpurun_out/pmc_mfma.json gpurun_out/${TAG}_pmc_mfma.json
rm -rf gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE gpurun_out/pmc_mfma gpurun_out/prof
python3 - <<PY
import json
t=json.load(open('gpurun_out/${TAG}_pmc_traffic.json')); m=json.load(open('gpurun_out/${TAG}_pmc_mfma.json'))
for k in list(m)[:6]:
    print(k, t.get(k), m.get(k))
PY
