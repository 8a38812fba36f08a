#!/bin/bash
# MFMA utilisation of the step's kernels: one rocprofv3 --pmc pass (SQ + GRBM counters), kernel-trace only, eager step so
# every kernel is a dispatch.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 1024 SIMDs), kernel cycles =
# GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs: a 6.3 us fill reads 289 605 = 8 x 2.4 GHz x 6.3 us x 2.39);
# one 16x16x32 bf16 MFMA keeps a SIMD busy 16 cycles (checked: busy = 16 x SQ_INSTS_MFMA), so the ratio is the
# fraction of the dense bf16 MFMA peak.  ROCm 7.2 ships no derived metrics for gfx950 (MI355X_MICROARCH.md).
set -u
export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/pmc_mfma; mkdir -p $R/gpurun_out/pmc_mfma
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE \
   --output-format csv -d $R/gpurun_out/pmc_mfma -o p -- \
   python3 $R/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing ${BENCH_ARGS:-} > $R/gpurun_out/pmc_mfma/stdout.log 2>&1)
echo "rc=$?"
find gpurun_out/pmc_mfma -name '*kernel_trace.csv' -delete
python3 - <<'PY' > gpurun_out/pmc_mfma.json
import csv, glob, json, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if n.startswith(('Cijk', 'Custom_Cijk')):
            n = 'hipBLASLt GEMM (all shapes)'
        else:
            n = re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', '')).strip()
        a = acc[n][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
out = {}
for k, d in acc.items():
    m = {c: v[0] / v[1] for c, v in d.items()}
    if not m.get('GRBM_GUI_ACTIVE') or 'at::native' in k or k.startswith('__amd'):
        continue
    out[k] = dict(launches=int(d['GRBM_GUI_ACTIVE'][1]),
                  mfma_util=round(m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (m['GRBM_GUI_ACTIVE'] / 8 * 1024), 4),
                  valu_active_frac_of_wave_cycles=round(m.get('SQ_ACTIVE_INST_VALU', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1), 4),
                  wait_frac_of_wave_cycles=round(m.get('SQ_WAIT_ANY', 0) / max(m.get('SQ_WAVE_CYCLES', 1), 1), 4),
                  mfma_insts_per_launch=round(m.get('SQ_INSTS_MFMA', 0)), gui_active_cycles=round(m['GRBM_GUI_ACTIVE']))
import os, sys
sys.path.insert(0, 'tools')
from csrc_hash import csrc_sha16
res = dict(sorted(out.items(), key=lambda kv: -kv[1]['gui_active_cycles'] * kv[1]['launches']))
res['_meta'] = dict(commit=os.environ.get('CLOVER_COMMIT', 'unknown'), csrc_sha16=csrc_sha16(), command='bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing ' + os.environ.get('BENCH_ARGS', ''))
print(json.dumps(res, indent=1))
PY
head -c 1800 gpurun_out/pmc_mfma.json
find gpurun_out/pmc_mfma -name '*counter_collection.csv' -size +20M -delete
