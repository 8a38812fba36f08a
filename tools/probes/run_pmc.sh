#!/bin/bash
# usage: run_pmc.sh <kernel pattern> <python script> [args] -- two SQ counter passes, per-kernel means
pat=$1; shift
export TMPDIR=/tmp
R=$PWD
for i in 1 2; do
  rm -rf $R/gpurun_out/pmcq$i; mkdir -p $R/gpurun_out/pmcq$i
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -i $R/tools/probes/pmc_sq$i.txt --output-format csv -d $R/gpurun_out/pmcq$i -o p -- python3 $R/"$@" > $R/gpurun_out/pmcq$i/stdout.log 2>&1)
  find gpurun_out/pmcq$i -name '*kernel_trace.csv' -delete
done
python3 - "$pat" <<'PY'
import csv, glob, sys, collections, re
pat = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmcq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\(.*$', '', n)
        if pat not in n: continue
        a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, d in acc.items():
    print(k)
    m = {c: v[0] / v[1] for c, v in d.items()}
    for c in sorted(m): print(f'   {c:24s} {m[c]:16.0f}')
    if 'SQ_WAVE_CYCLES' in m:
        w = m['SQ_WAVE_CYCLES']
        print('   fractions of wave cycles: wait_any %.2f  wait_inst %.2f  active %.2f  (valu %.2f lds %.2f)' % (
            m.get('SQ_WAIT_ANY', 0) / w, m.get('SQ_WAIT_INST_ANY', 0) / w, m.get('SQ_ACTIVE_INST_ANY', 0) / w,
            m.get('SQ_ACTIVE_INST_VALU', 0) / w, m.get('SQ_ACTIVE_INST_LDS', 0) / w))
PY
find gpurun_out/pmcq1 gpurun_out/pmcq2 -name '*counter_collection.csv' -delete
