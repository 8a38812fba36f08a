#!/bin/bash
# FETCH_SIZE of the grouped weight-gradient launch per Swin stage (tools/probes/wgrad_group.py, ONLY=s0..s3) against the
# algorithmic bytes: where does the over-fetch come from?
export TMPDIR=/tmp
R=$PWD
for only in "" s0 s1 s2; do
  rm -rf /tmp/wf; mkdir -p /tmp/wf
  (cd /tmp && ONLY=$only timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/wf -o p -- python3 $R/tools/probes/wgrad_group.py > /tmp/wf/out.log 2>&1)
  tail -1 /tmp/wf/out.log
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/wf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == 'FETCH_SIZE' and 'wgrad' in r['Kernel_Name']:
            n = r['Kernel_Name'].split('(')[0][-40:]
            acc[n][0] += float(r['Counter_Value']); acc[n][1] += 1
for n, (v, c) in acc.items():
    print('ONLY=$only', n, 'launches', c, 'fetch MB/launch', round(2 * 1024 * v / c / 1e6, 1))
PY
done
