#!/bin/bash
# time (plain run) and FETCH_SIZE (rocprofv3 --pmc run) of the grouped weight-gradient launch for env variants
#   usage: wgrad_sweep.sh "ONLY=s2" "ONLY=s2 CLV_WGRAD_GROUP_ROWS=512" ...
export TMPDIR=/tmp
R=$PWD
for cfg in "$@"; do
  t=$(env $cfg python3 $R/tools/probes/wgrad_group.py 2>/dev/null | tail -1)
  rm -rf /tmp/wf; mkdir -p /tmp/wf
  (cd /tmp && env $cfg timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/wf -o p -- python3 $R/tools/probes/wgrad_group.py > /tmp/wf/out.log 2>&1)
  f=$(python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('/tmp/wf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == 'FETCH_SIZE' and 'wgrad' in r['Kernel_Name']:
            n = r['Kernel_Name'].split('(')[0].split('::')[-1][:28]
            acc[n][0] += float(r['Counter_Value']); acc[n][1] += 1
print(' | '.join(f'{n} x{c // 23}: {2 * 1024 * v / 23 / 1e6:.0f} MB' for n, (v, c) in acc.items()))
PY
)
  echo "[$cfg] $t || fetch per run: $f"
done
