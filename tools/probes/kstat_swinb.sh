#!/bin/bash
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/ks; mkdir -p /tmp/ks
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python3 $R/bench.py --variant B --frames 16 --batch 8 --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3 > /tmp/ks/out.log 2>&1)
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
adam = [r for r in rows if 'adamw_dev' in r['Name']][0]; S = int(adam['Calls']) / 2
tot = sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / S
print('steps', S, 'kernel ms/step', round(tot, 2))
for r in rows[:22]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    print(f"{n[:64]:64s} {int(r['Calls']) / S:6.1f} x {float(r['AverageNs']) / 1e3:8.1f} us = {float(r['TotalDurationNs']) / 1e6 / S:6.2f} ms")
PY
