#!/bin/bash
# per-kernel average (us) from a short rocprofv3 --stats run of bench.py:  kstat.sh <regex>   (env passes through)
export TMPDIR=/tmp
R=$PWD
rm -rf /tmp/ks; mkdir -p /tmp/ks
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o b -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 > /tmp/ks/out.log 2>&1)
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1)
python3 - "$f" "$1" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r['Name']):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        print(f"{n[:70]:70s} calls {r['Calls']:>6s} avg {float(r['AverageNs']) / 1e3:8.1f} us")
PY
