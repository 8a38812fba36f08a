#!/bin/bash
# Per-kernel means of a set of rocprofv3 PMC counters for one command (kernel-trace only, as the pool requires):
#   bash tools/pmc_probe.sh <tag> "<kernel-name regex>" "<counters of pass 1>" ["<counters of pass 2>" ...] -- python3 script.py args
set -u
TAG=$1; PAT=$2; shift 2
PASSES=()
while [ "$1" != "--" ]; do PASSES+=("$1"); shift; done
shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1)); D=/tmp/pmcp_$i; rm -rf $D; mkdir -p $D
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $D -o p -- "$@" > $D/stdout.log 2>&1); echo "pass $i rc=$?"
done
python3 - "$PAT" <<'PY' | tee $R/gpurun_out/${TAG}_pmc_probe.txt
import csv, glob, collections, re, sys
pat = re.compile(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob('/tmp/pmcp_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')).strip()
        if not pat.search(n):
            continue
        a = acc[n][r['Counter_Name']]
        a[0] += float(r['Counter_Value']); a[1] += 1
for k, d in acc.items():
    print(k)
    for c in sorted(d):
        print(f'    {c:32s} {d[c][0] / d[c][1]:16.1f}   ({d[c][1]} launches)')
PY
