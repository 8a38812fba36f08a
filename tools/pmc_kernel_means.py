"""Per-kernel mean of every counter of a rocprofv3 --pmc pass: pmc_kernel_means.py <dir> [name filter substring]."""
import csv, glob, sys, re, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')).strip()
        if flt and flt not in n:
            continue
        a = acc[n][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
for n, d in acc.items():
    print(n, {c: round(v[0] / v[1]) for c, v in sorted(d.items())}, 'launches', max(v[1] for v in d.values()))
