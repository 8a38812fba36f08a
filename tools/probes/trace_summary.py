"""Median duration of each consecutive run of one (kernel, grid) in a rocprofv3 --kernel-trace CSV, in launch
order (a probe that loops shape by shape therefore prints one line per shape and kernel)."""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
rows.sort(key=lambda r: int(r['Start_Timestamp']))
runs = []
for r in rows:
    n = r['Kernel_Name']
    if pat not in n: continue
    short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][-44:]
    key = (short, r.get('Grid_Size_X', '?'))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    if runs and runs[-1][0] == key: runs[-1][1].append(d)
    else: runs.append((key, [d]))
if '--agg' in sys.argv:                      # aggregate by (kernel, grid) instead of consecutive runs
    agg = {}
    for k, v in runs:
        agg.setdefault(k, []).extend(v)
    runs = list(agg.items())
tot = 0
for k, v in runs:
    if len(v) < 5: continue
    tot += statistics.median(v)
    print(f'{k[0]:40s} grid {k[1]:>8s} n={len(v):4d} median {statistics.median(v)/1e3:8.1f} us  min {min(v)/1e3:8.1f}')
print('sum of medians (us):', tot / 1e3)
