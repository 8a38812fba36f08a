"""Median duration per consecutive run of kernels matching a pattern (runs are broken by ANY other kernel):
   trace_runs.py <kernel_trace.csv> <pattern> [label,label,...]"""
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
pat = sys.argv[2]
labels = sys.argv[3].split('|') if len(sys.argv) > 3 else []
runs, cur = [], None
for r in rows:
    if pat in r['Kernel_Name']:
        d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        if cur is None: cur = []; runs.append(cur)
        cur.append(d)
    elif ('fold_partials' in r['Kernel_Name'] or 'wgrad_' in r['Kernel_Name']) and 'fill' not in r['Kernel_Name'].lower():
        continue                                  # wgrad / fold launches alternate inside one shape's run
    else:
        cur = None
out = []
for i, v in enumerate(runs):
    if len(v) < 4: continue
    out.append(statistics.median(v) / 1e3)
print(' '.join(f'{(labels[i] if i < len(labels) else i)}={v:.1f}' for i, v in enumerate(out)))
