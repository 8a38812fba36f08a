"""Ordered kernel list of ONE steady-state step from a rocprofv3 --kernel-trace CSV (start offset us, duration us, gap to the
previous kernel's end, stream/queue, short name): the anatomy of the eager sections between the hipGraph replays.

    python tools/probes/step_sequence.py <kernel_trace.csv> > step_sequence.txt
"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows))
adam = [i for i, e in enumerate(ev) if 'adamw' in e[2] and e[1] - e[0] > 200000]
k = len(adam) // 2
k -= k % 2                      # two AdamW launches per step: start at the first one of a step
lo, hi = adam[k - 2], adam[k]
t0, end = ev[lo][0], ev[lo][0]
for s, e, n, q in ev[lo:hi]:
    short = n.replace('(anonymous namespace)::', '').replace('void ', '').replace('at::native::', '')
    short = short.split('(')[0][:90]
    print(f'{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} gap {(s - end) / 1e3:7.1f} q{q} {short}')
    end = max(end, e)
