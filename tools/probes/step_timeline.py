"""Chronological kernel timeline of ONE steady-state step from a rocprofv3 --kernel-trace CSV of `bench.py`
(steps delimited by the big AdamW launches): start offset, duration, gap, queue, grid, kernel.  Also totals per queue
and per kernel family, so that the main-stream critical path can be read separately from the text tower's stream.

    python tools/probes/step_timeline.py <kernel_trace.csv> [out.txt]
"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
cols = rows[0].keys()
qcol = 'Stream_Id' if 'Stream_Id' in cols else 'Queue_Id'
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get(qcol, '?'),
              r.get('Grid_Size_X', '?'), r.get('Workgroup_Size_X', '?'), r.get('Queue_Id', '?')) for r in rows),
            key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if 'adamw' in e[2] and e[1] - e[0] > 200000]
per = [ev[adam[i + 1]][0] - ev[adam[i]][0] for i in range(len(adam) - 1)]
med = sorted(per)[len(per) // 2]
good = [i for i, p in enumerate(per) if abs(p - med) < 0.05 * med]
i0 = good[len(good) // 2]
# a step = from the end of one step's last adamw to the end of the next one's; adamw launches come in groups
lo = ev[adam[i0]][1]
nxt = [a for a in adam if ev[a][0] > lo + 0.5 * med]
hi = ev[nxt[-1] if len(nxt) == 1 else [a for a in nxt if ev[a][0] < lo + 1.3 * med][-1]][1]
win = [e for e in ev if e[0] >= lo and e[1] <= hi]


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    if n.startswith(('Cijk', 'Custom')):
        m = re.search(r'MT(\d+x\d+x\d+)', n)
        return 'hipBLASLt ' + (m.group(1) if m else '')
    if 'at::native' in n:
        fn = re.findall(r'at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)', n)
        return 'aten ' + ' '.join(fn[:3])[:60]
    return n.split('(')[0][:70]


print(f'columns: {list(cols)}', file=out)
print(f'step wall {(hi - lo) / 1e3:.1f} us, {len(win)} kernels, median period {med / 1e3:.1f} us; stream column {qcol}', file=out)
byq = collections.defaultdict(lambda: [0, 0])
fam = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
last_end = collections.defaultdict(lambda: lo)
for s, e, n, q, g, w, qq in win:
    byq[q][0] += 1
    byq[q][1] += e - s
    fam[q][short(n)][0] += 1
    fam[q][short(n)][1] += e - s
for q, (c, t) in sorted(byq.items(), key=lambda kv: -kv[1][1]):
    print(f'stream {q}: {c} kernels, {t / 1e3:.1f} us of kernel time', file=out)
    for k, (c2, t2) in sorted(fam[q].items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'    {k:72s} n={c2:4d} {t2 / 1e3:9.1f} us', file=out)
print('\n  t_us     dur    gap  stream  grid      wg   kernel', file=out)
prev_end = lo
for s, e, n, q, g, w, qq in win:
    print(f'{(s - lo) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {(s - prev_end) / 1e3:6.1f}  {q:>4s}  {g:>8s} {w:>4s}  {short(n)}', file=out)
    prev_end = max(prev_end, e)
