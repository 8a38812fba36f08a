"""Steady-state anatomy of the graph-replayed step from a rocprofv3 --kernel-trace CSV of `bench.py`:
picks a window of STEPS consecutive steps in the middle of the timed region (delimited by the big AdamW launches),
and reports kernels / step, busy time (union over streams), idle time, the per-kernel totals and the gap histogram.

    python tools/probes/steady_trace.py <kernel_trace.csv> [steps=8] [top=45]
"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 8
TOP = int(sys.argv[3]) if len(sys.argv) > 3 else 45
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows), key=lambda e: e[0])
adam = [i for i, e in enumerate(ev) if 'adamw' in e[2] and e[1] - e[0] > 200000]
# the timed graph-replay steps are the most regular run of AdamW-to-AdamW intervals: take the middle of the
# longest run whose period is within 10 % of the median period
per = [ev[adam[i + 1]][0] - ev[adam[i]][0] for i in range(len(adam) - 1)]
med = sorted(per)[len(per) // 2]
best, cur = (0, 0), None
for i, p in enumerate(per + [0]):
    ok = abs(p - med) < 0.1 * med
    if ok and cur is None:
        cur = i
    if not ok and cur is not None:
        if i - cur > best[1] - best[0]:
            best = (cur, i)
        cur = None
a0 = best[0] + max(0, (best[1] - best[0] - STEPS) // 2)
lo, hi = ev[adam[a0]][1], ev[adam[a0 + STEPS]][1]
win = [e for e in ev if e[0] >= lo and e[1] <= hi]
wall = hi - lo
busy, end, gaps = 0, lo, []
for s, e, n in win:
    if s > end:
        gaps.append(s - end)
        busy += e - s
        end = e
    elif e > end:
        busy += e - end
        end = e
ksum = sum(e - s for s, e, _ in win)
print(f'steady window: {STEPS} steps of {len(per)} intervals (median period {med / 1e6:.3f} ms, run {best})')
print(f'per step: wall {wall / STEPS / 1e6:.3f} ms  kernels {len(win) / STEPS:.0f}  kernel-time sum {ksum / STEPS / 1e6:.3f} ms  '
      f'busy(union) {busy / STEPS / 1e6:.3f} ms  idle {(wall - busy) / STEPS / 1e6:.3f} ms')
h = collections.Counter()
for g in gaps:
    h['<1us' if g < 1000 else '1-2us' if g < 2000 else '2-4us' if g < 4000 else '4-10us' if g < 10000 else '10-50us' if g < 50000 else '>50us'] += 1
tg = collections.Counter()
for g in gaps:
    tg['<1us' if g < 1000 else '1-2us' if g < 2000 else '2-4us' if g < 4000 else '4-10us' if g < 10000 else '10-50us' if g < 50000 else '>50us'] += g
for k in ['<1us', '1-2us', '2-4us', '4-10us', '10-50us', '>50us']:
    print(f'  gaps {k:8s} {h[k] / STEPS:7.1f}/step  {tg[k] / STEPS / 1e3:8.1f} us/step')
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    short = n.replace('(anonymous namespace)::', '').replace('void ', '')
    short = short.split('(')[0] if not short.startswith(('Cijk', 'Custom')) else 'hipBLASLt GEMM'
    if 'at::native' in short:
        import re
        body = short.split('at::native::', 1)[1]
        fn = re.findall(r'at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+(?:<[a-zA-Z0-9_:]+>)?)', short)
        short = 'aten ' + body.split('<')[0][:28] + ' ' + ' '.join(fn[1:3])[:60]
    agg[short][0] += 1
    agg[short][1] += e - s
print(f'{"kernel":70s} {"n/step":>7s} {"us/step":>9s} {"share":>6s}')
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:TOP]:
    print(f'{k[:70]:70s} {c / STEPS:7.1f} {t / STEPS / 1e3:9.1f} {100 * t / ksum:5.1f}%')
