"""Where does a step's wall time go?  Reads a rocprofv3 --kernel-trace CSV of `bench.py` and splits the last steps into
(a) time with >= 1 kernel running, attributed to the kernels that ran ALONE (nothing else on the device) or overlapped,
(b) idle time — no kernel on the device at all — attributed to the kernel that ended before each hole,
(c) per-queue busy time and the distribution of the holes.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 $R/bench.py --steps 12 --warmup 6 \
        --no-cpu-baseline --no-kernel-timing
    python tools/probes/timeline.py out/.../t_kernel_trace.csv [marker-substring]

A step = the span between two consecutive launches of the marker kernel (default: optim_prep_kernel, once per step)."""
import csv
import sys
from collections import defaultdict


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    cut = name.find('(')
    return (name if cut < 0 else name[:cut])[:70]


def main():
    path = sys.argv[1]
    marker = sys.argv[2] if len(sys.argv) > 2 else 'optim_prep_kernel'
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r.get('Queue_Id', '0'),
                         int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0) // max(1, int(r.get('Workgroup_Size_X') or r.get('Workgroup_Size') or 1))))
    rows.sort()
    marks = [r[0] for r in rows if marker in r[2]]
    if len(marks) < 4:
        print('marker', marker, 'seen', len(marks), 'times; kernels:', len(rows))
        return
    lo, hi = marks[-7] if len(marks) >= 7 else marks[1], marks[-1]
    nsteps = (6 if len(marks) >= 7 else len(marks) - 2)
    if len(sys.argv) > 3:        # one step, kernel by kernel: start (us from the marker), duration, hole before it on its queue, queue, workgroups, name
        with open(sys.argv[3], 'w') as out:
            qend = {}
            for s, e, n, q, g in rows:
                if marks[-2] <= s < marks[-1]:
                    out.write(f'{(s - marks[-2]) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - qend.get(q, s)) / 1e3:7.1f} q{q} {g:6d} {n}\n')
                qend[q] = max(qend.get(q, 0), e)
    ks = [r[:4] for r in rows if lo <= r[0] < hi]
    wall = (hi - lo) / nsteps
    print(f'{nsteps} steps, {len(ks) / nsteps:.0f} kernels / step, wall {wall / 1e6:.3f} ms / step, '
          f'sum of kernel durations {sum(e - s for s, e, _, _ in ks) / nsteps / 1e6:.3f} ms')
    # sweep: events
    ev = []
    for i, (s, e, n, q) in enumerate(ks):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    active = set()
    alone = defaultdict(float)
    shared = defaultdict(float)
    idle_after = defaultdict(float)
    idle_n = defaultdict(int)
    holes = []
    last_t, last_ended = lo, None
    busy = 0.0
    for t, kind, i in ev:
        dt = t - last_t
        if dt > 0:
            if not active:
                if last_ended is not None:
                    idle_after[ks[last_ended][2]] += dt
                    idle_n[ks[last_ended][2]] += 1
                    holes.append(dt)
            else:
                busy += dt
                if len(active) == 1:
                    alone[ks[next(iter(active))][2]] += dt
                else:
                    for j in active:
                        shared[ks[j][2]] += dt / len(active)
        if kind:
            active.add(i)
        else:
            active.discard(i)
            last_ended = i
        last_t = t
    idle = sum(holes)
    print(f'device busy {busy / nsteps / 1e6:.3f} ms, idle {idle / nsteps / 1e6:.3f} ms / step in {len(holes) / nsteps:.0f} holes '
          f'(median {sorted(holes)[len(holes) // 2] / 1e3:.1f} us, p90 {sorted(holes)[int(len(holes) * .9)] / 1e3:.1f} us, '
          f'max {max(holes) / 1e3:.1f} us)')
    qb = defaultdict(float)
    qn = defaultdict(int)
    for s, e, n, q in ks:
        qb[q] += e - s
        qn[q] += 1
    for q in sorted(qb, key=lambda q: -qb[q]):
        print(f'  queue {q}: {qn[q] / nsteps:.0f} kernels, {qb[q] / nsteps / 1e6:.3f} ms busy / step')
    print('\nper step, by kernel: alone = only kernel on the device; shared = its share of overlapped time; hole-after = idle time '
          'that follows it')
    names = sorted(set(alone) | set(shared) | set(idle_after), key=lambda n: -(alone[n] + shared[n] + idle_after[n]))
    cnt = defaultdict(int)
    for _, _, n, _ in ks:
        cnt[n] += 1
    print(f'{"kernel":70s} {"calls":>6s} {"alone us":>9s} {"shared us":>9s} {"hole us":>8s} {"holes":>6s}')
    for n in names[:60]:
        print(f'{n:70s} {cnt[n] / nsteps:6.1f} {alone[n] / nsteps / 1e3:9.1f} {shared[n] / nsteps / 1e3:9.1f} '
              f'{idle_after[n] / nsteps / 1e3:8.1f} {idle_n[n] / nsteps:6.1f}')
    # coarse phases of the step: the largest holes
    big = sorted(holes, reverse=True)[:10]
    print('\nlargest holes (us):', ' '.join(f'{h / 1e3:.0f}' for h in big))


if __name__ == '__main__':
    main()
