"""Split a rocprofv3 --pmc pass over tools/probes/wgrad_traffic.py per problem set: dispatches are walked in order, a scan
kernel (the probe's MARK.cumsum) closes a set.  Usage: wgrad_traffic_sum.py <dir FETCH_SIZE> <dir WRITE_SIZE> <stdout.log>
FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md); both counters are in KiB."""
import csv, glob, sys, re, json, collections


def walk(d, counter):
    rows = []
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') == counter:
                rows.append((int(r['Dispatch_Id']), r['Kernel_Name'], float(r['Counter_Value'])))
    rows.sort()
    sets, cur = [], collections.defaultdict(lambda: [0.0, 0])
    for _, k, v in rows:
        if 'scan' in k.lower() or 'cumsum' in k.lower():
            if cur:                                         # a cumsum is more than one kernel: only the first closes a set
                sets.append(cur)
            cur = collections.defaultdict(lambda: [0.0, 0])
            continue
        k = re.sub(r'\(.*$', '', k.replace('(anonymous namespace)::', '').replace('void ', '')).strip()
        if k.startswith(('wgrad', 'fold_')):
            cur[k][0] += v
            cur[k][1] += 1
    return sets


fetch, write = walk(sys.argv[1], 'FETCH_SIZE'), walk(sys.argv[2], 'WRITE_SIZE')
names = [l.split() for l in open(sys.argv[3]) if l.startswith('SET ')]
out = {}
for i, w in enumerate(names):
    name, reps, alg = w[1], int(w[5]), int(w[7])
    ent = dict(alg_bytes=alg, kernels={})
    tot = 0.0
    for k in sorted(set(fetch[i]) | set(write[i])):
        f, wv = fetch[i].get(k, [0, 0]), write[i].get(k, [0, 0])
        fb, wb = 2 * 1024 * f[0] / reps, 1024 * wv[0] / reps
        ent['kernels'][k] = dict(launches_per_run=f[1] / reps, fetch_bytes=round(fb), write_bytes=round(wb))
        tot += fb + wb
    ent['traffic_bytes'] = round(tot)
    ent['traffic_over_alg'] = round(tot / alg, 3)
    out[name] = ent
print(json.dumps(out, indent=1))
