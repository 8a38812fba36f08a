"""Per-kernel mean FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc passes -> JSON {kernel: {...}}.
Units: FETCH_SIZE / WRITE_SIZE are reported in kilobytes (1024 B).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request of a wide
coalesced streaming read -> doubled here; WRITE_SIZE is taken as reported (uncalibrated)."""
import csv, glob, json, sys, collections, re


def load(d, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get('Counter_Name') != counter:
                continue
            n = r['Kernel_Name']
            n = re.sub(r'\(.*$', '', n.replace('(anonymous namespace)::', '').replace('void ', '')).strip()
            a = acc[n]
            a[0] += float(r['Counter_Value'])
            a[1] += 1
    return acc


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 1])[0])):
    if not (k.startswith(('wgrad', 'attn_', 'lnv_', 'rowgemm', 'fold_', 'dbias', 'adamw', 'gelu', 'ln_', 'gemm_nt', 'gemm_ws', 'splitk_', 'sgemm', 'focal', 'transpose_', 'patch_', 'sumsq'))):
        continue
    f, w = fetch.get(k, [0.0, 0]), write.get(k, [0.0, 0])
    fb = 2.0 * 1024.0 * f[0] / max(f[1], 1)
    wb = 1024.0 * w[0] / max(w[1], 1)
    out[k] = dict(launches=f[1], fetch_bytes_per_launch=round(fb), write_bytes_per_launch=round(wb),
                  hbm_bytes_per_launch=round(fb + wb))
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_hash import csrc_sha16
out['_meta'] = dict(commit=os.environ.get('CLOVER_COMMIT', 'unknown'), csrc_sha16=csrc_sha16(), command='bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing ' + os.environ.get('BENCH_ARGS', ''),
                    note='FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B); per launch = mean over the launches of the run')
print(json.dumps(out, indent=1))
