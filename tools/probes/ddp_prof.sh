#!/bin/bash
# kernel-time families of the forced-collectives step (1-rank RCCL group, cut backward graphs) next to the plain step
export TMPDIR=/tmp
R=$PWD
for fc in 0 1; do
  rm -rf /tmp/dp$fc; mkdir -p /tmp/dp$fc
  (cd /tmp && CLOVER_FORCE_COLLECTIVES=$fc timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp$fc -o b -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timing --steps 30 --warmup 5 > /tmp/dp$fc/out.log 2>&1)
  f=$(find /tmp/dp$fc -name '*kernel_stats.csv' | head -1)
  python3 - "$f" $fc <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
adam = [r for r in rows if 'adamw_dev' in r['Name']][0]
S = int(adam['Calls']) / 2
def fam(n):
    if 'gemm_nt_kernel' in n: return 'gemm_nt'
    if 'attn_' in n or 'dbias' in n or 'seq_combine' in n: return 'attention'
    if 'lnv_' in n or 'ln_' in n or 'gelu' in n: return 'ln+gelu'
    if 'wgrad' in n or 'fold' in n: return 'wgrad'
    if 'rowgemm' in n: return 'rowgemm'
    if 'adamw' in n or 'sumsq' in n or 'optim' in n or 'transpose_batch' in n or 'pack_bf16' in n: return 'optimizer+pack'
    if n.startswith('Cijk'): return 'library'
    if 'ccl' in n.lower(): return 'rccl'
    if 'at::native' in n or 'rocclr' in n: return 'aten'
    return 'other'
t, c = {}, {}
for r in rows:
    f = fam(r['Name']); t[f] = t.get(f, 0) + float(r['TotalDurationNs']) / 1e6 / S; c[f] = c.get(f, 0) + int(r['Calls']) / S
print('FORCE_COLLECTIVES=' + sys.argv[2], 'steps', S, ' total %.2f ms' % sum(t.values()))
print('  ' + '  '.join(f'{k} {v:.2f}/{c[k]:.0f}' for k, v in sorted(t.items(), key=lambda kv: -kv[1])))
for r in rows:
    if 'wgrad' in r['Name'] or 'pack' in r['Name'] or 'ccl' in r['Name'].lower() or 'fold' in r['Name']:
        print('   ', r['Name'][:70], round(int(r['Calls']) / S, 1), round(float(r['AverageNs']) / 1e3, 1))
PY
done
