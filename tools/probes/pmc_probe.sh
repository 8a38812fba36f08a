#!/bin/bash
# usage: pmc_probe.sh <out-tag> <python probe + args...>   -- two SQ counter passes, per-kernel means
set -u
export TMPDIR=/tmp
R=$PWD; tag=$1; shift; probe=$1; shift
for i in 1 2 3; do
  rm -rf $R/gpurun_out/pmc_$tag$i; mkdir -p $R/gpurun_out/pmc_$tag$i
  ctrs=$(sed 's/^pmc: //' $R/tools/probes/pmc_sq$i.txt)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag$i -o p -- python3 $R/$probe "$@" > $R/gpurun_out/pmc_$tag$i/stdout.log 2>&1)
  find gpurun_out/pmc_$tag$i -name '*kernel_trace.csv' -delete
done
python3 - "$tag" <<'PY'
import csv, glob, collections, re, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f'gpurun_out/pmc_{tag}*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r'\(.*$', '', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')).strip()
        if n.startswith(('Cijk', 'at::', '__amd')): continue
        a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for k, d in acc.items():
    m = {c: v[0] / v[1] for c, v in d.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(f"{k[:60]:60s} waves {m.get('SQ_WAVES',0):8.0f} wave_cyc(quad) {wc:12.0f} busy {m.get('SQ_BUSY_CYCLES',0):10.0f}")
    print(f"    frac of wave cycles: wait_any {m.get('SQ_WAIT_ANY',0)/wc:.3f} wait_inst_any {m.get('SQ_WAIT_INST_ANY',0)/wc:.3f} active_any {m.get('SQ_ACTIVE_INST_ANY',0)/wc:.3f} valu {m.get('SQ_ACTIVE_INST_VALU',0)/wc:.3f} lds {m.get('SQ_ACTIVE_INST_LDS',0)/wc:.3f}")
    if 'TCC_HIT_sum' in m: print(f"    L2: hit {m['TCC_HIT_sum']:12.0f} miss {m['TCC_MISS_sum']:12.0f} hit-rate {m['TCC_HIT_sum']/max(m['TCC_HIT_sum']+m['TCC_MISS_sum'],1):.3f} ea_rdreq {m.get('TCC_EA0_RDREQ_sum',0):12.0f} req {m.get('TCC_REQ_sum',0):12.0f}")
    print(f"    insts/wave: valu {m.get('SQ_INSTS_VALU',0)/max(m.get('SQ_WAVES',1),1):8.0f} mfma {m.get('SQ_INSTS_MFMA',0)/max(m.get('SQ_WAVES',1),1):6.0f} lds {m.get('SQ_INSTS_LDS',0)/max(m.get('SQ_WAVES',1),1):7.0f} vmem {m.get('SQ_INSTS_VMEM',0)/max(m.get('SQ_WAVES',1),1):6.0f} salu {m.get('SQ_INSTS_SALU',0)/max(m.get('SQ_WAVES',1),1):7.0f}  lds_bank_conflict {m.get('SQ_LDS_BANK_CONFLICT',0):12.0f} lds_idx_active {m.get('SQ_LDS_IDX_ACTIVE',0):12.0f} vmem_cycles {m.get('SQ_INST_CYCLES_VMEM',0):10.0f}")
PY
