export TMPDIR=/tmp
R=$PWD
rm -rf $R/gpurun_out/prof16; mkdir -p $R/gpurun_out/prof16
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof16 -o b -- python3 $R/bench.py --variant B --frames 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof16/stdout.log 2>&1
cd $R
find gpurun_out/prof16 -name '*kernel_trace.csv' -delete
f=$(find gpurun_out/prof16 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
lib=sum(float(r['TotalDurationNs']) for r in rows if r['Name'].startswith(('Cijk','Custom')))
print('total kernel ms', tot/1e6, 'library gemm share', round(lib/tot,3))
for r in rows[:16]:
    n=r['Name'].replace('(anonymous namespace)::','').replace('void ','')[:60]
    print(f"{n:60s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']:>6s}%")
PY
