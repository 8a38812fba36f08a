set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped_weight or wgrad or two_partial" 2>&1 | tail -5
CLOVER_LIB_PATH=$PWD/tools/probes/bin/libclover_trace.so WT_TRACE=1 SETS=s0,s1,s2,s3 python tools/probes/wgrad_traffic.py 2>&1 | grep SET | cut -d' ' -f2,12-
for v in base nodma nocompute; do
  if [ $v = base -o $v = old ]; then L=clover_amd/libclover_hip.so; else L=tools/probes/bin/libclover_$v.so; fi
  T=2; if [ $v = old ]; then T=0; fi
  NOFOLD=1 CLV_WGRAD_TILE=$T CLOVER_LIB_PATH=$PWD/$L SETS=${SETS:-s0,s1,s2,s3,all} python tools/probes/wgrad_traffic.py 2>&1 | grep SET | sed "s/^/$v /" | cut -d' ' -f1,3,4,5,12-
done
