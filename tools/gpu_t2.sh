set -u
cd $GRAFT_REPO_ROOT
for v in old base nodma nocompute nomfma noread; do
  if [ $v = base -o $v = old ]; then L=clover_amd/libclover_hip.so; else L=tools/probes/bin/libclover_$v.so; fi
  T=2; if [ $v = old ]; then T=0; fi
  CLV_WGRAD_TILE=$T CLOVER_LIB_PATH=$PWD/$L SETS=${SETS:-s0,s1,s2,s3,all} python tools/probes/wgrad_traffic.py 2>&1 | grep SET | sed "s/^/$v /" | cut -d' ' -f1,3,4,5,12-
done
