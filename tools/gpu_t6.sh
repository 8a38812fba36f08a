set -u
cd $GRAFT_REPO_ROOT
for v in noloop; do
  L=tools/probes/bin/libclover_$v.so
  NOFOLD=1 CLV_WGRAD_TILE=2 CLOVER_LIB_PATH=$PWD/$L SETS=s0,s1,s2,s3,all python tools/probes/wgrad_traffic.py 2>&1 | grep SET | sed "s/^/$v /" | cut -d' ' -f1,3,4,5,12-
done
