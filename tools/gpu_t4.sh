set -u
cd $GRAFT_REPO_ROOT
CLOVER_LIB_PATH=$PWD/tools/probes/bin/libclover_trace.so WT_TRACE=1 SETS=s0,s1,s2,s3 python tools/probes/wgrad_traffic.py 2>&1 | grep SET | cut -d' ' -f2,12-
NOFOLD=1 SETS=s0,s1,s2,s3,all python tools/probes/wgrad_traffic.py 2>&1 | grep SET | sed "s/^/nofold /" | cut -d' ' -f1,3,12-
