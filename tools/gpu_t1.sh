set -u
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "grouped_weight or wgrad or two_partial" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_engine_gpu.py -x -q -k "first_touch or use_checkpoint" 2>&1 | tail -15
for m in 0 2; do CLV_WGRAD_TILE=$m SETS=s0,s1,s2,s3,all python tools/probes/wgrad_traffic.py 2>&1 | grep SET | sed "s/^/TILE=$m /"; done
