set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$PWD
for m in 0 2; do
  rm -rf /tmp/pl$m; mkdir -p /tmp/pl$m
  (cd /tmp && CLV_WGRAD_TILE=$m SETS=s2,s3 TIMED=0 timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY --output-format csv -d /tmp/pl$m -o p -- python3 $R/tools/probes/wgrad_traffic.py > /tmp/pl$m/out.log 2>&1)
  echo "TILE=$m rc=$?"; tail -2 /tmp/pl$m/out.log
  python3 tools/pmc_kernel_means.py /tmp/pl$m wgrad
done
