set -u
cd $GRAFT_REPO_ROOT
run() { python $1 --steps 40 --warmup 10 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
  echo "old rep$rep: $(run tools/probes/bin/old_tree/bench.py)"
  echo "new rep$rep: $(run bench.py)"
  echo "new-olddbias rep$rep: $(CLV_DBIAS_GROUPS_PER_SLICE=16 CLOVER_DEFER_DBIAS=0 run bench.py)"
done
