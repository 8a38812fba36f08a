# usage: ab3.sh "<env A>" "<env B>" "<env C>" ...  -- alternates the configurations twice on one box
for i in 1 2; do
  for cfg in "$@"; do
    r=$(env $cfg python bench.py --no-kernel-timing --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$cfg] $r"
  done
done
