# usage: ab.sh "<env A>" "<env B>" [bench args]   -- alternates A/B three times on one box
A="$1"; B="$2"; shift 2
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    r=$(env $cfg python bench.py --no-kernel-timing --steps 40 --warmup 8 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])")
    echo "[$cfg] $r"
  done
done
