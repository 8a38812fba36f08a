# usage: run_ablate.sh <epi> <tile> [<tile> ...]
epi=${1:-1}; shift
for t in "$@"; do for f in tools/probes/bin/gemm_ablate_*; do echo "== $(basename $f) epi=$epi tile=$t"; CLV_GEMM_TILE=$t timeout 120 $f $epi; done; done
