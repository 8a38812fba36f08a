echo "== atomics"; python tools/probes/wgrad_bench.py 2>/dev/null | grep -E " s1| s2| s3|fusion" | cut -c1-100
echo "== partials+fold"; CLV_WGRAD_NOATOMIC=1 python tools/probes/wgrad_bench.py 2>/dev/null | grep -E " s1| s2| s3|fusion" | cut -c1-100
