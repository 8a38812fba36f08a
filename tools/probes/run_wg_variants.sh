python tools/probes/wgrad_group.py 2>&1 | grep problems
for n in "$@"; do echo "== $n"; CLOVER_LIB_PATH=$PWD/tools/probes/bin/libclover_$n.so python tools/probes/wgrad_group.py 2>&1 | grep problems; done
