set -u
for t in 32 64 128 256; do echo "target $t"; CLV_WGRAD_GROUP_TARGET=$t python tools/probes/wgrad_group.py; done
for r in 784 1568 3136 6272; do echo "rows $r"; CLV_WGRAD_GROUP_ROWS=$r python tools/probes/wgrad_group.py; done
echo bigfirst; ORDER=bigfirst python tools/probes/wgrad_group.py
