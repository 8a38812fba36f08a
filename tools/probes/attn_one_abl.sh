#!/bin/bash
# Ablation builds of the one-kernel attention backward (ONE_ABL bit mask, see attention.hip) -> tools/probes/bin/libclover_abl<mask>.so;
# run on the GPU box:  for m in 0 1 2 ...; do CLOVER_LIB_PATH=tools/probes/bin/libclover_abl$m.so python tools/probes/attn_one_bench.py; done
set -u
cd "$(dirname "$0")/../.."
mkdir -p tools/probes/bin /tmp/abl
for m in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DONE_ABL=$m -c clover_amd/csrc/attention.hip -o /tmp/abl/attention_$m.o 2>/dev/null &&
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/abl/attention_$m.o $(ls clover_amd/csrc/build/*.o | grep -v attention.o) -o tools/probes/bin/libclover_abl$m.so && echo built $m ) &
done
wait
