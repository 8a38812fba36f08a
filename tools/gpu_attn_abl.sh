#!/bin/bash
set -u
CLV_ATTN_BWD_ONE=0 python tools/probes/attn_one_bench.py 2>&1 | tail -1
python tools/probes/attn_one_bench.py 2>&1 | tail -1
for m in "$@"; do CLOVER_LIB_PATH=$PWD/tools/probes/bin/libclover_abl$m.so python tools/probes/attn_one_bench.py 2>&1 | tail -1; done
