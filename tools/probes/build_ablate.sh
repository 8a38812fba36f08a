# builds tools/probes/bin/gemm_ablate_<variant> for each -D set given as arguments (quoted), plus "full"
mkdir -p tools/probes/bin; rm -f tools/probes/bin/gemm_ablate_*
for v in "" "$@"; do n=$(echo "$v" | sed 's/-DGN_ABL_//g; s/ //g'); /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Wno-unused-result -I clover_amd/csrc $v tools/probes/gemm_ablate.cpp -o tools/probes/bin/gemm_ablate_${n:-full} 2>&1 | grep -E "error"; done; ls tools/probes/bin/
