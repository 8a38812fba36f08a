# builds tools/probes/bin/libclover_<name>.so with gemm_wgrad.hip compiled under the given -D sets: build_wg_variants.sh name1 "-DX" name2 "-DY -DZ" ...
set -e
mkdir -p tools/probes/bin
make -C clover_amd/csrc -j8 > /dev/null
while [ $# -ge 2 ]; do
  n=$1; d=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics $d -c clover_amd/csrc/gemm_wgrad.hip -o tools/probes/bin/gemm_wgrad_$n.o
  objs=$(ls clover_amd/csrc/build/*.o | grep -v gemm_wgrad.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs tools/probes/bin/gemm_wgrad_$n.o -o tools/probes/bin/libclover_$n.so
  echo built $n
done
