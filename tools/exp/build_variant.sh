#!/bin/bash
# variant build of the library for same-box A/B runs (ab/ is git-ignored, travels to the GPU box with gpurun):
#   tools/exp/build_variant.sh <name> [extra hipcc flags ...]   ->  ab/libpggan_<name>.so   (use with PGGAN_HIP_LIB=ab/libpggan_<name>.so)
# per-file flags of __graft_entry__.FILE_FLAGS are applied as in the product build
cd "$(dirname "$0")/../.." && mkdir -p ab/obj_$1 || exit 1
name=$1; shift
pids=()
for f in pggan-pytorch_amd/csrc/*.hip; do
  b=$(basename $f .hip); extra=""
  [ $b = conv_wino_wgrad ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude $extra "$@" -c $f -o ab/obj_$name/$b.o 2>/dev/null &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p || { echo "compile failed"; exit 1; }; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC ab/obj_$name/*.o -o ab/libpggan_$name.so && echo ab/libpggan_$name.so
