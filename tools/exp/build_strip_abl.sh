#!/bin/bash
# Ablation builds of the row-streaming conv (csrc/conv_strip.hip, -DPG_STRIP_ABL=bits: 1 no DMA after the prologue, 2 no stores,
# 4 one tap of nine) as ab/libpggan_abl<bits>.so; select one with PGGAN_HIP_LIB=ab/... python tools/sweeps/bench_strip.py.
# (ab/ is git-ignored; delete the libraries afterwards: gpurun ships the directory.)
cd "$(dirname "$0")/../.." && mkdir -p ab && python __graft_entry__.py > /dev/null || exit 1
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -DPG_STRIP_ABL=$a -c pggan-pytorch_amd/csrc/conv_strip.hip -o build/obj/abl_$a.o || exit 1
  objs=$(ls build/obj/*.o | grep -v -e abl_ -e conv_strip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/obj/abl_$a.o -o ab/libpggan_abl$a.so || exit 1
done
ls -la ab
