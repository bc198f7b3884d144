#!/bin/bash
# ablation builds of the library for the row-streaming Winograd conv (ab/ is git-ignored, travels to the GPU box with gpurun):
# only conv_wino_strip.hip is recompiled with -DPG_WS_ABL=<bits>, the other objects come from build/obj
cd "$(dirname "$0")/../.." && mkdir -p ab
for a in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -Iinclude -DPG_WS_ABL=$a -c pggan-pytorch_amd/csrc/conv_wino_strip.hip -o ab/ws_abl$a.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $(ls build/obj/*.o | grep -v conv_wino_strip) ab/ws_abl$a.o -o ab/libpggan_abl$a.so ) &
done
wait
