#!/bin/bash
# traced build of the library (ab/ is git-ignored, travels to the GPU box with gpurun)
cd "$(dirname "$0")/../.." && mkdir -p ab && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -shared -Iinclude -DPG_WINO_TRACE -fno-slp-vectorize pggan-pytorch_amd/csrc/*.hip -o ab/libpggan_trace.so
