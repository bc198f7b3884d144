// L2 -> LDS staging bandwidth of one CU through LDS-DMA (buffer_load_dwordx4 ... lds), the path conv_wino2_kernel stages its K
// chunks on.  Every workgroup (4 waves) copies ITS OWN region of `kb` KiB (L2-resident after the first pass: footprint <= 24 MB)
// into LDS `iters` times; mode 0: free running (one vmcnt(0) per pass), mode 1: + a workgroup barrier per pass (the chunk loop of
// the Winograd conv), mode 2: every workgroup reads the SAME region (L1 / one L2 line set).  Prints bytes per clock per CU at
// WGs/CU = 1, 2, 4.       hipcc --offload-arch=gfx950 -O3 tools/exp/dma_bw.hip -o gpurun_out/dma_bw && gpurun_out/dma_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void dma_kernel(const float* __restrict__ src, int kb, int iters, int mode, float* sink)
{
    extern __shared__ __align__(16) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long a = (unsigned long long)(src + (size_t)(mode == 2 ? 0 : blockIdx.x) * kb * 256);
    const u32x4 rs = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                      (unsigned)kb * 1024u, 0x00020000u};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const int ni = kb / 4;                                  // 1 KiB instructions per wave per pass
    for (int it = 0; it < iters; ++it) {
        for (int i = 0; i < ni; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(i * 4 + wave) * 1024u);
            const unsigned vo = (unsigned)((i * 4 + wave) * 1024 + lane * 16);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(vo), "s"(rs), "s"(dst) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode == 1) __syncthreads();
    }
    __syncthreads();
    if (sink && threadIdx.x == 0) sink[blockIdx.x] = lds[lane];
}

int main()
{
    const int iters = 400;
    float* src; float* sink;
    hipMalloc(&src, (size_t)1024 * 40 * 1024); hipMalloc(&sink, 4096 * 4);
    hipMemset(src, 1, (size_t)1024 * 40 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int kb : {8, 20, 32})
            for (int wpc : {1, 2, 4}) {
                const int wgs = 256 * wpc;
                hipFuncSetAttribute((const void*)dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
                hipLaunchKernelGGL(dma_kernel, dim3(wgs), dim3(256), 40 * 1024, 0, src, kb, 10, mode, sink);
                hipEventRecord(e0);
                hipLaunchKernelGGL(dma_kernel, dim3(wgs), dim3(256), 40 * 1024, 0, src, kb, iters, mode, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)wgs * kb * 1024.0 * iters;
                printf("mode %d  %2d KiB per pass  %d WG/CU: %7.1f us  %6.2f TB/s  = %5.1f B/clk/CU at 2.4 GHz (%.0f clk per pass)\n", mode, kb, wpc,
                       ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / 256.0 / 2.4e9, ms * 1e-3 * 2.4e9 / iters);
            }
    return 0;
}
