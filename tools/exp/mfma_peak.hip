// Sustained fp32-input MFMA rate of this GPU (pure register loop, no memory traffic): the practical ceiling for
// the conv kernels.  Nominal: 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz = 157.3 TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void spin(float* out, int iters, float a, float b)
{
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}
int main()
{
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode)
        for (int wpc = 1; wpc <= 4; wpc *= 2) {                 // workgroups (4 waves) per CU
            const int iters = 20000, blocks = 256 * wpc;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
                else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)blocks * 4 * iters * 8 * (mode == 0 ? 2048.0 : 512.0);
            printf("%s  %d workgroups/CU: %.1f TFLOP/s (%.2f ms)\n", mode == 0 ? "16x16x4f32" : "4x4x1f32 ", wpc, flop / ms / 1e9, ms);
        }
    return 0;
}
