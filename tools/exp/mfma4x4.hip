// Probe of the v_mfma_f32_4x4x1_16B_f32 operand layout (tuning experiment, not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float* a, const float* b, float* d)
{
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}
int main()
{
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024);
    for (int mode = 0; mode < 2; ++mode) {
        for (int l = 0; l < 64; ++l) { ha[l] = mode == 0 ? (float)l : 1.f; hb[l] = mode == 0 ? 1.f : (float)l; }
        hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
        hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
        printf("mode %d (%s supplies lane id)\n", mode, mode == 0 ? "A" : "B");
        for (int l = 0; l < 64; ++l) printf("lane %2d: %3.0f %3.0f %3.0f %3.0f%s", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3], (l % 4 == 3) ? "\n" : " | ");
    }
    return 0;
}
