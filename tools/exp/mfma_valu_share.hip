// Do fp32-input MFMAs and fp32 VALU instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// Two workgroups of 4 waves per CU (two waves per SIMD).  Wave role by workgroup parity: MODE bit 0 = even workgroups spin on
// v_mfma_f32_16x16x4_f32, bit 1 = odd workgroups spin on independent v_fma_f32 (8 chains) / v_add_u32 / v_mfma_f32_32x32x8f16.
// If the pipes are separate, time(both) ~= max(time(mfma only), time(valu only)); if they share the execution resource, the sum.
// Build: hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_valu_share.hip -o ab/mfma_valu_share   (run on the GPU box)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// KIND of the non-MFMA partner: 0 v_fma_f32, 1 v_add_u32 (integer), 2 v_mfma_f32_32x32x8_f16 (the 16-bit matrix pipe), 3 v_pk_fma_f32
template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters_mfma, int iters_other, float a, float b)
{
    const bool mfma_role = ((blockIdx.x >> 8) & 1) == 0;   // (blocks b, b + 256 share a CU: the dispatcher deals blocks round-robin over the 8 XCDs, then over the 32 CUs of an XCD; with b & 1 the two roles land on different XCDs)
    float s = 0.f;
    if (mfma_role) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters_mfma; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 0) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = a * (float)(i + threadIdx.x);
        for (int it = 0; it < iters_other; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        for (int i = 0; i < 8; ++i) s += v[i];
    } else if (KIND == 1) {
        unsigned v[8];
        for (int i = 0; i < 8; ++i) v[i] = (unsigned)(i + threadIdx.x);
        const unsigned k = (unsigned)iters_other | 1u;
        for (int it = 0; it < iters_other; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(k));
        for (int i = 0; i < 8; ++i) s += (float)v[i];
    } else if (KIND == 2) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        const f16x4 ha = {(_Float16)a, (_Float16)a, (_Float16)b, (_Float16)b};
        for (int it = 0; it < iters_other; ++it)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(ha, ha, acc[i], 0, 0, 0);
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][5];
    } else {
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 v[8];
        for (int i = 0; i < 8; ++i) v[i] = v2{a * (float)i, b + (float)threadIdx.x};
        const v2 bb = {b, b}, aa = {a, a};
        for (int it = 0; it < iters_other; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(bb), "v"(aa));
        for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    }
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
float run(float* d, int im, int io)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<KIND>, dim3(512), dim3(256), 0, 0, d, im, io, 1.0f, 0.5f);     // 2 workgroups per CU: one of each role
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}

int main()
{
    float* d; hipMalloc(&d, 4);
    const int IM = 20000;                                         // 160 000 MFMAs per wave ~ 2.1 ms at 32 cycles each
    const char* names[4] = {"v_fma_f32", "v_add_u32", "v_mfma_f32_32x32x8_f16", "v_pk_fma_f32"};
    const int IO[4] = {40000, 40000, 20000, 40000};
    printf("one MFMA wave + one partner wave per SIMD; ms for: mfma alone | partner alone | both\n");
    { float a = run<0>(d, IM, 0), b = run<0>(d, 0, IO[0]), c = run<0>(d, IM, IO[0]); printf("%-24s %.3f | %.3f | %.3f   (max %.3f, sum %.3f)\n", names[0], a, b, c, a > b ? a : b, a + b); }
    { float a = run<1>(d, IM, 0), b = run<1>(d, 0, IO[1]), c = run<1>(d, IM, IO[1]); printf("%-24s %.3f | %.3f | %.3f   (max %.3f, sum %.3f)\n", names[1], a, b, c, a > b ? a : b, a + b); }
    { float a = run<2>(d, IM, 0), b = run<2>(d, 0, IO[2]), c = run<2>(d, IM, IO[2]); printf("%-24s %.3f | %.3f | %.3f   (max %.3f, sum %.3f)\n", names[2], a, b, c, a > b ? a : b, a + b); }
    { float a = run<3>(d, IM, 0), b = run<3>(d, 0, IO[3]), c = run<3>(d, IM, IO[3]); printf("%-24s %.3f | %.3f | %.3f   (max %.3f, sum %.3f)\n", names[3], a, b, c, a > b ? a : b, a + b); }
    return 0;
}
