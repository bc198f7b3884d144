// The two steps either side of the hot path, as HBM-streaming kernels so that the
// real batch never has to be prepared by host workers and sample grids never leave the device as fp32:
//   pg_real_prepare_u8 : uint8 dataset image -> fade-in blend with its 2x2 box-filtered copy -> dynamic range -> fp32
//                        (reference dataset.py:54-67 __getitem__, :109-113 alpha_fade, utils.py:24-30)
//   pg_image_grid_u8   : fp32 G output -> nearest upsample -> tiled grid -> dynamic range -> round/clip -> uint8 HWC
//                        (reference output_postprocess.py:35-62, utils.py:33-53)
// Arithmetic mirrors numpy's: the input step is evaluated in fp64 (uint8 -> float64 promotion in the reference)
// and rounded once to fp32; the output step in fp32 with round-half-even -> both are bit-exact.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pggan_hip.h"

// numpy never fuses a multiply with the following add; hipcc contracts them into FMAs by default.  Bit-exact
// parity with the reference needs the unfused sequence, for the whole translation unit:
#pragma clang fp contract(off)

namespace {

// An empty asm that 'modifies' the value: the product is materialised in a register, so the compiler cannot
// contract it with the following add into an FMA (numpy rounds the product and the sum separately).
__device__ __forceinline__ double rounded(double v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ float rounded(float v) { asm volatile("" : "+v"(v)); return v; }

// one thread per 2x2 block of one (n,c) plane
__global__ __launch_bounds__(256) void real_prepare_u8_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                              long long planes, int H, int W, int fade, double one_minus_alpha,
                                                              double min_in, double scale, double min_out, int rescale)
{
    const int H2 = H >> 1, W2 = W >> 1;
    const long long total = planes * H2 * W2;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int w2 = (int)(idx % W2);
        const long long r = idx / W2;
        const int h2 = (int)(r % H2);
        const long long pl = r / H2;
        const size_t base = ((size_t)pl * H + 2 * h2) * W + 2 * w2;
        double v[4] = {(double)in[base], (double)in[base + 1], (double)in[base + W], (double)in[base + W + 1]};
        if (fade) {
            const double t = (v[0] + v[1] + v[2] + v[3]) / 4.0;          // reshape(...).mean((2,4)) of uint8 -> float64
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] + rounded((t - v[i]) * one_minus_alpha);   // dataset.py:112
            // (explicit round-to-nearest mul and add: numpy does not fuse them, hipcc's default contraction would)
        }
        if (rescale) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = rounded((v[i] - min_in) * scale) + min_out;     // utils.py:29
        }
        out[base] = (float)v[0]; out[base + 1] = (float)v[1]; out[base + W] = (float)v[2]; out[base + W + 1] = (float)v[3];
    }
}

// one thread per output pixel (all channels): grid [GH*h*up][GW*w*up][C] uint8
__global__ __launch_bounds__(256) void image_grid_u8_kernel(const float* __restrict__ img, uint8_t* __restrict__ grid,
                                                            int n, int C, int h, int w, int up, int grid_w, int grid_h,
                                                            float min_in, float scale, float min_out, int rescale)
{
    const int oh = h * up, ow = w * up;
    const int GH = grid_h * oh, GW = grid_w * ow;
    const long long total = (long long)GH * GW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % GW), y = (int)(idx / GW);
        const int gi = (y / oh) * grid_w + (x / ow);
        for (int c = 0; c < C; ++c) {
            float v = 0.f;                                                   // np.zeros grid for unused cells
            if (gi < n) v = img[(((size_t)gi * C + c) * h + (y % oh) / up) * w + (x % ow) / up];
            if (rescale) v = rounded((v - min_in) * scale) + min_out;   // utils.py:29 (float32, unfused)
            v = rintf(v);                                                    // np.round: half to even
            v = fminf(fmaxf(v, 0.f), 255.f);
            grid[(size_t)idx * C + c] = (uint8_t)v;
        }
    }
}

inline int grid_for(long long total, int block = 256, int cap = 4096)
{
    long long g = (total + block - 1) / block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

// One pyramid level (dataset.py:243-250): out[y][x] = uint8(clip(rint((p00 + p01 + p10 + p11) / 4))) with the four
// samples at (y*st + {0,1}, x*st + {0,1}), st = 2^depthdiff.  Sums of four bytes are exact in fp32, /4 is exact,
// rintf is round-half-even like np.round -> bit-exact.
__global__ __launch_bounds__(256) void pyramid_level_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                               long long planes, int H, int W, int st, float lo, float hi)
{
    const int Ho = H / st, Wo = W / st;
    const long long total = planes * Ho * Wo;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo);
        long long r = i / Wo;
        const int y = (int)(r % Ho); const long long pl = r / Ho;
        const uint8_t* p = in + (pl * H + (long long)y * st) * W + (long long)x * st;
        float v = (((float)p[0] + (float)p[1]) + (float)p[W]) + (float)p[W + 1];
        v = rintf(v * 0.25f);
        v = fminf(fmaxf(v, lo), hi);
        out[i] = (uint8_t)v;
    }
}

}  // namespace

extern "C" int pg_pyramid_level_u8(const uint8_t* in, uint8_t* out, int64_t planes, int H, int W, int depthdiff,
                                   float min_in, float max_in, pg_stream_t stream)
{
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0 || depthdiff < 1 || depthdiff > 16) return PG_E_ARG;
    const int st = 1 << depthdiff;
    if ((H % st) || (W % st)) return PG_E_ALIGN;
    hipLaunchKernelGGL(pyramid_level_u8_kernel, dim3(grid_for(planes * (H / st) * (W / st))), dim3(256), 0, (hipStream_t)stream,
                       in, out, (long long)planes, H, W, st, min_in, max_in);
    return (int)hipGetLastError();
}

extern "C" int pg_real_prepare_u8(const uint8_t* in, float* out, int64_t planes, int H, int W, double alpha,
                                  double min_in, double max_in, double min_out, double max_out, pg_stream_t stream)
{
    if (!in || !out || planes <= 0 || H <= 0 || W <= 0) return PG_E_ARG;
    if ((H | W) & 1) return PG_E_ALIGN;
    const int rescale = !(min_in == min_out && max_in == max_out);          // utils.py:25 `if range_in != range_out`
    const double scale = rescale ? (max_out - min_out) / (max_in - min_in) : 1.0;
    const int fade = alpha < 1.0;                                             // dataset.py:62
    hipLaunchKernelGGL(real_prepare_u8_kernel, dim3(grid_for(planes * (H / 2) * (W / 2))), dim3(256), 0, (hipStream_t)stream,
                       in, out, (long long)planes, H, W, fade, 1.0 - alpha, min_in, scale, min_out, rescale);
    return (int)hipGetLastError();
}

extern "C" int pg_image_grid_u8(const float* img, uint8_t* grid, int n, int C, int h, int w, int up,
                                float min_in, float max_in, pg_stream_t stream)
{
    if (!img || !grid || n <= 0 || C <= 0 || h <= 0 || w <= 0 || up <= 0) return PG_E_ARG;
    int grid_w = 1; while (grid_w * grid_w < n) ++grid_w;                    // max(ceil(sqrt(count)), 1)
    const int grid_h = (n - 1) / grid_w + 1;
    const int rescale = !(min_in == 0.f && max_in == 255.f);
    const float scale = rescale ? (float)((255.0 - 0.0) / ((double)max_in - (double)min_in)) : 1.f;
    hipLaunchKernelGGL(image_grid_u8_kernel, dim3(grid_for((long long)grid_h * h * up * grid_w * w * up)), dim3(256), 0,
                       (hipStream_t)stream, img, grid, n, C, h, w, up, grid_w, grid_h, min_in, scale, 0.f, rescale);
    return (int)hipGetLastError();
}
