// fromRGB / toRGB: the 1x1 convolutions at the image boundary (C_img = 1..4 channels).
// These are HBM-streaming kernels (arithmetic intensity ~1 FLOP/B): no MFMA, coalesced float4
// traffic on the NHWC feature side, row-contiguous traffic on the NCHW image side, the
// fade-in blend / 2x2 pooling of the image fused so the image is touched exactly once.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <stdint.h>
#include "pggan_hip.h"

namespace {

constexpr int MAXC = 4;

__device__ __forceinline__ float img_fetch(const float* img, int n, int c, int h, int w, int C, int H, int W, int pool)
{
    if (!pool) return img[(((size_t)n * C + c) * H + h) * W + w];
    const int H2 = 2 * H, W2 = 2 * W;
    const float* p = img + (((size_t)n * C + c) * H2 + 2 * h) * W2 + 2 * w;
    return ((p[0] + p[1]) + (p[W2] + p[W2 + 1])) * 0.25f;
}

// thread -> (pixel, 4 couts).  y[pix][co4] float4 stores are fully coalesced.
__global__ __launch_bounds__(256) void fromrgb_fwd_kernel(
    const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ mask, float* __restrict__ y,
    int N, int C, int H, int W, int Cout, int pool, float scale, float slope, float mask_slope, int mbytes, unsigned char* __restrict__ ysigns)
{
    const int c4n = Cout >> 2;
    const size_t total = (size_t)N * H * W * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t pix = idx / c4n;
        const int wv = (int)(pix % W);
        const size_t r = pix / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float xin[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) xin[c] = c < C ? img_fetch(img, n, c, h, wv, C, H, W, pool) : 0.f;
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float* wr = w + (size_t)(4 * c4 + j) * C;
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) a = fmaf(xin[c], wr[c], a);
            o[j] = __fmul_rn(a, scale);          // (explicit roundings, as in fromrgb_fwd_pix_kernel)
        }
        const size_t off = pix * Cout + 4 * c4;
        if (mask) {
            if (mbytes) {
                const unsigned char b = reinterpret_cast<const unsigned char*>(mask)[off >> 2];
                o[0] *= (b & 1) ? 1.f : mask_slope; o[1] *= (b & 2) ? 1.f : mask_slope;
                o[2] *= (b & 4) ? 1.f : mask_slope; o[3] *= (b & 8) ? 1.f : mask_slope;
            } else {
                const float4 mk = *reinterpret_cast<const float4*>(mask + off);
                o[0] *= mk.x > 0.f ? 1.f : mask_slope; o[1] *= mk.y > 0.f ? 1.f : mask_slope;
                o[2] *= mk.z > 0.f ? 1.f : mask_slope; o[3] *= mk.w > 0.f ? 1.f : mask_slope;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = __fadd_rn(o[j], bias ? bias[4 * c4 + j] : 0.f);
                o[j] = v > 0.f ? v : v * slope;
            }
            if (ysigns) ysigns[off >> 2] = (unsigned char)((o[0] > 0.f ? 1 : 0) | (o[1] > 0.f ? 2 : 0) | (o[2] > 0.f ? 4 : 0) | (o[3] > 0.f ? 8 : 0));
        }
        *reinterpret_cast<float4*>(y + off) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// thread -> pixel variant for the 8-channel stage (and 16 channels without a mask): all CO outputs of a pixel from one thread,
// the image read once per pixel.  Measured (tools/sweeps/bench_rgb_stream.py, n9 @1024^2, 8 channels): 105 us vs 239 us for the
// (pixel, 4 couts) mapping, which is the better one from 16 masked channels on.
template <int CO>
__global__ __launch_bounds__(256) void fromrgb_fwd_pix_kernel(
    const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ mask, float* __restrict__ y,
    int N, int C, int H, int W, int pool, float scale, float slope, float mask_slope, int mbytes, unsigned char* __restrict__ ysigns)
{
    const unsigned total = (unsigned)N * H * W, HW = (unsigned)H * W;
    for (unsigned pix = blockIdx.x * 256u + threadIdx.x; pix < total; pix += gridDim.x * 256u) {
        const unsigned n = pix / HW, hw = pix - n * HW;
        const int h = (int)(hw / (unsigned)W), wv = (int)(hw - (unsigned)h * W);
        float xin[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) xin[c] = c < C ? img_fetch(img, (int)n, c, h, wv, C, H, W, pool) : 0.f;
        const size_t off = (size_t)pix * CO;
#pragma unroll
        for (int c4 = 0; c4 < CO / 4; ++c4) {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* wr = w + (4 * c4 + j) * C;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c) if (c < C) a = fmaf(xin[c], wr[c], a);
                o[j] = __fmul_rn(a, scale);      // (explicit roundings here and at the bias: conv_strip_rgb_kernel restates this pixel bit for bit)
            }
            if (mask) {
                if (mbytes) {
                    const unsigned char b = reinterpret_cast<const unsigned char*>(mask)[(off >> 2) + c4];
                    o[0] *= (b & 1) ? 1.f : mask_slope; o[1] *= (b & 2) ? 1.f : mask_slope;
                    o[2] *= (b & 4) ? 1.f : mask_slope; o[3] *= (b & 8) ? 1.f : mask_slope;
                } else {
                    const float4 mk = *reinterpret_cast<const float4*>(mask + off + 4 * c4);
                    o[0] *= mk.x > 0.f ? 1.f : mask_slope; o[1] *= mk.y > 0.f ? 1.f : mask_slope;
                    o[2] *= mk.z > 0.f ? 1.f : mask_slope; o[3] *= mk.w > 0.f ? 1.f : mask_slope;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = __fadd_rn(o[j], bias ? bias[4 * c4 + j] : 0.f);
                    o[j] = v > 0.f ? v : v * slope;
                }
                if (ysigns) ysigns[(off >> 2) + c4] = (unsigned char)((o[0] > 0.f ? 1 : 0) | (o[1] > 0.f ? 2 : 0) | (o[2] > 0.f ? 4 : 0) | (o[3] > 0.f ? 8 : 0));
            }
            *reinterpret_cast<float4*>(y + off + 4 * c4) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// thread -> pixel; loops over Cout in float4 steps.
__global__ __launch_bounds__(256) void fromrgb_bwd_data_kernel(
    const float* __restrict__ gz, const float* __restrict__ w, float* __restrict__ gimg,
    int N, int C, int H, int W, int Cout, int pool, int accumulate, float scale)
{
    extern __shared__ float wl[];                 // [Cout][C]
    for (int i = threadIdx.x; i < Cout * C; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const size_t total = (size_t)N * H * W;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
        const int wv = (int)(pix % W);
        const size_t r = pix / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float a[MAXC] = {0.f, 0.f, 0.f, 0.f};
        const float* g = gz + pix * Cout;
        for (int co = 0; co < Cout; co += 4) {
            const float4 gv = *reinterpret_cast<const float4*>(g + co);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) {
                a[c] = fmaf(gv.x, wl[(co + 0) * C + c], a[c]);
                a[c] = fmaf(gv.y, wl[(co + 1) * C + c], a[c]);
                a[c] = fmaf(gv.z, wl[(co + 2) * C + c], a[c]);
                a[c] = fmaf(gv.w, wl[(co + 3) * C + c], a[c]);
            }
        }
        for (int c = 0; c < C; ++c) {
            if (!pool) {
                float* o = gimg + (((size_t)n * C + c) * H + h) * W + wv;
                const float v = a[c] * scale;
                *o = accumulate ? *o + v : v;
            } else {
                const int H2 = 2 * H, W2 = 2 * W;
                float* o = gimg + (((size_t)n * C + c) * H2 + 2 * h) * W2 + 2 * wv;
                const float v = a[c] * scale * 0.25f;
                if (accumulate) { o[0] += v; o[1] += v; o[W2] += v; o[W2 + 1] += v; }
                else { o[0] = v; o[1] = v; o[W2] = v; o[W2 + 1] = v; }
            }
        }
    }
}

// dw[co][c] += scale * sum_pix gz[pix][co]*img[pix][c]; db[co] += sum_pix gz[pix][co].
// Block: CPB couts per pass x PL pixel lanes; per-block partials reduced in LDS, one atomic each.
__global__ __launch_bounds__(256) void fromrgb_wgrad_kernel(
    const float* __restrict__ gz, const float* __restrict__ img, float* __restrict__ dw, float* __restrict__ db,
    int N, int C, int H, int W, int Cout, int pool, float scale, int pix_per_block)
{
    __shared__ float red[256 * (MAXC + 1)];
    const int cpb = Cout < 256 ? Cout : 256;          // couts handled per pass (Cout is a multiple of 4)
    const int PL = 256 / cpb;                          // pixel lanes
    const int col = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const size_t total = (size_t)N * H * W;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    const size_t p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
    for (int cbase = 0; cbase < Cout; cbase += cpb) {
        const int co = cbase + col;
        float a[MAXC + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (pl < PL && co < Cout) {
            for (size_t pix = p0 + pl; pix < p1; pix += PL) {
                const int wv = (int)(pix % W);
                const size_t r = pix / W;
                const int h = (int)(r % H), n = (int)(r / H);
                const float g = gz[pix * Cout + co];
#pragma unroll
                for (int c = 0; c < MAXC; ++c) if (c < C) a[c] = fmaf(g, img_fetch(img, n, c, h, wv, C, H, W, pool), a[c]);
                a[MAXC] += g;
            }
        }
#pragma unroll
        for (int c = 0; c <= MAXC; ++c) red[c * 256 + threadIdx.x] = a[c];
        __syncthreads();
        if (pl == 0 && co < Cout) {
#pragma unroll
            for (int c = 0; c <= MAXC; ++c) {
                float s = 0.f;
                for (int q = 0; q < PL; ++q) s += red[c * 256 + q * cpb + col];
                if (c < C) atomicAdd(dw + (size_t)co * C + c, s * scale);
                else if (c == MAXC && db) atomicAdd(db + co, s);
            }
        }
        __syncthreads();
    }
}

// Small-Cout variant (CO <= 32, the >=128^2 stages): one thread per pixel keeps all CO x (C+1) partial sums in
// registers, reads its pixel's gz row with float4 loads (coalesced across the wave) and the image planes
// row-contiguously; wave shuffle + LDS reduction, one atomic per output and workgroup.
template <int CO>
__global__ __launch_bounds__(256) void fromrgb_wgrad_small_kernel(
    const float* __restrict__ gz, const float* __restrict__ img, float* __restrict__ dw, float* __restrict__ db,
    int N, int C, int H, int W, int pool, float scale)
{
    constexpr int NV = CO * (MAXC + 1);
    __shared__ float red[4][NV];
    float acc[CO][MAXC + 1];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int c = 0; c <= MAXC; ++c) acc[o][c] = 0.f;
    const size_t total = (size_t)N * H * W;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
        const int wv = (int)(pix % W);
        const size_t r = pix / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float xin[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) xin[c] = c < C ? img_fetch(img, n, c, h, wv, C, H, W, pool) : 0.f;
        const float4* g4 = reinterpret_cast<const float4*>(gz + pix * CO);
#pragma unroll
        for (int o4 = 0; o4 < CO / 4; ++o4) {
            const float4 g = g4[o4];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c) acc[4 * o4 + j][c] = fmaf(gv[j], xin[c], acc[4 * o4 + j][c]);
                acc[4 * o4 + j][MAXC] += gv[j];
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int c = 0; c <= MAXC; ++c) {
            float v = acc[o][c];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
            if (lane == 0) red[wave][o * (MAXC + 1) + c] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) {
        const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const int o = i / (MAXC + 1), c = i % (MAXC + 1);
        if (c < C) atomicAdd(dw + (size_t)o * C + c, v * scale);
        else if (c == MAXC && db) atomicAdd(db + o, v);
    }
}

// thread -> pixel: out[n,c,h,w] for all c; reads the pixel's Cin features with float4 loads.
__global__ __launch_bounds__(256) void torgb_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ prev, float* __restrict__ out,
    int N, int C, int H, int W, int Cin, float scale, float out_mul, float prev_mul)
{
    extern __shared__ float wl[];                 // [C][Cin]
    for (int i = threadIdx.x; i < C * Cin; i += blockDim.x) wl[i] = w[i];
    __syncthreads();
    const size_t total = (size_t)N * H * W;
    for (size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (size_t)gridDim.x * blockDim.x) {
        const int wv = (int)(pix % W);
        const size_t r = pix / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float a[MAXC] = {0.f, 0.f, 0.f, 0.f};
        const float* xp = x + pix * Cin;
        for (int ci = 0; ci < Cin; ci += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + ci);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) {
                const float* wr = wl + c * Cin + ci;
                a[c] = fmaf(xv.x, wr[0], a[c]); a[c] = fmaf(xv.y, wr[1], a[c]);
                a[c] = fmaf(xv.z, wr[2], a[c]); a[c] = fmaf(xv.w, wr[3], a[c]);
            }
        }
        for (int c = 0; c < C; ++c) {
            float v = (a[c] * scale + (bias ? bias[c] : 0.f)) * out_mul;
            if (prev) v += prev_mul * prev[(((size_t)n * C + c) * (H >> 1) + (h >> 1)) * (W >> 1) + (wv >> 1)];
            out[(((size_t)n * C + c) * H + h) * W + wv] = v;
        }
    }
}

__device__ __forceinline__ float g_fetch(const float* g, int n, int c, int h, int w, int C, int H, int W, int down)
{
    if (!down) return g[(((size_t)n * C + c) * H + h) * W + w];
    const int H2 = 2 * H, W2 = 2 * W;
    const float* p = g + (((size_t)n * C + c) * H2 + 2 * h) * W2 + 2 * w;
    return (p[0] + p[1]) + (p[W2] + p[W2 + 1]);
}

// thread -> (pixel, 4 cins): gx[pix][ci4] = mul_scale * sum_c g[pix,c]*w[c][ci]
__global__ __launch_bounds__(256) void torgb_bwd_data_kernel(
    const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gx,
    int N, int C, int H, int W, int Cin, int down, float mul_scale)
{
    const int c4n = Cin >> 2;
    const size_t total = (size_t)N * H * W * c4n;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(idx % c4n);
        const size_t pix = idx / c4n;
        const int wv = (int)(pix % W);
        const size_t r = pix / W;
        const int h = (int)(r % H), n = (int)(r / H);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float gv = g_fetch(g, n, c, h, wv, C, H, W, down);
            const float4 wv4 = *reinterpret_cast<const float4*>(w + (size_t)c * Cin + 4 * c4);
            o[0] = fmaf(gv, wv4.x, o[0]); o[1] = fmaf(gv, wv4.y, o[1]);
            o[2] = fmaf(gv, wv4.z, o[2]); o[3] = fmaf(gv, wv4.w, o[3]);
        }
        *reinterpret_cast<float4*>(gx + pix * Cin + 4 * c4) =
            make_float4(o[0] * mul_scale, o[1] * mul_scale, o[2] * mul_scale, o[3] * mul_scale);
    }
}

// Narrow layers on big maps (toRGB adjoint at 256^2 .. 1024^2): same mapping as above, 32-bit index math, compile-time CI.
template <int CI>
__global__ __launch_bounds__(256) void torgb_bwd_data_narrow_kernel(
    const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gx,
    int N, int C, int H, int W, int down, float mul_scale)
{
    constexpr unsigned C4N = CI / 4;
    const unsigned total = (unsigned)N * H * W * C4N, HW = (unsigned)H * W;
    for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < total; idx += gridDim.x * 256u) {
        const unsigned c4 = idx % C4N, pix = idx / C4N;
        const unsigned n = pix / HW, hw = pix - n * HW;
        const int h = (int)(hw / (unsigned)W), wv = (int)(hw - (unsigned)h * W);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < MAXC; ++c) if (c < C) {
            const float gv = g_fetch(g, (int)n, c, h, wv, C, H, W, down);
            const float4 wv4 = *reinterpret_cast<const float4*>(w + c * CI + 4 * c4);
            o[0] = fmaf(gv, wv4.x, o[0]); o[1] = fmaf(gv, wv4.y, o[1]);
            o[2] = fmaf(gv, wv4.z, o[2]); o[3] = fmaf(gv, wv4.w, o[3]);
        }
        *reinterpret_cast<float4*>(gx + (size_t)idx * 4) =
            make_float4(o[0] * mul_scale, o[1] * mul_scale, o[2] * mul_scale, o[3] * mul_scale);
    }
}

// thread -> pixel variant for 8 features (see fromrgb_fwd_pix_kernel).
// PNB: the adjoint of the block's (LeakyReLU -> PixelNorm) on top (network.py:44-52 after :32-41): the thread holds all CI
// features of its pixel, so  gx = r * (gh - y * mean_c(gh * y)) * lrelu'(y)  needs no second pass over the 1024^2 map.
template <int CI, bool PNB = false>
__global__ __launch_bounds__(256) void torgb_bwd_data_pix_kernel(
    const float* __restrict__ g, const float* __restrict__ w, float* __restrict__ gx,
    int N, int C, int H, int W, int down, float mul_scale,
    const float* __restrict__ pnb_y = nullptr, const float* __restrict__ pnb_r = nullptr, float slope = 1.f)
{
    const unsigned total = (unsigned)N * H * W, HW = (unsigned)H * W;
    for (unsigned pix = blockIdx.x * 256u + threadIdx.x; pix < total; pix += gridDim.x * 256u) {
        const unsigned n = pix / HW, hw = pix - n * HW;
        const int h = (int)(hw / (unsigned)W), wv = (int)(hw - (unsigned)h * W);
        float gv[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) gv[c] = c < C ? g_fetch(g, (int)n, c, h, wv, C, H, W, down) : 0.f;
        float4 ov[CI / 4];
#pragma unroll
        for (int c4 = 0; c4 < CI / 4; ++c4) {
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) {
                const float4 wv4 = *reinterpret_cast<const float4*>(w + c * CI + 4 * c4);
                o[0] = fmaf(gv[c], wv4.x, o[0]); o[1] = fmaf(gv[c], wv4.y, o[1]);
                o[2] = fmaf(gv[c], wv4.z, o[2]); o[3] = fmaf(gv[c], wv4.w, o[3]);
            }
            ov[c4] = make_float4(o[0] * mul_scale, o[1] * mul_scale, o[2] * mul_scale, o[3] * mul_scale);
        }
        if (PNB) {
            float4 yv[CI / 4];
            float dot = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < CI / 4; ++c4) {
                yv[c4] = *reinterpret_cast<const float4*>(pnb_y + (size_t)pix * CI + 4 * c4);
                dot += (ov[c4].x * yv[c4].x + ov[c4].y * yv[c4].y) + (ov[c4].z * yv[c4].z + ov[c4].w * yv[c4].w);
            }
            const float rr = pnb_r[pix], mean = dot / (float)CI;
#pragma unroll
            for (int c4 = 0; c4 < CI / 4; ++c4) {
                const float4 gq = ov[c4], y4 = yv[c4];
                ov[c4] = make_float4(rr * (gq.x - y4.x * mean) * (y4.x > 0.f ? 1.f : slope), rr * (gq.y - y4.y * mean) * (y4.y > 0.f ? 1.f : slope),
                                     rr * (gq.z - y4.z * mean) * (y4.z > 0.f ? 1.f : slope), rr * (gq.w - y4.w * mean) * (y4.w > 0.f ? 1.f : slope));
            }
        }
#pragma unroll
        for (int c4 = 0; c4 < CI / 4; ++c4) *reinterpret_cast<float4*>(gx + (size_t)pix * CI + 4 * c4) = ov[c4];
    }
}

// Narrow layers on big maps (the 256^2 .. 1024^2 toRGB: 8 / 16 / 32 features): thread -> pixel with all CI x C partial sums in
// registers, one shuffle + LDS fold per workgroup -- the per-(pixel, channel) thread mapping above spends its time on index math.
template <int CI>
__global__ __launch_bounds__(256) void torgb_wgrad_small_kernel(
    const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
    int N, int C, int H, int W, int down, float mul_scale, float mul)
{
    constexpr int NV = (CI + 1) * MAXC;
    __shared__ float red[4][NV];
    float acc[CI + 1][MAXC];                      // row CI: sum of g (bias)
#pragma unroll
    for (int i = 0; i <= CI; ++i)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[i][c] = 0.f;
    const unsigned total = (unsigned)N * H * W, HW = (unsigned)H * W;
    for (unsigned pix = blockIdx.x * 256u + threadIdx.x; pix < total; pix += gridDim.x * 256u) {
        const unsigned n = pix / HW, hw = pix - n * HW;
        const int h = (int)(hw / (unsigned)W), wv = (int)(hw - (unsigned)h * W);
        float gv[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) gv[c] = c < C ? g_fetch(g, (int)n, c, h, wv, C, H, W, down) : 0.f;
        const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)pix * CI);
#pragma unroll
        for (int i4 = 0; i4 < CI / 4; ++i4) {
            const float4 v = x4[i4];
            const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int c = 0; c < MAXC; ++c) acc[4 * i4 + j][c] = fmaf(gv[c], xv[j], acc[4 * i4 + j][c]);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[CI][c] += gv[c];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i <= CI; ++i)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            float v = acc[i][c];
#pragma unroll
            for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_xor(v, sft, 64);
            if (lane == 0) red[wave][i * MAXC + c] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < NV; i += 256) {
        const float v = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        const int ci = i / MAXC, c = i % MAXC;
        if (c >= C) continue;
        if (ci < CI) atomicAdd(dw + (size_t)c * CI + ci, v * mul_scale);
        else if (db) atomicAdd(db + c, v * mul);
    }
}

// dw[c][ci] += mul_scale * sum_pix g[pix,c]*x[pix,ci];  db[c] += mul * sum_pix g[pix,c]
__global__ __launch_bounds__(256) void torgb_wgrad_kernel(
    const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
    int N, int C, int H, int W, int Cin, int down, float mul_scale, float mul, int pix_per_block)
{
    __shared__ float red[256 * (MAXC + 1)];
    const int cpb = Cin < 256 ? Cin : 256;
    const int PL = 256 / cpb;
    const int col = threadIdx.x % cpb, pl = threadIdx.x / cpb;
    const size_t total = (size_t)N * H * W;
    const size_t p0 = (size_t)blockIdx.x * pix_per_block;
    const size_t p1 = p0 + pix_per_block < total ? p0 + pix_per_block : total;
    for (int cbase = 0; cbase < Cin; cbase += cpb) {
        const int ci = cbase + col;
        float a[MAXC + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
        if (pl < PL && ci < Cin) {
            for (size_t pix = p0 + pl; pix < p1; pix += PL) {
                const int wv = (int)(pix % W);
                const size_t r = pix / W;
                const int h = (int)(r % H), n = (int)(r / H);
                const float xv = x[pix * Cin + ci];
#pragma unroll
                for (int c = 0; c < MAXC; ++c) if (c < C) {
                    const float gv = g_fetch(g, n, c, h, wv, C, H, W, down);
                    a[c] = fmaf(gv, xv, a[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) red[c * 256 + threadIdx.x] = a[c];
        __syncthreads();
        if (pl == 0 && ci < Cin) {
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) {
                float s = 0.f;
                for (int q = 0; q < PL; ++q) s += red[c * 256 + q * cpb + col];
                atomicAdd(dw + (size_t)c * Cin + ci, s * mul_scale);
            }
        }
        __syncthreads();
    }
    // bias: db[c] += mul * sum over this block's pixels of g[pix,c]
    if (db) {
        float a[MAXC] = {0.f, 0.f, 0.f, 0.f};
        for (size_t pix = p0 + threadIdx.x; pix < p1; pix += 256) {
            const int wv = (int)(pix % W);
            const size_t r = pix / W;
            const int h = (int)(r % H), n = (int)(r / H);
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) a[c] += g_fetch(g, n, c, h, wv, C, H, W, down);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) red[c * 256 + threadIdx.x] = a[c];
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s)
#pragma unroll
                for (int c = 0; c < MAXC; ++c) red[c * 256 + threadIdx.x] += red[c * 256 + threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x < C) atomicAdd(db + threadIdx.x, red[threadIdx.x * 256] * mul);
    }
}

// ---- wide-channel variants (>= 256 feature channels: the 4x4 .. 32x32 stages, a few hundred pixels): one wave
// per pixel, each lane strides over the channels in float4 steps, xor-shuffle reduction.  The thread-per-pixel
// kernels above would leave all but 1-3 workgroups idle there and walk 2 KB rows serially.
__device__ __forceinline__ float wave_sum64(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void torgb_fwd_wide_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ prev, float* __restrict__ out,
    int N, int C, int H, int W, int Cin, float scale, float out_mul, float prev_mul)
{
    const int lane = threadIdx.x & 63;
    const size_t total = (size_t)N * H * W;
    const int c4n = Cin >> 2;
    for (size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < total; pix += (size_t)gridDim.x * 4) {
        const float4* xp = reinterpret_cast<const float4*>(x + pix * Cin);
        float a[MAXC] = {0.f, 0.f, 0.f, 0.f};
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const float4 xv = xp[c4];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C) {
                const float4 wv = *reinterpret_cast<const float4*>(w + (size_t)c * Cin + 4 * c4);
                a[c] += (xv.x * wv.x + xv.y * wv.y) + (xv.z * wv.z + xv.w * wv.w);
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) a[c] = wave_sum64(a[c]);
        if (lane == 0) {
            const int wv = (int)(pix % W);
            const size_t r = pix / W;
            const int h = (int)(r % H), n = (int)(r / H);
            for (int c = 0; c < C; ++c) {
                float v = (a[c] * scale + (bias ? bias[c] : 0.f)) * out_mul;
                if (prev) v += prev_mul * prev[(((size_t)n * C + c) * (H >> 1) + (h >> 1)) * (W >> 1) + (wv >> 1)];
                out[(((size_t)n * C + c) * H + h) * W + wv] = v;
            }
        }
    }
}

__global__ __launch_bounds__(256) void fromrgb_bwd_data_wide_kernel(
    const float* __restrict__ gz, const float* __restrict__ w, float* __restrict__ gimg,
    int N, int C, int H, int W, int Cout, int pool, int accumulate, float scale)
{
    const int lane = threadIdx.x & 63;
    const size_t total = (size_t)N * H * W;
    const int c4n = Cout >> 2;
    for (size_t pix = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < total; pix += (size_t)gridDim.x * 4) {
        const float4* gp = reinterpret_cast<const float4*>(gz + pix * Cout);
        float a[MAXC] = {0.f, 0.f, 0.f, 0.f};
        for (int c4 = lane; c4 < c4n; c4 += 64) {
            const float4 gv = gp[c4];
            const float* wr = w + (size_t)4 * c4 * C;             // w[co][c], 4 consecutive couts
#pragma unroll
            for (int c = 0; c < MAXC; ++c) if (c < C)
                a[c] += (gv.x * wr[c] + gv.y * wr[C + c]) + (gv.z * wr[2 * C + c] + gv.w * wr[3 * C + c]);
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) a[c] = wave_sum64(a[c]);
        if (lane == 0) {
            const int wv = (int)(pix % W);
            const size_t r = pix / W;
            const int h = (int)(r % H), n = (int)(r / H);
            for (int c = 0; c < C; ++c) {
                if (!pool) {
                    float* o = gimg + (((size_t)n * C + c) * H + h) * W + wv;
                    const float v = a[c] * scale;
                    *o = accumulate ? *o + v : v;
                } else {
                    const int H2 = 2 * H, W2 = 2 * W;
                    float* o = gimg + (((size_t)n * C + c) * H2 + 2 * h) * W2 + 2 * wv;
                    const float v = a[c] * scale * 0.25f;
                    if (accumulate) { o[0] += v; o[1] += v; o[W2] += v; o[W2 + 1] += v; }
                    else { o[0] = v; o[1] = v; o[W2] = v; o[W2 + 1] = v; }
                }
            }
        }
    }
}

// Workgroups of the narrow-layer weight gradients (thread -> pixel, (features + 1) x C partial sums per thread, one shuffle + LDS fold
// and features x C atomics per workgroup): the fold is a fixed cost per workgroup that grows with the feature count, so a thread
// should see 192 / features pixels -- swept per shape with tools/sweeps/bench_rgb_wgrad.py (n3 @512 16 features: 58 -> 35 us against the
// former flat cap of 1024 workgroups, n3 @256 32 features: 83 -> 44 us, n3 @1024 8 features: 42 -> 36 us, n9 unchanged).
inline int small_wgrad_grid(size_t total, int features)
{
    static const int forced = getenv("PG_RGB_WGRAD_GRID") ? atoi(getenv("PG_RGB_WGRAD_GRID")) : 0;       // sweeps
    if (forced > 0) return (int)((total + 255) / 256 < (size_t)forced ? (total + 255) / 256 : (size_t)forced);
    const size_t per_thread = (size_t)(192 / features < 1 ? 1 : 192 / features);
    size_t g = (total + 256 * per_thread - 1) / (256 * per_thread);
    if (g < 128) g = 128;
    if (g > 1024) g = 1024;
    if (g > (total + 255) / 256) g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : g);
}

inline int grid_for(size_t total, int block = 256, int cap = 256 * 16)
{
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int pg_fromrgb_fwd(const float* img, const float* w, const float* bias, const float* mask, float* y,
                              int N, int C, int H, int W, int Cout, int pool,
                              float scale, float slope, float mask_slope, pg_stream_t stream)
{
    if (!img || !w || !y || N <= 0 || H <= 0 || W <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cout & 3) return PG_E_ALIGN;
    const int flags = pool;                                 // PG_FLAG_*: bit 0 = 2x2 pooled image (the historic `pool`)
    pool = flags & 1;
    const int mbytes = (flags & PG_FLAG_MASK_BYTES) ? 1 : 0;
    unsigned char* ysigns = nullptr;
    if (flags & PG_FLAG_SIGNS_OUT) {
        if (!mask || mbytes) return PG_E_ARG;
        ysigns = reinterpret_cast<unsigned char*>(const_cast<float*>(mask));
        mask = nullptr;
    }
    const size_t npix = (size_t)N * H * W;
    if (npix >= 65536 && npix < (1ull << 31) && (Cout == 8 || (Cout == 16 && !mask))) {       // measured: tools/sweeps/bench_rgb_stream.py
        const int gr = grid_for(npix, 256, 256 * 16);
        hipStream_t s = (hipStream_t)stream;
        if (Cout == 8) hipLaunchKernelGGL(fromrgb_fwd_pix_kernel<8>, dim3(gr), dim3(256), 0, s, img, w, bias, mask, y, N, C, H, W, pool, scale, slope, mask_slope, mbytes, ysigns);
        else hipLaunchKernelGGL(fromrgb_fwd_pix_kernel<16>, dim3(gr), dim3(256), 0, s, img, w, bias, mask, y, N, C, H, W, pool, scale, slope, mask_slope, mbytes, ysigns);
        return (int)hipGetLastError();
    }
    const size_t total = npix * (Cout >> 2);
    hipLaunchKernelGGL(fromrgb_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       img, w, bias, mask, y, N, C, H, W, Cout, pool, scale, slope, mask_slope, mbytes, ysigns);
    return (int)hipGetLastError();
}

extern "C" int pg_fromrgb_bwd_data(const float* gz, const float* w, float* gimg,
                                   int N, int C, int H, int W, int Cout, int pool, int accumulate,
                                   float scale, pg_stream_t stream)
{
    if (!gz || !w || !gimg || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cout & 3) return PG_E_ALIGN;
    const size_t total = (size_t)N * H * W;
    if (Cout >= 256 && total <= 65536) {
        hipLaunchKernelGGL(fromrgb_bwd_data_wide_kernel, dim3(grid_for(total, 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                           gz, w, gimg, N, C, H, W, Cout, pool, accumulate, scale);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(fromrgb_bwd_data_kernel, dim3(grid_for(total)), dim3(256), (size_t)Cout * C * sizeof(float),
                       (hipStream_t)stream, gz, w, gimg, N, C, H, W, Cout, pool, accumulate, scale);
    return (int)hipGetLastError();
}

extern "C" int pg_fromrgb_wgrad(const float* gz, const float* img, float* dw, float* db,
                                int N, int C, int H, int W, int Cout, int pool, float scale, pg_stream_t stream)
{
    if (!gz || !img || !dw || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cout & 3) return PG_E_ALIGN;
    const size_t total = (size_t)N * H * W;
    if (total >= 65536 && (Cout == 8 || Cout == 16 || Cout == 32)) {
        const int g = small_wgrad_grid(total, Cout);
        hipStream_t s = (hipStream_t)stream;
        if (Cout == 8) hipLaunchKernelGGL(fromrgb_wgrad_small_kernel<8>, dim3(g), dim3(256), 0, s, gz, img, dw, db, N, C, H, W, pool, scale);
        else if (Cout == 16) hipLaunchKernelGGL(fromrgb_wgrad_small_kernel<16>, dim3(g), dim3(256), 0, s, gz, img, dw, db, N, C, H, W, pool, scale);
        else hipLaunchKernelGGL(fromrgb_wgrad_small_kernel<32>, dim3(g), dim3(256), 0, s, gz, img, dw, db, N, C, H, W, pool, scale);
        return (int)hipGetLastError();
    }
    // every workgroup ends with Cout*(C+1) atomics: >= 4 pixels per workgroup so the 4x4 stage still fills the chip,
    // ~128 workgroups beyond that so the atomics do not dominate (8x8 .. 32x32 stages)
    int blocks = grid_for(total, 4, total < 4096 ? 128 : (total < 12288 ? 256 : 512));      // measured (tools/sweeps/bench_rgb_wgrad.py)
    const int ppb = (int)((total + blocks - 1) / blocks);
    blocks = (int)((total + ppb - 1) / ppb);
    hipLaunchKernelGGL(fromrgb_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       gz, img, dw, db, N, C, H, W, Cout, pool, scale, ppb);
    return (int)hipGetLastError();
}

extern "C" int pg_torgb_fwd(const float* x, const float* w, const float* bias, const float* prev, float* out,
                            int N, int C, int H, int W, int Cin, float scale, float out_mul, float prev_mul,
                            pg_stream_t stream)
{
    if (!x || !w || !out || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cin & 3) return PG_E_ALIGN;
    const size_t total = (size_t)N * H * W;
    if (Cin >= 256 && total <= 65536) {
        hipLaunchKernelGGL(torgb_fwd_wide_kernel, dim3(grid_for(total, 4, 4096)), dim3(256), 0, (hipStream_t)stream,
                           x, w, bias, prev, out, N, C, H, W, Cin, scale, out_mul, prev_mul);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(torgb_fwd_kernel, dim3(grid_for(total, 256, 256 * 8)), dim3(256), (size_t)C * Cin * sizeof(float),
                       (hipStream_t)stream, x, w, bias, prev, out, N, C, H, W, Cin, scale, out_mul, prev_mul);
    return (int)hipGetLastError();
}

extern "C" int pg_torgb_bwd_data(const float* g, const float* w, float* gx,
                                 int N, int C, int H, int W, int Cin, int down, float mul_scale, pg_stream_t stream)
{
    if (!g || !w || !gx || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cin & 3) return PG_E_ALIGN;
    const size_t npix = (size_t)N * H * W;
    if (npix >= 65536 && npix < (1ull << 29) && (Cin == 8 || Cin == 16 || Cin == 32)) {      // measured: tools/sweeps/bench_rgb_stream.py
        hipStream_t s = (hipStream_t)stream;
        if (Cin == 8) hipLaunchKernelGGL((torgb_bwd_data_pix_kernel<8, false>), dim3(grid_for(npix, 256, 256 * 16)), dim3(256), 0, s, g, w, gx, N, C, H, W, down, mul_scale, (const float*)nullptr, (const float*)nullptr, 1.f);
        else if (Cin == 16) hipLaunchKernelGGL(torgb_bwd_data_narrow_kernel<16>, dim3(grid_for(npix * 4, 256, 256 * 32)), dim3(256), 0, s, g, w, gx, N, C, H, W, down, mul_scale);
        else hipLaunchKernelGGL(torgb_bwd_data_narrow_kernel<32>, dim3(grid_for(npix * 8, 256, 256 * 32)), dim3(256), 0, s, g, w, gx, N, C, H, W, down, mul_scale);
        return (int)hipGetLastError();
    }
    const size_t total = npix * (Cin >> 2);
    hipLaunchKernelGGL(torgb_bwd_data_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       g, w, gx, N, C, H, W, Cin, down, mul_scale);
    return (int)hipGetLastError();
}

// pg_torgb_bwd_data followed by the adjoint of the block's (LeakyReLU -> PixelNorm) in the same launch; 8 features on >= 256 x 256 maps
// (the 1024^2 stage of the default widths), PG_E_UNSUP otherwise (the caller runs pg_torgb_bwd_data + pg_pixelnorm_lrelu_bwd).
extern "C" int pg_torgb_bwd_data_pnbwd(const float* g, const float* w, const float* ysaved, const float* r, float* gx,
                                       int N, int C, int H, int W, int Cin, float mul_scale, float slope, pg_stream_t stream)
{
    if (!g || !w || !gx || !ysaved || !r || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    const size_t npix = (size_t)N * H * W;
    if (Cin != 8 || npix < 65536 || npix >= (1ull << 29)) return PG_E_UNSUP;
    hipLaunchKernelGGL((torgb_bwd_data_pix_kernel<8, true>), dim3(grid_for(npix, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream,
                       g, w, gx, N, C, H, W, 0, mul_scale, ysaved, r, slope);
    return (int)hipGetLastError();
}

extern "C" int pg_torgb_wgrad(const float* g, const float* x, float* dw, float* db,
                              int N, int C, int H, int W, int Cin, int down, float mul_scale, float mul,
                              pg_stream_t stream)
{
    if (!g || !x || !dw || N <= 0) return PG_E_ARG;
    if (C < 1 || C > MAXC) return PG_E_UNSUP;
    if (Cin & 3) return PG_E_ALIGN;
    const size_t total = (size_t)N * H * W;
    if (total >= 65536 && total < (1ull << 31) && (Cin == 8 || Cin == 16 || Cin == 32)) {
        const int gr = small_wgrad_grid(total, Cin);
        hipStream_t s = (hipStream_t)stream;
        if (Cin == 8) hipLaunchKernelGGL(torgb_wgrad_small_kernel<8>, dim3(gr), dim3(256), 0, s, g, x, dw, db, N, C, H, W, down, mul_scale, mul);
        else if (Cin == 16) hipLaunchKernelGGL(torgb_wgrad_small_kernel<16>, dim3(gr), dim3(256), 0, s, g, x, dw, db, N, C, H, W, down, mul_scale, mul);
        else hipLaunchKernelGGL(torgb_wgrad_small_kernel<32>, dim3(gr), dim3(256), 0, s, g, x, dw, db, N, C, H, W, down, mul_scale, mul);
        return (int)hipGetLastError();
    }
    int blocks = grid_for(total, 4, total < 8192 ? 256 : (total < 32768 ? 512 : 1024));     // Cin*C atomics per workgroup
    const int ppb = (int)((total + blocks - 1) / blocks);
    blocks = (int)((total + ppb - 1) / ppb);
    hipLaunchKernelGGL(torgb_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       g, x, dw, db, N, C, H, W, Cin, down, mul_scale, mul, ppb);
    return (int)hipGetLastError();
}
