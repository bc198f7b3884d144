// HBM-streaming and small-reduction kernels of the PGGAN hot path: pooling / upsample adjoint,
// fade-in blends, PixelNorm, minibatch-stddev (forward, adjoint, tangent, Hessian-vector term),
// the final Linear(nf0,1), WGAN-GP mixing / norms / seeds, loss algebra and Adam.
// All are float4-vectorised, grid-stride, with wavefront (64-lane) shuffle reductions.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pggan_hip.h"

namespace {

inline int grid_for(size_t total, int block = 256, int cap = 256 * 16)
{
    size_t g = (total + block - 1) / block;
    if (g > (size_t)cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// block-wide sum for blockDim.x == 1024 or 256 (multiple of 64); result valid in every thread.
__device__ __forceinline__ float block_sum(float v, float* sh)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

__device__ __forceinline__ float4 mask4(float4 v, float4 m, float slope)
{
    v.x *= m.x > 0.f ? 1.f : slope; v.y *= m.y > 0.f ? 1.f : slope;
    v.z *= m.z > 0.f ? 1.f : slope; v.w *= m.w > 0.f ? 1.f : slope;
    return v;
}

// ---------------------------------------------------------------------------------------- pooling
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float* __restrict__ x, const float* __restrict__ other,
                                                           float* __restrict__ y, int N, int H, int W, int C4, float a, float b)
{
    const size_t total = (size_t)N * H * W * C4;
    const size_t rs = (size_t)2 * W * C4;                 // input row stride in float4
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* o4 = reinterpret_cast<const float4*>(other);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        size_t r = idx / C4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const size_t n = r / H;
        const size_t base = ((n * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
        const float4 p0 = x4[base], p1 = x4[base + C4], p2 = x4[base + rs], p3 = x4[base + rs + C4];
        float4 v;
        v.x = ((p0.x + p1.x) + (p2.x + p3.x)) * 0.25f; v.y = ((p0.y + p1.y) + (p2.y + p3.y)) * 0.25f;
        v.z = ((p0.z + p1.z) + (p2.z + p3.z)) * 0.25f; v.w = ((p0.w + p1.w) + (p2.w + p3.w)) * 0.25f;
        if (other) {
            const float4 q = o4[idx];
            v.x = fmaf(v.x, a, b * q.x); v.y = fmaf(v.y, a, b * q.y); v.z = fmaf(v.z, a, b * q.z); v.w = fmaf(v.w, a, b * q.w);
        } else if (a != 1.f) { v.x *= a; v.y *= a; v.z *= a; v.w *= a; }
        y4[idx] = v;
    }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ mask,
                                                           float* __restrict__ gx, int N, int H, int W, int C4,
                                                           float mul, float mask_slope)
{
    // one thread per INPUT (fine) float4
    const int H2 = 2 * H, W2 = 2 * W;
    const size_t total = (size_t)N * H2 * W2 * C4;
    const float4* g4 = reinterpret_cast<const float4*>(gy);
    const float4* m4 = reinterpret_cast<const float4*>(mask);
    float4* o4 = reinterpret_cast<float4*>(gx);
    const float k = mul * 0.25f;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        size_t r = idx / C4;
        const int w = (int)(r % W2); r /= W2;
        const int h = (int)(r % H2); const size_t n = r / H2;
        float4 v = g4[((n * H + (h >> 1)) * W + (w >> 1)) * C4 + c];
        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
        if (mask) v = mask4(v, m4[idx], mask_slope);
        o4[idx] = v;
    }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const float* __restrict__ g, float* __restrict__ gx,
                                                            int N, int H, int W, int C4)
{
    const size_t total = (size_t)N * H * W * C4;
    const size_t rs = (size_t)2 * W * C4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* o4 = reinterpret_cast<float4*>(gx);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4);
        size_t r = idx / C4;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const size_t n = r / H;
        const size_t base = ((n * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
        const float4 p0 = g4[base], p1 = g4[base + C4], p2 = g4[base + rs], p3 = g4[base + rs + C4];
        float4 v;
        v.x = (p0.x + p1.x) + (p2.x + p3.x); v.y = (p0.y + p1.y) + (p2.y + p3.y);
        v.z = (p0.z + p1.z) + (p2.z + p3.z); v.w = (p0.w + p1.w) + (p2.w + p3.w);
        o4[idx] = v;
    }
}

__global__ __launch_bounds__(256) void axpby_mask_kernel(const float* __restrict__ x, const float* __restrict__ other,
                                                         const float* __restrict__ mask, float* __restrict__ y,
                                                         size_t n4, float a, float b, float mask_slope)
{
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* o4 = reinterpret_cast<const float4*>(other);
    const float4* m4 = reinterpret_cast<const float4*>(mask);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = x4[i];
        v.x *= a; v.y *= a; v.z *= a; v.w *= a;
        if (other) { const float4 q = o4[i]; v.x += b * q.x; v.y += b * q.y; v.z += b * q.z; v.w += b * q.w; }
        if (mask) v = mask4(v, m4[i], mask_slope);
        y4[i] = v;
    }
}

// -------------------------------------------------------------------------------------- pixelnorm
// LPP lanes cooperate on one pixel (LPP = power of two <= 64, each lane strides over C in float4).
template <int LPP>
__device__ __forceinline__ float group_sum(float v)
{
#pragma unroll
    for (int o = LPP >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <int LPP>
__global__ __launch_bounds__(256) void pixelnorm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                            float* __restrict__ r, size_t P, int C, float eps)
{
    // all LPP lanes of a group share `pix`, so a group enters/leaves the loop together and the
    // xor-shuffles below only ever read lanes of the own (active) group.
    const int C4 = C >> 2;
    const int sub = threadIdx.x % LPP;
    const size_t pstride = (size_t)gridDim.x * (256 / LPP);
    for (size_t pix = (size_t)blockIdx.x * (256 / LPP) + threadIdx.x / LPP; pix < P; pix += pstride) {
        const float4* xp = reinterpret_cast<const float4*>(x + pix * C);
        float s = 0.f;
        for (int c = sub; c < C4; c += LPP) { const float4 v = xp[c]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
        s = group_sum<LPP>(s);
        const float rr = rsqrtf(s / (float)C + eps);
        float4* yp = reinterpret_cast<float4*>(y + pix * C);
        for (int c = sub; c < C4; c += LPP) { float4 v = xp[c]; v.x *= rr; v.y *= rr; v.z *= rr; v.w *= rr; yp[c] = v; }
        if (sub == 0 && r) r[pix] = rr;
    }
}

template <int LPP>
__global__ __launch_bounds__(256) void pixelnorm_lrelu_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                  const float* __restrict__ r, const float* __restrict__ inj,
                                                                  float* __restrict__ gz, size_t P, int C, float slope)
{
    const int C4 = C >> 2;
    const int sub = threadIdx.x % LPP;
    const size_t pstride = (size_t)gridDim.x * (256 / LPP);
    for (size_t pix = (size_t)blockIdx.x * (256 / LPP) + threadIdx.x / LPP; pix < P; pix += pstride) {
        const float4* gp = reinterpret_cast<const float4*>(gy + pix * C);
        const float4* yp = reinterpret_cast<const float4*>(y + pix * C);
        float s = 0.f;
        if (r) for (int c = sub; c < C4; c += LPP) {
            const float4 g = gp[c], v = yp[c];
            s += g.x * v.x + g.y * v.y + g.z * v.z + g.w * v.w;
        }
        s = group_sum<LPP>(s);
        const float rr = r ? r[pix] : 1.f;
        const float mean = r ? s / (float)C : 0.f;
        float4* op = reinterpret_cast<float4*>(gz + pix * C);
        const float4* ip = inj ? reinterpret_cast<const float4*>(inj + pix * C) : nullptr;
        for (int c = sub; c < C4; c += LPP) {
            const float4 g = gp[c], v = yp[c];
            float4 o;
            o.x = rr * (g.x - v.x * mean); o.y = rr * (g.y - v.y * mean);
            o.z = rr * (g.z - v.z * mean); o.w = rr * (g.w - v.w * mean);
            if (ip) { const float4 j = ip[c]; o.x += j.x; o.y += j.y; o.z += j.z; o.w += j.w; }
            o.x *= v.x > 0.f ? 1.f : slope; o.y *= v.y > 0.f ? 1.f : slope;
            o.z *= v.z > 0.f ? 1.f : slope; o.w *= v.w > 0.f ? 1.f : slope;
            op[c] = o;
        }
    }
}

// Second-order companion of PixelNorm for the gradient penalty (Discriminator(pixelnorm=True)):
// with P(h) = r (I - y y^T / C) the Jacobian (== adjoint) of y = h * r(h):
//   ty  = P t                                                   tangent through the layer
//   inj = grad_h < t, P(h) a >  =  -(r^2 S / C) y - (r / C) P[(a.y) t + (t.y) a],   S = t.a - (t.y)(a.y)/C
// t: tangent at the PixelNorm input, a: first-backward adjoint at the PixelNorm output, y/r: saved forward.
template <int LPP>
__global__ __launch_bounds__(256) void pixelnorm_tangent_kernel(const float* __restrict__ t, const float* __restrict__ y,
                                                                const float* __restrict__ r, const float* __restrict__ a,
                                                                float* __restrict__ ty, float* __restrict__ inj,
                                                                size_t P, int C)
{
    const int C4 = C >> 2;
    const int sub = threadIdx.x % LPP;
    const size_t pstride = (size_t)gridDim.x * (256 / LPP);
    const float invC = 1.f / (float)C;
    for (size_t pix = (size_t)blockIdx.x * (256 / LPP) + threadIdx.x / LPP; pix < P; pix += pstride) {
        const float4* tp = reinterpret_cast<const float4*>(t + pix * C);
        const float4* yp = reinterpret_cast<const float4*>(y + pix * C);
        const float4* ap = reinterpret_cast<const float4*>(a + pix * C);
        float ty_ = 0.f, ay_ = 0.f, ta_ = 0.f;
        for (int c = sub; c < C4; c += LPP) {
            const float4 tv = tp[c], yv = yp[c], av = ap[c];
            ty_ += tv.x * yv.x + tv.y * yv.y + tv.z * yv.z + tv.w * yv.w;
            ay_ += av.x * yv.x + av.y * yv.y + av.z * yv.z + av.w * yv.w;
            ta_ += tv.x * av.x + tv.y * av.y + tv.z * av.z + tv.w * av.w;
        }
        ty_ = group_sum<LPP>(ty_); ay_ = group_sum<LPP>(ay_); ta_ = group_sum<LPP>(ta_);
        const float rr = r[pix];
        const float S = ta_ - ty_ * ay_ * invC;
        const float k_y = -(rr * rr * S * invC);            // coefficient of y from the d r term
        const float yv_dot = 2.f * ay_ * ty_;                // y . [(a.y) t + (t.y) a]
        const float kp = -(rr * invC) * rr;                  // -(r/C) * r  (outer r of P)
        float4* typ = reinterpret_cast<float4*>(ty + pix * C);
        float4* ip = reinterpret_cast<float4*>(inj + pix * C);
        for (int c = sub; c < C4; c += LPP) {
            const float4 tv = tp[c], yv = yp[c], av = ap[c];
            float4 o, j;
            o.x = rr * (tv.x - yv.x * ty_ * invC); o.y = rr * (tv.y - yv.y * ty_ * invC);
            o.z = rr * (tv.z - yv.z * ty_ * invC); o.w = rr * (tv.w - yv.w * ty_ * invC);
            j.x = k_y * yv.x + kp * (ay_ * tv.x + ty_ * av.x - yv.x * yv_dot * invC);
            j.y = k_y * yv.y + kp * (ay_ * tv.y + ty_ * av.y - yv.y * yv_dot * invC);
            j.z = k_y * yv.z + kp * (ay_ * tv.z + ty_ * av.z - yv.z * yv_dot * invC);
            j.w = k_y * yv.w + kp * (ay_ * tv.w + ty_ * av.w - yv.w * yv_dot * invC);
            typ[c] = o; ip[c] = j;
        }
    }
}

// Narrow layers (C = 8 / 16: the 512^2 and 1024^2 stages, where this kernel moves the most bytes): one thread per
// pixel keeps the whole channel vector of gy and y in registers -- one pass, no shuffles, no re-read.
template <int C4>
__global__ __launch_bounds__(256) void pixelnorm_lrelu_bwd_narrow_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                         const float* __restrict__ r, const float* __restrict__ inj,
                                                                         float* __restrict__ gz, size_t P, float slope)
{
    constexpr int C = 4 * C4;
    for (size_t pix = (size_t)blockIdx.x * 256 + threadIdx.x; pix < P; pix += (size_t)gridDim.x * 256) {
        float4 g[C4], v[C4];
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            g[c] = reinterpret_cast<const float4*>(gy + pix * C)[c];
            v[c] = reinterpret_cast<const float4*>(y + pix * C)[c];
        }
        float rr = 1.f, mean = 0.f;
        if (r) {
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < C4; ++c) s += g[c].x * v[c].x + g[c].y * v[c].y + g[c].z * v[c].z + g[c].w * v[c].w;
            rr = r[pix]; mean = s / (float)C;
        }
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            float4 o;
            o.x = rr * (g[c].x - v[c].x * mean); o.y = rr * (g[c].y - v[c].y * mean);
            o.z = rr * (g[c].z - v[c].z * mean); o.w = rr * (g[c].w - v[c].w * mean);
            if (inj) { const float4 q = reinterpret_cast<const float4*>(inj + pix * C)[c]; o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w; }
            o.x *= v[c].x > 0.f ? 1.f : slope; o.y *= v[c].y > 0.f ? 1.f : slope;
            o.z *= v[c].z > 0.f ? 1.f : slope; o.w *= v[c].w > 0.f ? 1.f : slope;
            reinterpret_cast<float4*>(gz + pix * C)[c] = o;
        }
    }
}

inline int pick_lpp(int C) { int c4 = C >> 2; int l = 1; while (l < c4 && l < 64) l <<= 1; return l; }

// ------------------------------------------------------------------------------ minibatch stddev
// The tensor of one group is n*HW*C <= a few 100 K floats (L2 resident).  MB_PARTS workgroups per group:
// a statistics launch leaves per-slice partials in the stats row, the writer launch merges them (every
// workgroup redundantly, in the same order -> identical mu/sigma everywhere) and streams its slice.
// stats row (PG_MBSTD_STATS_STRIDE floats): [mu, sigma, partials...];  tstats row: [mean(tx), <x-mu,tx>, partials...]
constexpr int MB_PARTS = 32;
constexpr int MB_STRIDE = PG_MBSTD_STATS_STRIDE;
static_assert(MB_STRIDE >= 2 + 4 * MB_PARTS, "stats row too small for the partials");

__device__ __forceinline__ void mb_slice(size_t M4, size_t& b, size_t& e)
{
    const size_t per = (M4 + MB_PARTS - 1) / MB_PARTS;
    b = (size_t)blockIdx.x * per; e = b + per < M4 ? b + per : M4;
    if (b > M4) b = M4;
}

// partial {count, mean, M2} of this workgroup's slice (two passes over the slice: exact local mean first)
__global__ __launch_bounds__(256) void mbstd_stats_kernel(const float* __restrict__ x, float* __restrict__ stats,
                                                          int n, int HW, int C)
{
    __shared__ float sh[16];
    const int g = blockIdx.y;
    const size_t M = (size_t)n * HW * C, M4 = M >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)g * M);
    size_t b, e; mb_slice(M4, b, e);
    float s = 0.f;
#pragma unroll 4
    for (size_t i = b + threadIdx.x; i < e; i += 256) { const float4 v = x4[i]; s += (v.x + v.y) + (v.z + v.w); }
    const float cnt = (float)(4 * (e - b));
    const float mean = cnt > 0.f ? block_sum(s, sh) / cnt : 0.f;
    float q = 0.f;
#pragma unroll 4
    for (size_t i = b + threadIdx.x; i < e; i += 256) {
        const float4 v = x4[i];
        const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float m2 = block_sum(q, sh);
    if (threadIdx.x == 0) {
        float* pr = stats + (size_t)g * MB_STRIDE + 2 + 4 * blockIdx.x;
        pr[0] = cnt; pr[1] = mean; pr[2] = m2;
    }
}

// Chan's pairwise merge of the partials, sequential and identical in every workgroup.  Exact-global mode under data parallelism
// (pg_mbstd_write with ``gathered``): the rows of every rank's shard of the group ([rank][group][MB_STRIDE], rank-major,
// ``rstride`` floats between ranks) are merged in rank order -- the same order on every rank, so mu / sigma are bit-identical everywhere.
__device__ __forceinline__ void mb_merge(const float* __restrict__ row, int nranks, size_t rstride, float M, float& mu, float& sigma)
{
    float cnt = 0.f, mean = 0.f, m2 = 0.f;
    for (int r = 0; r < nranks; ++r) {
        const float* pr = row + (size_t)r * rstride;
        for (int i = 0; i < MB_PARTS; ++i) {
            const float c = pr[2 + 4 * i], m = pr[3 + 4 * i], q = pr[4 + 4 * i];
            if (c > 0.f) {
                const float tot = cnt + c, d = m - mean;
                mean += d * (c / tot);
                m2 += q + d * d * (cnt * c / tot);
                cnt = tot;
            }
        }
    }
    mu = mean;
    sigma = sqrtf(m2 / M + 1.0e-8f);
}

// ``gathered`` (exact-global mode): the partial rows of all ``nranks`` shards, [rank][G][MB_STRIDE]; null: this rank's row alone.  M below is
// the element count of the WHOLE group (all shards: equal shard sizes, checked by the host side).
template <bool TANGENT>
__global__ __launch_bounds__(256) void mbstd_write_kernel(const float* __restrict__ src, float* __restrict__ y,
                                                          float* __restrict__ stats, const float* __restrict__ xstats,
                                                          const float* __restrict__ gathered, int nranks,
                                                          int n, int HW, int C, int CP)
{
    const int g = blockIdx.y;
    const size_t rows = (size_t)n * HW;
    const size_t M = rows * C;
    const float Mg = (float)M * (float)nranks;
    float* row = stats + (size_t)g * MB_STRIDE;
    const float* src_rows = gathered ? gathered + (size_t)g * MB_STRIDE : row;
    const size_t rstride = (size_t)gridDim.y * MB_STRIDE;
    float extra;                                             // value of channel C
    if (!TANGENT) {
        float mu, sigma; mb_merge(src_rows, gathered ? nranks : 1, rstride, Mg, mu, sigma);
        if (blockIdx.x == 0 && threadIdx.x == 0) { row[0] = mu; row[1] = sigma; }
        extra = sigma;
    } else {
        float ts = 0.f, dot = 0.f;
        for (int r = 0; r < (gathered ? nranks : 1); ++r)
            for (int i = 0; i < MB_PARTS; ++i) { ts += src_rows[r * rstride + 2 + 4 * i]; dot += src_rows[r * rstride + 3 + 4 * i]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) { row[0] = ts / Mg; row[1] = dot; }
        extra = dot / (Mg * xstats[(size_t)g * MB_STRIDE + 1]);
    }
    const int C4 = C >> 2, CP4 = CP >> 2;
    const float4* s4 = reinterpret_cast<const float4*>(src + (size_t)g * M);
    float4* y4 = reinterpret_cast<float4*>(y + (size_t)g * rows * CP);
    const size_t total = rows * CP4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / CP4; const int c = (int)(i % CP4);
        float4 v;
        if (c < C4) v = s4[r * C4 + c];
        else if (c == C4) v = make_float4(extra, 0.f, 0.f, 0.f);
        else v = make_float4(0.f, 0.f, 0.f, 0.f);
        y4[i] = v;
    }
}

// partial {sum tx, <x-mu, tx>} of this workgroup's slice
__global__ __launch_bounds__(256) void mbstd_tangent_stats_kernel(const float* __restrict__ x, const float* __restrict__ tx,
                                                                  const float* __restrict__ stats, float* __restrict__ tstats,
                                                                  int n, int HW, int C)
{
    __shared__ float sh[16];
    const int g = blockIdx.y;
    const size_t M = (size_t)n * HW * C, M4 = M >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)g * M);
    const float4* t4 = reinterpret_cast<const float4*>(tx + (size_t)g * M);
    const float mu = stats[(size_t)g * MB_STRIDE];
    size_t b, e; mb_slice(M4, b, e);
    float s = 0.f, d = 0.f;
#pragma unroll 4
    for (size_t i = b + threadIdx.x; i < e; i += 256) {
        const float4 v = x4[i], t = t4[i];
        s += (t.x + t.y) + (t.z + t.w);
        d += (v.x - mu) * t.x + (v.y - mu) * t.y + (v.z - mu) * t.z + (v.w - mu) * t.w;
    }
    s = block_sum(s, sh);
    d = block_sum(d, sh);
    if (threadIdx.x == 0) {
        float* pr = tstats + (size_t)g * MB_STRIDE + 2 + 4 * blockIdx.x;
        pr[0] = s; pr[1] = d;
    }
}

__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                        const float* __restrict__ stats, const float* __restrict__ tx,
                                                        const float* __restrict__ tstats, const float* __restrict__ gy_first,
                                                        float* __restrict__ gx, int n, int HW, int C, int CP,
                                                        int apply_mask, float mask_slope,
                                                        const float* __restrict__ gsums, int nranks)
{
    __shared__ float sh[16];
    const int g = blockIdx.y;
    const size_t rows = (size_t)n * HW;
    const size_t M = rows * C;
    const float mu = stats[(size_t)g * MB_STRIDE], sigma = stats[(size_t)g * MB_STRIDE + 1];
    // Gs over the (few hundred) rows: recomputed by every workgroup, same order -> same value.  Exact-global mode: ``gsums`` =
    // {Gs, Gs1} of the group summed over all shards (pg_mbstd_gsum + a sum all-reduce), M counts all shards.
    float gs = 0.f, gs1 = 0.f;
    if (!gsums) {
        if (gy) for (size_t r = threadIdx.x; r < rows; r += 256) gs += gy[((size_t)g * rows + r) * CP + C];
        if (tx) for (size_t r = threadIdx.x; r < rows; r += 256) gs1 += gy_first[((size_t)g * rows + r) * CP + C];
    }
    const float Gs = gsums ? (gy ? gsums[2 * g] : 0.f) : (gy ? block_sum(gs, sh) : 0.f);
    const float Gs1 = gsums ? (tx ? gsums[2 * g + 1] : 0.f) : (tx ? block_sum(gs1, sh) : 0.f);
    const float invMs = 1.f / ((float)M * (float)nranks * sigma);
    const float k1 = Gs * invMs;
    float tmean = 0.f, k2 = 0.f, k3 = 0.f;
    if (tx) {
        tmean = tstats[(size_t)g * MB_STRIDE];
        const float dot = tstats[(size_t)g * MB_STRIDE + 1];
        k2 = Gs1 * invMs;                                   // multiplies (tx - mean tx)
        k3 = k2 * dot / ((float)M * (float)nranks * sigma * sigma);         // multiplies (x - mu)
    }
    const int C4 = C >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x + (size_t)g * M);
    const float4* t4 = tx ? reinterpret_cast<const float4*>(tx + (size_t)g * M) : nullptr;
    float4* o4 = reinterpret_cast<float4*>(gx + (size_t)g * M);
    const float kx = k1 - k3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < rows * C4; i += (size_t)gridDim.x * 256) {
        const size_t row = i / C4; const int c = (int)(i % C4);
        const float4 xv = x4[i];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy) v = *reinterpret_cast<const float4*>(gy + ((size_t)g * rows + row) * CP + 4 * c);
        v.x += kx * (xv.x - mu); v.y += kx * (xv.y - mu); v.z += kx * (xv.z - mu); v.w += kx * (xv.w - mu);
        if (tx) {
            const float4 t = t4[i];
            v.x += k2 * (t.x - tmean); v.y += k2 * (t.y - tmean); v.z += k2 * (t.z - tmean); v.w += k2 * (t.w - tmean);
        }
        if (apply_mask) v = mask4(v, xv, mask_slope);
        o4[i] = v;
    }
}

// local {sum over rows of gy[.., C], of gy_first[.., C]} per group (exact-global mode: all-reduced, then handed to mbstd_bwd_kernel)
__global__ __launch_bounds__(256) void mbstd_gsum_kernel(const float* __restrict__ gy, const float* __restrict__ gy_first,
                                                         float* __restrict__ out, int n, int HW, int C, int CP)
{
    __shared__ float sh[16];
    const int g = blockIdx.x;
    const size_t rows = (size_t)n * HW;
    float a = 0.f, b = 0.f;
    if (gy) for (size_t r = threadIdx.x; r < rows; r += 256) a += gy[((size_t)g * rows + r) * CP + C];
    if (gy_first) for (size_t r = threadIdx.x; r < rows; r += 256) b += gy_first[((size_t)g * rows + r) * CP + C];
    a = block_sum(a, sh);
    b = block_sum(b, sh);
    if (threadIdx.x == 0) { out[2 * g] = a; out[2 * g + 1] = b; }
}

// ----------------------------------------------------------------------------------- Linear(C,1)
__global__ __launch_bounds__(64) void linear1_fwd_kernel(const float* __restrict__ h, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ s, int C)
{
    const int n = blockIdx.x;
    float a = 0.f;
    for (int c = threadIdx.x; c < C; c += 64) a = fmaf(h[(size_t)n * C + c], w[c], a);
    a = wave_sum(a);
    if (threadIdx.x == 0) s[n] = a + (b ? b[0] : 0.f);
}

__global__ __launch_bounds__(256) void linear1_bwd_data_kernel(const float* __restrict__ gs, const float* __restrict__ w,
                                                               const float* __restrict__ mask, float* __restrict__ gh,
                                                               int N, int C, float mask_slope)
{
    const size_t total = (size_t)N * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const size_t n = i / C;
        float v = gs[n] * w[c];
        if (mask) v *= mask[i] > 0.f ? 1.f : mask_slope;
        gh[i] = v;
    }
}

__global__ __launch_bounds__(256) void linear1_wgrad_kernel(const float* __restrict__ gs, const float* __restrict__ h,
                                                            float* __restrict__ dw, float* __restrict__ db, int N, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        float a = 0.f;
        for (int n = 0; n < N; ++n) a = fmaf(gs[n], h[(size_t)n * C + c], a);
        dw[c] += a;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && db) {
        float a = 0.f;
        for (int n = 0; n < N; ++n) a += gs[n];
        db[0] += a;
    }
}

// ------------------------------------------------------------------------------------- WGAN-GP
__global__ __launch_bounds__(256) void gp_mix_kernel(const float* __restrict__ real, const float* __restrict__ fake,
                                                     const float* __restrict__ m, float* __restrict__ mixed, size_t E4)
{
    const int n = blockIdx.y;
    const float mf = m[n], mr = 1.f - mf;
    const float4* r4 = reinterpret_cast<const float4*>(real) + (size_t)n * E4;
    const float4* f4 = reinterpret_cast<const float4*>(fake) + (size_t)n * E4;
    float4* o4 = reinterpret_cast<float4*>(mixed) + (size_t)n * E4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < E4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = r4[i], b = f4[i];
        o4[i] = make_float4(a.x * mr + b.x * mf, a.y * mr + b.y * mf, a.z * mr + b.z * mf, a.w * mr + b.w * mf);
    }
}

__global__ __launch_bounds__(256) void row_sumsq_kernel(const float* __restrict__ g, float* __restrict__ ss, size_t E4)
{
    __shared__ float sh[16];
    const int n = blockIdx.y;
    const float4* g4 = reinterpret_cast<const float4*>(g) + (size_t)n * E4;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < E4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i]; s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = block_sum(s, sh);
    if (threadIdx.x == 0) atomicAdd(ss + n, s);
}

__global__ __launch_bounds__(256) void gp_seed_kernel(const float* __restrict__ g, const float* __restrict__ ss,
                                                      float* __restrict__ gp, float* __restrict__ u, size_t E4,
                                                      float lambda, float target, float inv_n)
{
    const int n = blockIdx.y;
    const float norm = sqrtf(ss[n]);
    const float d = norm - target;
    const float coef = norm > 0.f ? inv_n * 2.f * lambda * d / (target * target * norm) : 0.f;
    if (blockIdx.x == 0 && threadIdx.x == 0) gp[n] = d * d * lambda / (target * target);
    const float4* g4 = reinterpret_cast<const float4*>(g) + (size_t)n * E4;
    float4* u4 = reinterpret_cast<float4*>(u) + (size_t)n * E4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < E4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = g4[i];
        u4[i] = make_float4(v.x * coef, v.y * coef, v.z * coef, v.w * coef);
    }
}

__global__ __launch_bounds__(64) void d_loss_kernel(const float* __restrict__ s, const float* __restrict__ gp,
                                                    float* __restrict__ d_cost, float* __restrict__ d_real_loss,
                                                    float* __restrict__ d_fake_loss, float* __restrict__ gscore,
                                                    int N, float eps)
{
    float acc = 0.f;
    const float invn = 1.f / (float)N;
    for (int n = threadIdx.x; n < N; n += 64) {
        const float sr = s[n], sf = s[N + n];
        const float rl = -sr + sr * sr * eps;
        d_real_loss[n] = rl; d_fake_loss[n] = sf;
        acc += sf + rl + gp[n];
        gscore[n] = (-1.f + 2.f * eps * sr) * invn;
        gscore[N + n] = invn;
        gscore[2 * N + n] = 0.f;
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) d_cost[0] = acc * invn;
}

__global__ __launch_bounds__(64) void g_loss_kernel(const float* __restrict__ s, float* __restrict__ g_cost,
                                                    float* __restrict__ gscore, int N)
{
    float acc = 0.f;
    const float invn = 1.f / (float)N;
    for (int n = threadIdx.x; n < N; n += 64) { acc -= s[n]; gscore[n] = -invn; }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) g_cost[0] = acc * invn;
}

// ---------------------------------------------------------------------------------------- Adam
// B1ZERO: beta1 == 0 (the reference's Adam(0, 0.99), train.py:195): exp_avg = grad, so the old first moment is never read -- one of the
// seven streams of this bandwidth-bound kernel (it is still written: checkpoints carry it, plugins.SaverPlugin).
template <bool B1ZERO>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float step_size, float beta1, float beta2, float eps,
                                                   float inv_bc2_sqrt, float grad_scale)
{
    const size_t n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
#define ADAM1(P, G, Mm, V) { const float g_ = (G) * grad_scale; Mm = Mm * beta1 + omb1 * g_; V = V * beta2 + omb2 * g_ * g_; \
                             P -= step_size * Mm / (sqrtf(V) * inv_bc2_sqrt + eps); }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 pp = p4[i], mm = B1ZERO ? make_float4(0.f, 0.f, 0.f, 0.f) : m4[i], vv = v4[i]; const float4 gg = g4[i];
        ADAM1(pp.x, gg.x, mm.x, vv.x) ADAM1(pp.y, gg.y, mm.y, vv.y) ADAM1(pp.z, gg.z, mm.z, vv.z) ADAM1(pp.w, gg.w, mm.w, vv.w)
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n not a multiple of 4)
    const size_t tail0 = n4 << 2;
    if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
        const size_t i = tail0 + threadIdx.x;
        float pp = p[i], mm = B1ZERO ? 0.f : m[i], vv = v[i];
        ADAM1(pp, g[i], mm, vv)
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
#undef ADAM1
}

}  // namespace

#define LAUNCH(kern, grid, block, smem, stream, ...) \
    hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__); \
    return (int)hipGetLastError()

namespace {
__global__ __launch_bounds__(256) void signbytes_to_mask_kernel(const unsigned char* __restrict__ b, float* __restrict__ m, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned char v = b[i];
        reinterpret_cast<float4*>(m)[i] = make_float4((v & 1) ? 1.f : -1.f, (v & 2) ? 1.f : -1.f, (v & 4) ? 1.f : -1.f, (v & 8) ? 1.f : -1.f);
    }
}
}  // namespace

extern "C" int pg_signbytes_to_mask(const unsigned char* bytes, float* mask, int64_t nbytes, pg_stream_t stream)
{
    if (!bytes || !mask || nbytes <= 0) return PG_E_ARG;
    size_t g = ((size_t)nbytes + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(signbytes_to_mask_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, bytes, mask, (size_t)nbytes);
    return (int)hipGetLastError();
}

extern "C" int pg_abi_version(void) { return 25; }

extern "C" int pg_avgpool2_fwd(const float* x, const float* other, float* y, int N, int H, int W, int C,
                               float a, float b, pg_stream_t stream)
{
    if (!x || !y || N <= 0 || H <= 0 || W <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    LAUNCH(avgpool2_fwd_kernel, dim3(grid_for((size_t)N * H * W * (C >> 2))), dim3(256), 0, stream, x, other, y, N, H, W, C >> 2, a, b);
}

extern "C" int pg_avgpool2_bwd(const float* gy, const float* mask, float* gx, int N, int H, int W, int C,
                               float mul, float mask_slope, pg_stream_t stream)
{
    if (!gy || !gx || N <= 0 || H <= 0 || W <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    LAUNCH(avgpool2_bwd_kernel, dim3(grid_for((size_t)N * 4 * H * W * (C >> 2))), dim3(256), 0, stream, gy, mask, gx, N, H, W, C >> 2, mul, mask_slope);
}

extern "C" int pg_upsample2_bwd(const float* g, float* gx, int N, int H, int W, int C, pg_stream_t stream)
{
    if (!g || !gx || N <= 0 || H <= 0 || W <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    LAUNCH(upsample2_bwd_kernel, dim3(grid_for((size_t)N * H * W * (C >> 2))), dim3(256), 0, stream, g, gx, N, H, W, C >> 2);
}

extern "C" int pg_axpby_mask(const float* x, const float* other, const float* mask, float* y, int64_t n,
                             float a, float b, float mask_slope, pg_stream_t stream)
{
    if (!x || !y || n <= 0) return PG_E_ARG;
    if (n & 3) return PG_E_ALIGN;
    LAUNCH(axpby_mask_kernel, dim3(grid_for((size_t)n >> 2)), dim3(256), 0, stream, x, other, mask, y, (size_t)n >> 2, a, b, mask_slope);
}

extern "C" int pg_pixelnorm_fwd(const float* x, float* y, float* r, int64_t P, int C, float eps, pg_stream_t stream)
{
    if (!x || !y || P <= 0 || C <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    const int lpp = pick_lpp(C);
    const int grid = grid_for((size_t)P * lpp);
    hipStream_t s = (hipStream_t)stream;
    switch (lpp) {
#define CASE(L) case L: hipLaunchKernelGGL(pixelnorm_fwd_kernel<L>, dim3(grid), dim3(256), 0, s, x, y, r, (size_t)P, C, eps); break;
        CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
    return (int)hipGetLastError();
}

extern "C" int pg_pixelnorm_lrelu_bwd(const float* gy, const float* y, const float* r, float* gz,
                                      int64_t P, int C, float slope, pg_stream_t stream)
{
    if (!gy || !y || !gz || P <= 0 || C <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (C == 8 || C == 16) {
        const int g = grid_for((size_t)P, 256, 1 << 20);
        if (C == 8) hipLaunchKernelGGL(pixelnorm_lrelu_bwd_narrow_kernel<2>, dim3(g), dim3(256), 0, s, gy, y, r, (const float*)nullptr, gz, (size_t)P, slope);
        else hipLaunchKernelGGL(pixelnorm_lrelu_bwd_narrow_kernel<4>, dim3(g), dim3(256), 0, s, gy, y, r, (const float*)nullptr, gz, (size_t)P, slope);
        return (int)hipGetLastError();
    }
    const int lpp = pick_lpp(C);
    const int grid = grid_for((size_t)P * lpp);
    switch (lpp) {
#define CASE(L) case L: hipLaunchKernelGGL(pixelnorm_lrelu_bwd_kernel<L>, dim3(grid), dim3(256), 0, s, gy, y, r, (const float*)nullptr, gz, (size_t)P, C, slope); break;
        CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
    return (int)hipGetLastError();
}

extern "C" int pg_pixelnorm_lrelu_bwd_inj(const float* gy, const float* y, const float* r, const float* inj, float* gz,
                                          int64_t P, int C, float slope, pg_stream_t stream)
{
    if (!gy || !y || !r || !inj || !gz || P <= 0 || C <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (C == 8 || C == 16) {
        const int g = grid_for((size_t)P, 256, 1 << 20);
        if (C == 8) hipLaunchKernelGGL(pixelnorm_lrelu_bwd_narrow_kernel<2>, dim3(g), dim3(256), 0, s, gy, y, r, inj, gz, (size_t)P, slope);
        else hipLaunchKernelGGL(pixelnorm_lrelu_bwd_narrow_kernel<4>, dim3(g), dim3(256), 0, s, gy, y, r, inj, gz, (size_t)P, slope);
        return (int)hipGetLastError();
    }
    const int lpp = pick_lpp(C);
    const int grid = grid_for((size_t)P * lpp);
    switch (lpp) {
#define CASE(L) case L: hipLaunchKernelGGL(pixelnorm_lrelu_bwd_kernel<L>, dim3(grid), dim3(256), 0, s, gy, y, r, inj, gz, (size_t)P, C, slope); break;
        CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
    return (int)hipGetLastError();
}

extern "C" int pg_pixelnorm_tangent(const float* t, const float* y, const float* r, const float* a, float* ty, float* inj,
                                    int64_t P, int C, pg_stream_t stream)
{
    if (!t || !y || !r || !a || !ty || !inj || P <= 0 || C <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    const int lpp = pick_lpp(C);
    const int grid = grid_for((size_t)P * lpp);
    hipStream_t s = (hipStream_t)stream;
    switch (lpp) {
#define CASE(L) case L: hipLaunchKernelGGL(pixelnorm_tangent_kernel<L>, dim3(grid), dim3(256), 0, s, t, y, r, a, ty, inj, (size_t)P, C); break;
        CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64)
#undef CASE
    }
    return (int)hipGetLastError();
}

extern "C" int pg_mbstd_fwd(const float* x, float* y, float* stats, int G, int n, int HW, int C, int CP, pg_stream_t stream)
{
    if (!x || !y || !stats || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mbstd_stats_kernel, dim3(MB_PARTS, G), dim3(256), 0, s, x, stats, n, HW, C);
    hipLaunchKernelGGL(mbstd_write_kernel<false>, dim3(MB_PARTS, G), dim3(256), 0, s, x, y, stats, (const float*)nullptr, (const float*)nullptr, 1, n, HW, C, CP);
    return (int)hipGetLastError();
}

// ---- exact-global minibatch stddev under data parallelism (optional mode): the two launches of pg_mbstd_fwd /
// pg_mbstd_tangent as separate entry points, so that the host can exchange the partial rows between them
extern "C" int pg_mbstd_stats(const float* x, float* stats, int G, int n, int HW, int C, pg_stream_t stream)
{
    if (!x || !stats || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    LAUNCH(mbstd_stats_kernel, dim3(MB_PARTS, G), dim3(256), 0, stream, x, stats, n, HW, C);
}

extern "C" int pg_mbstd_write(const float* x, float* y, float* stats, const float* gathered, int nranks,
                              int G, int n, int HW, int C, int CP, pg_stream_t stream)
{
    if (!x || !y || !stats || G <= 0 || n <= 0 || HW <= 0 || nranks < 1 || (nranks > 1 && !gathered)) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    LAUNCH(mbstd_write_kernel<false>, dim3(MB_PARTS, G), dim3(256), 0, stream, x, y, stats, (const float*)nullptr, gathered, nranks, n, HW, C, CP);
}

extern "C" int pg_mbstd_tangent_stats(const float* x, const float* tx, const float* stats, float* tstats,
                                      int G, int n, int HW, int C, pg_stream_t stream)
{
    if (!x || !tx || !stats || !tstats || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if (C & 3) return PG_E_ALIGN;
    LAUNCH(mbstd_tangent_stats_kernel, dim3(MB_PARTS, G), dim3(256), 0, stream, x, tx, stats, tstats, n, HW, C);
}

extern "C" int pg_mbstd_tangent_write(const float* tx, float* ty, float* tstats, const float* stats, const float* gathered, int nranks,
                                      int G, int n, int HW, int C, int CP, pg_stream_t stream)
{
    if (!tx || !ty || !tstats || !stats || G <= 0 || n <= 0 || HW <= 0 || nranks < 1 || (nranks > 1 && !gathered)) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    LAUNCH(mbstd_write_kernel<true>, dim3(MB_PARTS, G), dim3(256), 0, stream, tx, ty, tstats, stats, gathered, nranks, n, HW, C, CP);
}

extern "C" int pg_mbstd_gsum(const float* gy, const float* gy_first, float* out, int G, int n, int HW, int C, int CP, pg_stream_t stream)
{
    if (!out || (!gy && !gy_first) || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    LAUNCH(mbstd_gsum_kernel, dim3(G), dim3(256), 0, stream, gy, gy_first, out, n, HW, C, CP);
}

extern "C" int pg_mbstd_bwd_global(const float* gy, const float* x, const float* stats,
                                   const float* tx, const float* tstats, const float* gy_first,
                                   float* gx, const float* gsums, int nranks,
                                   int G, int n, int HW, int C, int CP, int apply_mask, float mask_slope, pg_stream_t stream)
{
    if (!x || !stats || !gx || !gsums || nranks < 1 || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if (!gy && !tx) return PG_E_ARG;
    if (tx && (!tstats || !gy_first)) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    LAUNCH(mbstd_bwd_kernel, dim3(MB_PARTS, G), dim3(256), 0, stream, gy, x, stats, tx, tstats, gy_first, gx, n, HW, C, CP, apply_mask, mask_slope,
           gsums, nranks);
}

extern "C" int pg_mbstd_tangent(const float* x, const float* tx, const float* stats, float* ty, float* tstats,
                                int G, int n, int HW, int C, int CP, pg_stream_t stream)
{
    if (!x || !tx || !stats || !ty || !tstats || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(mbstd_tangent_stats_kernel, dim3(MB_PARTS, G), dim3(256), 0, s, x, tx, stats, tstats, n, HW, C);
    hipLaunchKernelGGL(mbstd_write_kernel<true>, dim3(MB_PARTS, G), dim3(256), 0, s, tx, ty, tstats, stats, (const float*)nullptr, 1, n, HW, C, CP);
    return (int)hipGetLastError();
}

extern "C" int pg_mbstd_bwd(const float* gy, const float* x, const float* stats,
                            const float* tx, const float* tstats, const float* gy_first,
                            float* gx, int G, int n, int HW, int C, int CP, int apply_mask, float mask_slope,
                            pg_stream_t stream)
{
    if (!x || !stats || !gx || G <= 0 || n <= 0 || HW <= 0) return PG_E_ARG;
    if (!gy && !tx) return PG_E_ARG;
    if (tx && (!tstats || !gy_first)) return PG_E_ARG;
    if ((C & 3) || (CP & 3) || CP < C + 4) return PG_E_ALIGN;
    LAUNCH(mbstd_bwd_kernel, dim3(MB_PARTS, G), dim3(256), 0, stream, gy, x, stats, tx, tstats, gy_first, gx, n, HW, C, CP, apply_mask, mask_slope,
           (const float*)nullptr, 1);
}

extern "C" int pg_linear1_fwd(const float* h, const float* w, const float* b, float* s, int N, int C, pg_stream_t stream)
{
    if (!h || !w || !s || N <= 0 || C <= 0) return PG_E_ARG;
    LAUNCH(linear1_fwd_kernel, dim3(N), dim3(64), 0, stream, h, w, b, s, C);
}

extern "C" int pg_linear1_bwd_data(const float* gs, const float* w, const float* mask, float* gh, int N, int C,
                                   float mask_slope, pg_stream_t stream)
{
    if (!gs || !w || !gh || N <= 0 || C <= 0) return PG_E_ARG;
    LAUNCH(linear1_bwd_data_kernel, dim3(grid_for((size_t)N * C)), dim3(256), 0, stream, gs, w, mask, gh, N, C, mask_slope);
}

extern "C" int pg_linear1_wgrad(const float* gs, const float* h, float* dw, float* db, int N, int C, pg_stream_t stream)
{
    if (!gs || !h || !dw || N <= 0 || C <= 0) return PG_E_ARG;
    LAUNCH(linear1_wgrad_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, gs, h, dw, db, N, C);
}

extern "C" int pg_gp_mix(const float* real, const float* fake, const float* m, float* mixed, int N, int64_t E, pg_stream_t stream)
{
    if (!real || !fake || !m || !mixed || N <= 0 || E <= 0) return PG_E_ARG;
    if (E & 3) return PG_E_ALIGN;
    LAUNCH(gp_mix_kernel, dim3(grid_for((size_t)E >> 2, 256, 1024), N), dim3(256), 0, stream, real, fake, m, mixed, (size_t)E >> 2);
}

extern "C" int pg_row_sumsq(const float* g, float* ss, int N, int64_t E, pg_stream_t stream)
{
    if (!g || !ss || N <= 0 || E <= 0) return PG_E_ARG;
    if (E & 3) return PG_E_ALIGN;
    LAUNCH(row_sumsq_kernel, dim3(grid_for((size_t)E >> 2, 256 * 8, 256), N), dim3(256), 0, stream, g, ss, (size_t)E >> 2);
}

extern "C" int pg_gp_seed(const float* g, const float* ss, float* gp, float* u, int N, int64_t E,
                          float lambda, float target, float inv_n, pg_stream_t stream)
{
    if (!g || !ss || !gp || !u || N <= 0 || E <= 0) return PG_E_ARG;
    if (E & 3) return PG_E_ALIGN;
    LAUNCH(gp_seed_kernel, dim3(grid_for((size_t)E >> 2, 256, 1024), N), dim3(256), 0, stream, g, ss, gp, u, (size_t)E >> 2, lambda, target, inv_n);
}

extern "C" int pg_d_loss(const float* s, const float* gp, float* d_cost, float* d_real_loss, float* d_fake_loss,
                         float* gscore, int N, float eps, pg_stream_t stream)
{
    if (!s || !gp || !d_cost || !d_real_loss || !d_fake_loss || !gscore || N <= 0) return PG_E_ARG;
    LAUNCH(d_loss_kernel, dim3(1), dim3(64), 0, stream, s, gp, d_cost, d_real_loss, d_fake_loss, gscore, N, eps);
}

extern "C" int pg_g_loss(const float* s, float* g_cost, float* gscore, int N, pg_stream_t stream)
{
    if (!s || !g_cost || !gscore || N <= 0) return PG_E_ARG;
    LAUNCH(g_loss_kernel, dim3(1), dim3(64), 0, stream, s, g_cost, gscore, N);
}

extern "C" int pg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                       float eps, float bc1, float bc2_sqrt, float grad_scale, pg_stream_t stream)
{
    if (!p || !g || !m || !v || n <= 0) return PG_E_ARG;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return PG_E_ALIGN;
    if (beta1 == 0.f) {
        LAUNCH(adam_kernel<true>, dim3(grid_for(((size_t)n + 3) >> 2, 256, 2048)), dim3(256), 0, stream, p, g, m, v, (size_t)n,
               lr / bc1, beta1, beta2, eps, 1.f / bc2_sqrt, grad_scale);
    }
    LAUNCH(adam_kernel<false>, dim3(grid_for(((size_t)n + 3) >> 2, 256, 2048)), dim3(256), 0, stream, p, g, m, v, (size_t)n,
           lr / bc1, beta1, beta2, eps, 1.f / bc2_sqrt, grad_scale);
}

// U[0,1) draws for the gradient-penalty mixing factors (reference wgan_gp_loss.py:15-17: torch.cuda.FloatTensor(n, 1).uniform_()).
// Counter-based Philox4x32-10 (Salmon et al. 2011): element i of draw number `offset` under `seed` is a pure function of (seed,
// offset, i), so a step can be replayed and every rank draws its own stream; 24 random bits -> [0, 1) like torch's uniform_.
__global__ void uniform_kernel(float* __restrict__ out, long long n, unsigned long long seed, unsigned long long offset)
{
    const long long i4 = (long long)blockIdx.x * blockDim.x + threadIdx.x;          // one Philox block = 4 floats
    if (4 * i4 >= n) return;
    unsigned c0 = (unsigned)i4, c1 = (unsigned)((unsigned long long)i4 >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned c[4] = {c0, c1, c2, c3};
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (4 * i4 + j < n) out[4 * i4 + j] = (float)(c[j] >> 8) * (1.0f / 16777216.0f);
}

extern "C" int pg_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, pg_stream_t stream)
{
    if (!out || n <= 0) return PG_E_ARG;
    const long long blocks = ((n + 3) / 4 + 255) / 256;
    hipLaunchKernelGGL(uniform_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, (long long)n,
                       (unsigned long long)seed, (unsigned long long)offset);
    return (int)hipGetLastError();
}

extern "C" int pg_zero(void* p, int64_t bytes, pg_stream_t stream)
{
    if (!p || bytes < 0) return PG_E_ARG;
    if (bytes == 0) return 0;
    return (int)hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
}
