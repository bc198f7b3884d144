// Shared by conv_wino.hip (tile kernels) and conv_wino_strip.hip (row-streaming kernel): the launch parameter block of the
// Winograd F(2x2,3x3) convs, the lane-local output transform and the fused epilogues on the 2x2 outputs of a tile.  A lane
// (li = lane & 15, kk = lane >> 4) of either kernel holds the 16 Winograd-domain products of tile li for couts 4 kk .. 4 kk + 3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pggan_hip.h"
#include "bufload.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace pgw {

struct WinoP {
    const float* x; const float* u; const float* bias; const float* mask; float* y;
    int N, H, W, Cin, Cout, ups;
    float scale, slope, mask_slope;
    int lgTW, lgTH, TN, blocksW, blocksH, ncob, cout_minor; // workgroup = TN images x 2^lgTH x 2^lgTW tiles (64 tiles); cout blocks of 16
    unsigned mWT, mHT;                         // magic reciprocals of the halo region width / height in pixels
    int ntb, lgBW, lgBH; unsigned mDiv;        // second generation: tile blocks, log2(blocksW / blocksH), magic reciprocal of ncob (cout_minor) or ntb
    // fused epilogues (same semantics as the direct kernel, see pggan_hip.h)
    float* ypool; const float* pool_other; float pool_a, pool_b; int pool_only;
    float* yup; const float* upmask; float up_mul;
    int mask_bytes, y_bytes;                   // sign-byte activations (PG_FLAG_MASK_BYTES / PG_FLAG_Y_BYTES, pggan_hip.h)
    unsigned char* ysigns;                     // PG_FLAG_SIGNS_OUT
    float* pn_r; float pn_eps;                 // PixelNorm epilogue (pg_conv2d_wino_pixelnorm_nhwc): r[pixel] = rsqrt(mean_c y^2 + eps)
    const float* pnb_y; const float* pnb_r;    // adjoint of (LeakyReLU -> PixelNorm) on the (optionally pooled) result (pg_conv2d_wino_pnbwd_nhwc)
    // K split across workgroups (small maps at minibatch 3: fewer than one workgroup per CU otherwise): ksplit workgroups share a
    // (tile block, cout block), each takes kcper chunks; partial 2x2 outputs go to ks_part, the last arriver (ks_count) adds them
    // in split order and runs the fused epilogue
    int ksplit, kcper; unsigned mKs; float* ks_part; unsigned* ks_count;
#ifdef PG_WINO_TRACE
    unsigned long long* trace;                 // [workgroup][wave][chunk][8] s_memtime stamps (tools/exp/wino_trace.py)
#endif
};

#ifdef PG_WINO_TRACE
#define PG_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (p.trace && lane == 0 && blockIdx.x < 1024 && (k0 / KC) < 8) \
    p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + (k0 / KC)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define PG_STAMP(i) do { } while (0)
#endif

// Winograd-domain weights are stored in 8-channel packs: U[Cin/8][16 positions][Cout][8] — the slice a workgroup stages per
// 8-channel K chunk (16 positions x its couts x 8 channels) is then 16 contiguous pieces of couts*32 bytes, i.e. whole 128-byte
// cache lines that are consumed completely while they are hot, instead of 32 bytes out of every Cin*4-byte row of a
// [16][Cout][Cin] array (every line fetched from L2 four times, 2-4 us apart).  Measured on the second-generation kernel:
// 128.6 -> 112.0 us on n9 @64 128->256 with 16-channel packs (tools/sweeps/sweep_wino.py).
__host__ __device__ __forceinline__ size_t wino_u_index(size_t xi, size_t co, size_t ci, size_t Cout)
{
    return (((ci >> 3) * 16 + xi) * Cout + co) * 8 + (ci & 7);
}

__device__ __forceinline__ float4 sign_factors(unsigned char b, float slope)
{
    return make_float4((b & 1) ? 1.f : slope, (b & 2) ? 1.f : slope, (b & 4) ? 1.f : slope, (b & 8) ? 1.f : slope);
}

__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// Output transform Y = A^T M A of one lane's (tile, 4 couts) products (lane-local: the lane holds all 16 Winograd positions):
// yq[2 a + b] = output pixel (a, b) of the tile.
__device__ __forceinline__ void wino_output_transform(const f32x4 (&acc)[16], f32x4 (&yq)[4])
{
    f32x4 s[2][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s[0][j] = acc[0 + j] + acc[4 + j] + acc[8 + j];
        s[1][j] = acc[4 + j] - acc[8 + j] - acc[12 + j];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        yq[2 * a + 0] = s[a][0] + s[a][1] + s[a][2];
        yq[2 * a + 1] = s[a][1] - s[a][2] - s[a][3];
    }
}

// The fused epilogue on the 2x2 outputs of a tile.  cb: first of the lane's 4 couts, ni: image, (oy0, ox0): first output pixel.
__device__ __forceinline__ void wino_epilogue_q(const WinoP& p, const f32x4 (&yq)[4], int cb, int ni, int oy0, int ox0)
{
    if (cb >= p.Cout || ni >= p.N) return;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + cb);
    float4 ov[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = yq[q];
        // 32-bit element offsets (the host refuses tensors of 2^29 elements and more; yup has four times the pixels: < 2^31)
        const unsigned off = (((unsigned)ni * p.H + oy0 + (q >> 1)) * p.W + ox0 + (q & 1)) * p.Cout + cb;
        float4 o = make_float4(v[0] * p.scale, v[1] * p.scale, v[2] * p.scale, v[3] * p.scale);
        if (p.mask) {
            float4 f;
            if (p.mask_bytes) f = sign_factors(reinterpret_cast<const unsigned char*>(p.mask)[off >> 2], p.mask_slope);
            else {
                const float4 mk = *reinterpret_cast<const float4*>(p.mask + off);
                f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
            }
            o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
        } else {                                             // (explicit fma: the same rounding in every epilogue variant, whatever hipcc contracts)
            o = make_float4(fmaf(v[0], p.scale, bv.x), fmaf(v[1], p.scale, bv.y), fmaf(v[2], p.scale, bv.z), fmaf(v[3], p.scale, bv.w));
            o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
            o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
        }
        ov[q] = o;
        if (p.yup) {                                         // pool adjoint: four masked copies of every output
            const float k = p.up_mul * 0.25f;
            const unsigned W2 = 2u * p.W;
            const unsigned ubase = (((unsigned)ni * 2 * p.H + 2 * (oy0 + (q >> 1))) * W2 + 2 * (ox0 + (q & 1))) * p.Cout + cb;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const unsigned uo = ubase + ((unsigned)(dd >> 1) * W2 + (dd & 1)) * p.Cout;
                float4 w4 = make_float4(o.x * k, o.y * k, o.z * k, o.w * k);
                if (p.upmask) {
                    float4 f;
                    if (p.mask_bytes) f = sign_factors(reinterpret_cast<const unsigned char*>(p.upmask)[uo >> 2], p.mask_slope);
                    else {
                        const float4 mk = *reinterpret_cast<const float4*>(p.upmask + uo);
                        f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                        mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
                    }
                    w4.x *= f.x; w4.y *= f.y; w4.z *= f.z; w4.w *= f.w;
                }
                *reinterpret_cast<float4*>(p.yup + uo) = w4;
            }
        } else if (p.y_bytes) {                                  // only the sign is kept (the pooled output follows)
            reinterpret_cast<unsigned char*>(p.y)[off >> 2] =
                (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        } else if (!(p.ypool && p.pool_only)) {
            *reinterpret_cast<float4*>(p.y + off) = o;
        }
        if (p.ysigns)
            p.ysigns[off >> 2] = (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
    }
    if (p.ypool) {                                           // the 2x2 outputs of a tile ARE one pooled pixel
        float4 v;
        v.x = ((ov[0].x + ov[1].x) + (ov[2].x + ov[3].x)) * 0.25f; v.y = ((ov[0].y + ov[1].y) + (ov[2].y + ov[3].y)) * 0.25f;
        v.z = ((ov[0].z + ov[1].z) + (ov[2].z + ov[3].z)) * 0.25f; v.w = ((ov[0].w + ov[1].w) + (ov[2].w + ov[3].w)) * 0.25f;
        const unsigned poff = (((unsigned)ni * (p.H >> 1) + (oy0 >> 1)) * (p.W >> 1) + (ox0 >> 1)) * p.Cout + cb;
        if (p.pool_other) {
            const float4 q = *reinterpret_cast<const float4*>(p.pool_other + poff);
            v.x = fmaf(v.x, p.pool_a, p.pool_b * q.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * q.y);
            v.z = fmaf(v.z, p.pool_a, p.pool_b * q.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * q.w);
        } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
        *reinterpret_cast<float4*>(p.ypool + poff) = v;
    }
}

__device__ __forceinline__ void wino_epilogue(const WinoP& p, const f32x4 (&acc)[16], int cb, int ni, int oy0, int ox0)
{
    f32x4 yq[4];
    wino_output_transform(acc, yq);
    wino_epilogue_q(p, yq, cb, ni, oy0, ox0);
}

// The two epilogues most launches of a train step take, without the run-time option tests and the 64-bit address arithmetic of
// the general one: EPI_PLAIN = bias + LeakyReLU -> y (forward convs), EPI_MASKB = LeakyReLU' factors from sign bytes -> y
// (backward-data and tangent convs).  One raw buffer per image, 32-bit offsets; the host picks them when nothing else is asked for.
enum { EPI_GENERIC = 0, EPI_PLAIN = 1, EPI_MASKB = 2 };

template <int EPI>
__device__ __forceinline__ void wino_epilogue_fast(const WinoP& p, const f32x4 (&yq)[4], int cb, int ni, int oy0, int ox0)
{
    static_assert(EPI == EPI_PLAIN || EPI == EPI_MASKB, "specialised epilogues");
    // The 16 tiles of a wave lie in ONE image (a tile block holds >= 16 tiles of each of its images), so the image index is
    // wave-uniform: saying so keeps the per-image buffer descriptors in SGPRs.  (Left as a per-lane value hipcc wrapped every buffer
    // load / store of this epilogue in a waterfall loop -- readfirstlane x4, compare, saveexec, branch: ~12 instructions around each of
    // the 4 stores and 4 byte loads.)
    ni = __builtin_amdgcn_readfirstlane(ni);
    if (cb >= p.Cout || ni >= p.N) return;
    const unsigned npix = (unsigned)(p.H * p.W);
    const __amdgpu_buffer_rsrc_t ry = pg_make_rsrc(p.y + (size_t)ni * npix * p.Cout, npix * (unsigned)p.Cout * 4u);
    const unsigned pix0 = (unsigned)(oy0 * p.W + ox0);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned char mb[4] = {0, 0, 0, 0};
    if constexpr (EPI == EPI_PLAIN) {
        if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + cb);
    } else {
        const __amdgpu_buffer_rsrc_t rm = pg_make_rsrc(reinterpret_cast<const unsigned char*>(p.mask) + (size_t)ni * npix * (p.Cout >> 2),
                                                       npix * (unsigned)(p.Cout >> 2));
#pragma unroll
        for (int q = 0; q < 4; ++q)
            mb[q] = __builtin_amdgcn_raw_buffer_load_b8(rm, (int)((pix0 + (unsigned)((q >> 1) * p.W + (q & 1))) * (unsigned)(p.Cout >> 2) + (unsigned)(cb >> 2)), 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = yq[q];
        float4 o = make_float4(v[0] * p.scale, v[1] * p.scale, v[2] * p.scale, v[3] * p.scale);
        if constexpr (EPI == EPI_PLAIN) {
            o = make_float4(fmaf(v[0], p.scale, bv.x), fmaf(v[1], p.scale, bv.y), fmaf(v[2], p.scale, bv.z), fmaf(v[3], p.scale, bv.w));
            o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
            o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
        } else {
            const float4 f = sign_factors(mb[q], p.mask_slope);
            o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
        }
        __builtin_amdgcn_raw_buffer_store_b128(pg_u32x4{__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)},
                                               ry, (int)(((pix0 + (unsigned)((q >> 1) * p.W + (q & 1))) * (unsigned)p.Cout + (unsigned)cb) * 4u), 0, 0);
    }
}

template <int EPI>
__device__ __forceinline__ void wino_epilogue_sel(const WinoP& p, const f32x4 (&yq)[4], int cb, int ni, int oy0, int ox0)
{
    if constexpr (EPI == EPI_GENERIC) wino_epilogue_q(p, yq, cb, ni, oy0, ox0);
    else wino_epilogue_fast<EPI>(p, yq, cb, ni, oy0, ox0);
}


// conv -> bias -> LeakyReLU -> PixelNorm (network.py:44-52 after :32-41) for a workgroup that holds ALL couts of its tiles (Cout <=
// 16 NCB): the lane's 4 couts per block are squared and summed lane-locally, the four lanes of a tile (li + 16 kk) fold with two
// xor-shuffles.  Every lane takes part in the shuffles; lanes without a live (cout, image) contribute zeros and store nothing.
template <int NCB>
__device__ __forceinline__ void wino_epilogue_pixelnorm(const WinoP& p, const f32x4 (&acc)[NCB][16], int cb0, int ni, int oy0, int ox0)
{
    float4 o[NCB][4];
    float ss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        const int cb = cb0 + 16 * c;
        const bool live = cb < p.Cout;
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[c][0 + j] + acc[c][4 + j] + acc[c][8 + j];
            s[1][j] = acc[c][4 + j] - acc[c][8 + j] - acc[c][12 + j];
        }
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && p.bias) bv = *reinterpret_cast<const float4*>(p.bias + cb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = q >> 1;
            const f32x4 v = (q & 1) ? s[a][1] - s[a][2] - s[a][3] : s[a][0] + s[a][1] + s[a][2];
            float4 t = make_float4(v[0] * p.scale + bv.x, v[1] * p.scale + bv.y, v[2] * p.scale + bv.z, v[3] * p.scale + bv.w);
            t.x = t.x > 0.f ? t.x : t.x * p.slope; t.y = t.y > 0.f ? t.y : t.y * p.slope;
            t.z = t.z > 0.f ? t.z : t.z * p.slope; t.w = t.w > 0.f ? t.w : t.w * p.slope;
            if (!live) t = make_float4(0.f, 0.f, 0.f, 0.f);
            o[c][q] = t;
            ss[q] += (t.x * t.x + t.y * t.y) + (t.z * t.z + t.w * t.w);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ss[q] += __shfl_xor(ss[q], 16, 64);
        ss[q] += __shfl_xor(ss[q], 32, 64);
    }
    if (ni >= p.N) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float rr = rsqrtf(ss[q] / (float)p.Cout + p.pn_eps);
        const unsigned pix = ((unsigned)ni * p.H + oy0 + (q >> 1)) * p.W + ox0 + (q & 1);       // (32-bit: < 2^29 elements per tensor)
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = cb0 + 16 * c;
            if (cb < p.Cout) {
                const float4 t = o[c][q];
                *reinterpret_cast<float4*>(p.y + pix * p.Cout + cb) = make_float4(t.x * rr, t.y * rr, t.z * rr, t.w * rr);
            }
        }
        if (cb0 == 0) p.pn_r[pix] = rr;
    }
}


// Backward-data conv (optionally + the 2x2 pool that is the adjoint of the nearest x2 upsample) followed by the adjoint of the
// previous layer's (LeakyReLU -> PixelNorm): out = r * (g - y * mean_c(g * y)) * lrelu'(y), y / r saved by the forward pass
// (network.py:44-52).  Like the PixelNorm epilogue it needs every cout of a pixel in the workgroup (Cout <= 16 NCB).
template <int NCB>
__device__ __forceinline__ void wino_epilogue_pnbwd(const WinoP& p, const f32x4 (&acc)[NCB][16], int cb0, int ni, int oy0, int ox0)
{
    const bool pooled = p.ypool != nullptr;
    float4 g[NCB][4];                                        // pooled: only g[c][0] is used
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = acc[c][0 + j] + acc[c][4 + j] + acc[c][8 + j];
            s[1][j] = acc[c][4 + j] - acc[c][8 + j] - acc[c][12 + j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int a = q >> 1;
            const f32x4 v = (q & 1) ? s[a][1] - s[a][2] - s[a][3] : s[a][0] + s[a][1] + s[a][2];
            g[c][q] = make_float4(v[0] * p.scale, v[1] * p.scale, v[2] * p.scale, v[3] * p.scale);
        }
    }
    const int Ho = pooled ? p.H >> 1 : p.H, Wo = pooled ? p.W >> 1 : p.W;
    const int nq = pooled ? 1 : 4;
    if (pooled) {                                            // same form as the pooled epilogue of wino_epilogue
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = cb0 + 16 * c;
            float4 v;
            v.x = ((g[c][0].x + g[c][1].x) + (g[c][2].x + g[c][3].x)) * 0.25f; v.y = ((g[c][0].y + g[c][1].y) + (g[c][2].y + g[c][3].y)) * 0.25f;
            v.z = ((g[c][0].z + g[c][1].z) + (g[c][2].z + g[c][3].z)) * 0.25f; v.w = ((g[c][0].w + g[c][1].w) + (g[c][2].w + g[c][3].w)) * 0.25f;
            if (p.pool_other && cb < p.Cout && ni < p.N) {
                const unsigned poff = (((unsigned)ni * Ho + (oy0 >> 1)) * Wo + (ox0 >> 1)) * p.Cout + cb;
                const float4 o = *reinterpret_cast<const float4*>(p.pool_other + poff);
                v.x = fmaf(v.x, p.pool_a, p.pool_b * o.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * o.y);
                v.z = fmaf(v.z, p.pool_a, p.pool_b * o.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * o.w);
            } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
            g[c][0] = v;
        }
    }
    float* out = pooled ? p.ypool : p.y;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= nq) break;
        const int oy = pooled ? (oy0 >> 1) : oy0 + (q >> 1), ox = pooled ? (ox0 >> 1) : ox0 + (q & 1);
        const unsigned pix = ((unsigned)ni * Ho + oy) * Wo + ox;
        float4 yv[NCB];
        float dot = 0.f;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = cb0 + 16 * c;
            yv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cb < p.Cout && ni < p.N) yv[c] = *reinterpret_cast<const float4*>(p.pnb_y + pix * p.Cout + cb);
            dot += (g[c][q].x * yv[c].x + g[c][q].y * yv[c].y) + (g[c][q].z * yv[c].z + g[c][q].w * yv[c].w);
        }
        dot += __shfl_xor(dot, 16, 64);                       // (every lane takes part: lanes without a live pixel add zeros)
        dot += __shfl_xor(dot, 32, 64);
        if (ni >= p.N) continue;
        const float rr = p.pnb_r[pix], mean = dot / (float)p.Cout;
#pragma unroll
        for (int c = 0; c < NCB; ++c) {
            const int cb = cb0 + 16 * c;
            if (cb >= p.Cout) continue;
            const float4 gv = g[c][q], y4 = yv[c];
            *reinterpret_cast<float4*>(out + pix * p.Cout + cb) =
                make_float4(rr * (gv.x - y4.x * mean) * (y4.x > 0.f ? 1.f : p.mask_slope), rr * (gv.y - y4.y * mean) * (y4.y > 0.f ? 1.f : p.mask_slope),
                            rr * (gv.z - y4.z * mean) * (y4.z > 0.f ? 1.f : p.mask_slope), rr * (gv.w - y4.w * mean) * (y4.w > 0.f ? 1.f : p.mask_slope));
        }
    }
}

// Row-streaming form (conv_wino_strip.hip): the steps (two tile rows of a 64-column strip) of all strips form one sequence
// [image][strip][row step] of `total` steps; workgroup b handles steps [b spw, (b + 1) spw) for one of the ncog cout groups.
// strips, stepsH and ncog are powers of two.
struct WinoStripGeo { int strips, stepsH, total, spw, nrun, ncog, lgStrips, lgStepsH, lgCog, stagger; };
int launch_wino_strip(WinoP& p, int mode, hipStream_t s, char* name, size_t name_len);

}  // namespace pgw
