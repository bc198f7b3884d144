// Equalized-lr convolution as an implicit GEMM on v_mfma_f32_16x16x4_f32 (gfx950 / CDNA4).
//
//   forward / backward-data / GP-tangent :  conv_igemm_kernel      (pg_conv2d_nhwc)
//   weight gradient                      :  conv_wgrad_kernel      (pg_conv2d_wgrad_nhwc)
//
// Layout: activations NHWC, weights [KH][KW][Cout][Cin] => both MFMA operands are K-contiguous,
// so one ds_read_b128 per lane feeds four 16x16x4 k-steps (lane (i = l&15, kk = l>>4) owns
// channels 4kk..4kk+3 of a 16-channel chunk; k-step s contracts channels {s, 4+s, 8+s, 12+s}).
// MFMA roles: A = weights (row i = cout), B = activations (col j = pixel); the C/D fragment then
// holds 4 consecutive couts of one pixel per lane -> 16-byte NHWC stores.
// Exact fp32 FMA chain (no reduced-precision path): parity with the fp32 CPU oracle to ~1e-6.
//
// Pipeline (both kernels): per-thread load descriptors are computed once; the NEXT K-chunk / pixel
// tile is fetched global->VGPR while the MFMAs of the current one run out of LDS (one LDS buffer,
// two workgroups per CU), and inside a chunk the fragments of the next tap / k-step are read
// from LDS while the current MFMAs issue.  LDS row strides are chosen conflict-free for the
// lane groups of ds_read_b128 / ds_read_b32 on gfx950 (24 floats for 16-channel rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pggan_hip.h"
#include "bufload.h"
#include "convp.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#include <cstdio>

namespace {

thread_local char g_last_kernel[96] = "";     // symbol of the last conv kernel launched by this thread
thread_local int g_tune[4] = {-1, -1, -1, -1}; // tuning overrides (pg_debug_set_tuning): [0] conv tile, [1] wgrad config, [2] conv split-K, [3] 1: generic path for the 4x4 boundary layers, 2: generic path for the 8/16-cout layers, 3: unfused pooling, 10: unfused unpooling, 11 / 12: unfused PixelNorm forward / adjoint, 20: tile kernels instead of the row-streaming ones, 21: 4x4 -> 1x1 layer on one workgroup per cout block

template <typename K>
inline int set_smem(K kern, size_t smem)
{
    if (smem > 160 * 1024) return PG_E_UNSUP;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    return 0;
}

template <int VEC> __device__ __forceinline__ void lds_load(const float* p, float (&o)[VEC]);
template <> __device__ __forceinline__ void lds_load<4>(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void lds_load<2>(const float* p, float (&o)[2]) {
    float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[1] = v.y;
}
template <> __device__ __forceinline__ void lds_load<1>(const float* p, float (&o)[1]) { o[0] = *p; }

using pgk::ConvP;
using pgk::pg_sign_factors;
using pgk::pg_sign_byte;

// LDS row stride (floats) of a KC-channel row: conflict-free for the gfx950 lane groups
//   VEC=4 (ds_read_b128, 4x16 lanes, 64 banks): 24   VEC=2 (ds_read_b64): 12   VEC=1: 8
template <int VEC> struct RowStride { static constexpr int value = VEC == 4 ? 24 : (VEC == 2 ? 12 : 8); };

// Upper bound of the halo pixels of one tile (sizes the register prefetch): KS=3 with TH,TW >= 4 needs at most
// 2.25*BPX; tiles of >= 512 pixels are always 32 wide (make_geom), so (BPX/32+2)*34 is exact there.
constexpr int halo_max(int KS, int BPX)
{
    return KS == 1 ? BPX : (KS == 4 ? 16 * BPX : (BPX >= 512 ? (BPX / 32 + 2) * 34 : (BPX * 9) / 4));
}

// One workgroup (4 waves) computes BCO couts x BPX output pixels; the pixel tile is
// TN images x TH x TW (powers of two) so that the (KS-1)-halo of the input is staged once in LDS
// and every tap is a shifted read of the same tile.
template <int KS, int VEC, int WAVES_CO, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p)
{
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int BCO = 16 * WM * WAVES_CO, BPX = 16 * WN * WAVES_PX;
    constexpr int KC = 4 * VEC, KCP = RowStride<VEC>::value;
    constexpr int TAPS = KS * KS;
    constexpr int WEL = TAPS * BCO * VEC;                       // float4 elements of one weight chunk
    constexpr int WPT = (WEL + 255) / 256;
    constexpr int XMAX = halo_max(KS, BPX);
    constexpr int XPT = (XMAX * VEC + 255) / 256;
    extern __shared__ __align__(16) float lds[];

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
    const int HT = TH + KS - 1, WT = TW + KS - 1;
    float* wt = lds;                          // [TAPS][BCO][KCP]
    float* xt = lds + TAPS * BCO * KCP;       // [TN][HT][WT][KCP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_co = wave % WAVES_CO, wave_px = wave / WAVES_CO;
    const int li = lane & 15, kk = lane >> 4;

    int t = (int)pg_xcd_remap(blockIdx.x, gridDim.x);       // contiguous tile ranges per XCD (bufload.h)
    const int tw_i = t % p.tilesW; t /= p.tilesW;
    const int th_i = t % p.tilesH; t /= p.tilesH;
    const int n0 = t * p.TN;
    const int oh0 = th_i << p.lgTH, ow0 = tw_i << p.lgTW;
    const int co0 = blockIdx.y * BCO;

    int pixbase[WN], wbase[WM];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int j = (wave_px * WN + n) * 16 + li;
        const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
        pixbase[n] = ((tn * HT + th) * WT + tw) * KCP + VEC * kk;
    }
#pragma unroll
    for (int m = 0; m < WM; ++m) wbase[m] = ((wave_co * WM + m) * 16 + li) * KCP + VEC * kk;

    // ---- per-thread load descriptors (element offsets without the channel-chunk offset; -1 = zero fill)
    const int xH = p.ups ? (p.Hin >> 1) : p.Hin, xW = p.ups ? (p.Win >> 1) : p.Win;
    const int npix = p.TN * HT * WT;
    // global loads go through raw buffers (bufload.h): byte offsets, PG_OOB = zero fill, x relative to image n0
    const size_t img = (size_t)xH * xW * p.Cin;
    const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(p.x + (size_t)n0 * img, (unsigned)(min(p.TN, p.N - n0) * img * 4));
    const __amdgpu_buffer_rsrc_t rw = pg_make_rsrc(p.w, (unsigned)((size_t)TAPS * p.Cout * p.Cin * 4));
    unsigned wsrc[WPT], xsrc[XPT];
    int wdst[WPT], xdst[XPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / VEC, v = idx - r * VEC;
        const int tap = r / BCO, col = r - tap * BCO, co = co0 + col;
        wdst[i] = idx < WEL ? r * KCP + 4 * v : -1;
        wsrc[i] = (idx < WEL && co < p.Cout) ? 4u * (unsigned)((tap * p.Cout + co) * p.Cin + 4 * v) : PG_OOB;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VEC, v = idx - q * VEC;
        const int r2 = (int)__umulhi((unsigned)q, p.mWT), tw = q - r2 * WT;       // q / WT, q % WT (q < 2^16)
        const int tn = (int)__umulhi((unsigned)r2, p.mHT), th = r2 - tn * HT;
        const int n = n0 + tn;
        int ih = oh0 + th - p.pad, iw = ow0 + tw - p.pad;
        const bool in_tile = q < npix;
        const bool ok = in_tile && n < p.N && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        if (p.ups) { ih >>= 1; iw >>= 1; }
        xdst[i] = in_tile ? q * KCP + 4 * v : -1;
        xsrc[i] = ok ? 4u * (unsigned)(((tn * xH + ih) * xW + iw) * p.Cin + 4 * v) : PG_OOB;
    }
    int tapoff[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) tapoff[tp] = ((tp / KS) * WT + (tp % KS)) * KCP;

    f32x4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};                  // second accumulation chain of the single-tile variants

    const int nchunks = p.Cin / KC;
    const int cper = (nchunks + p.ksplit - 1) / p.ksplit;
    const int kc_begin = blockIdx.z * cper;
    const int kc_end = min(nchunks, kc_begin + cper);

    float4 wreg[WPT], xreg[XPT];
    auto fetch = [&](int kc) {
        const int k0 = kc * KC;
#pragma unroll
        for (int i = 0; i < WPT; ++i) wreg[i] = pg_buf_load4(rw, wsrc[i], 4u * (unsigned)k0);
#pragma unroll
        for (int i = 0; i < XPT; ++i) xreg[i] = pg_buf_load4(rx, xsrc[i], 4u * (unsigned)k0);
    };

    if (kc_begin < kc_end) fetch(kc_begin);
    for (int kc = kc_begin; kc < kc_end; ++kc) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) if (wdst[i] >= 0) *reinterpret_cast<float4*>(wt + wdst[i]) = wreg[i];
#pragma unroll
        for (int i = 0; i < XPT; ++i) if (xdst[i] >= 0) *reinterpret_cast<float4*>(xt + xdst[i]) = xreg[i];
        __syncthreads();
        if (kc + 1 < kc_end) fetch(kc + 1);              // in flight while the MFMAs below run

        float a[2][WM][VEC], b[2][WN][VEC];
#pragma unroll
        for (int m = 0; m < WM; ++m) lds_load<VEC>(wt + wbase[m], a[0][m]);
#pragma unroll
        for (int n = 0; n < WN; ++n) lds_load<VEC>(xt + pixbase[n] + tapoff[0], b[0][n]);
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const int cur = tp & 1, nxt = cur ^ 1;
            if (tp + 1 < TAPS) {
#pragma unroll
                for (int m = 0; m < WM; ++m) lds_load<VEC>(wt + (tp + 1) * BCO * KCP + wbase[m], a[nxt][m]);
#pragma unroll
                for (int n = 0; n < WN; ++n) lds_load<VEC>(xt + pixbase[n] + tapoff[tp + 1], b[nxt][n]);
            }
            // pin the order "issue next tap's LDS reads, THEN this tap's MFMAs": hipcc otherwise sinks the reads
            // next to their first use and every tap starts with an exposed LDS round trip
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < VEC; ++s) {
                if constexpr (WM * WN == 1) {
                    // one output tile per wave: alternate two accumulators so that two MFMAs on the same accumulator are
                    // never adjacent (32-cycle issue, 40-cycle dependent latency, ~43 extra with anything in between)
                    if (s & 1) acc_odd = MFMA16(a[cur][0][s], b[cur][0][s], acc_odd);
                    else acc[0][0] = MFMA16(a[cur][0][s], b[cur][0][s], acc[0][0]);
                } else {
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n) acc[m][n] = MFMA16(a[cur][m][s], b[cur][n][s], acc[m][n]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if constexpr (WM * WN == 1 && VEC > 1) acc[0][0] += acc_odd;

    // epilogue: lane holds couts cb..cb+3 of pixel j
    if (p.pn_r != nullptr) {                     // conv -> bias -> LeakyReLU -> PixelNorm; the workgroup holds every cout of its pixels
        float4 o[WM][WN];
        float ss[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n) ss[n] = 0.f;
#pragma unroll
        for (int m = 0; m < WM; ++m) {
            const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.bias && cb < p.Cout) bv = *reinterpret_cast<const float4*>(p.bias + cb);
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                float4 v = make_float4(acc[m][n][0] * p.scale + bv.x, acc[m][n][1] * p.scale + bv.y,
                                       acc[m][n][2] * p.scale + bv.z, acc[m][n][3] * p.scale + bv.w);
                v.x = v.x > 0.f ? v.x : v.x * p.slope; v.y = v.y > 0.f ? v.y : v.y * p.slope;
                v.z = v.z > 0.f ? v.z : v.z * p.slope; v.w = v.w > 0.f ? v.w : v.w * p.slope;
                if (cb >= p.Cout) v = make_float4(0.f, 0.f, 0.f, 0.f);
                o[m][n] = v;
                ss[n] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
            }
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            ss[n] += __shfl_xor(ss[n], 16, 64);
            ss[n] += __shfl_xor(ss[n], 32, 64);
            const float rr = rsqrtf(ss[n] / (float)p.Cout + p.pn_eps);
            const int j = (wave_px * WN + n) * 16 + li;
            const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
            const int ni = n0 + tn;
            if (ni >= p.N) continue;
            const size_t pix = ((size_t)ni * p.Hout + oh0 + th) * p.Wout + ow0 + tw;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
                if (cb >= p.Cout) continue;
                *reinterpret_cast<float4*>(p.y + pix * p.Cout + cb) =
                    make_float4(o[m][n].x * rr, o[m][n].y * rr, o[m][n].z * rr, o[m][n].w * rr);
            }
            if (kk == 0) p.pn_r[pix] = rr;
        }
        return;
    }
    if (p.pnb_y != nullptr) {                    // backward-data conv + adjoint of the previous layer's LeakyReLU -> PixelNorm
        float4 gq[WM][WN], yq[WM][WN];
        float dot[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n) dot[n] = 0.f;
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int j = (wave_px * WN + n) * 16 + li;
            const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
            const int ni = n0 + tn;
            const size_t pix = ((size_t)(ni < p.N ? ni : 0) * p.Hout + oh0 + th) * p.Wout + ow0 + tw;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f), yv = gv;
                if (cb < p.Cout && ni < p.N) {
                    gv = make_float4(acc[m][n][0] * p.scale, acc[m][n][1] * p.scale, acc[m][n][2] * p.scale, acc[m][n][3] * p.scale);
                    yv = *reinterpret_cast<const float4*>(p.pnb_y + pix * p.Cout + cb);
                }
                gq[m][n] = gv; yq[m][n] = yv;
                dot[n] += (gv.x * yv.x + gv.y * yv.y) + (gv.z * yv.z + gv.w * yv.w);
            }
        }
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            dot[n] += __shfl_xor(dot[n], 16, 64);
            dot[n] += __shfl_xor(dot[n], 32, 64);
            const int j = (wave_px * WN + n) * 16 + li;
            const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
            const int ni = n0 + tn;
            if (ni >= p.N) continue;
            const size_t pix = ((size_t)ni * p.Hout + oh0 + th) * p.Wout + ow0 + tw;
            const float rr = p.pnb_r[pix], mean = dot[n] / (float)p.Cout;
#pragma unroll
            for (int m = 0; m < WM; ++m) {
                const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
                if (cb >= p.Cout) continue;
                const float4 gv = gq[m][n], yv = yq[m][n];
                float4 o;
                o.x = rr * (gv.x - yv.x * mean) * (yv.x > 0.f ? 1.f : p.mask_slope);
                o.y = rr * (gv.y - yv.y * mean) * (yv.y > 0.f ? 1.f : p.mask_slope);
                o.z = rr * (gv.z - yv.z * mean) * (yv.z > 0.f ? 1.f : p.mask_slope);
                o.w = rr * (gv.w - yv.w * mean) * (yv.w > 0.f ? 1.f : p.mask_slope);
                *reinterpret_cast<float4*>(p.y + pix * p.Cout + cb) = o;
            }
        }
        return;
    }
    const bool pooling = p.ypool != nullptr && p.ksplit == 1;
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
        const bool cvalid = cb < p.Cout;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias && cvalid) bv = *reinterpret_cast<const float4*>(p.bias + cb);
        float4 ov[WN];
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            ov[n] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int j = (wave_px * WN + n) * 16 + li;
            const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
            const int ni = n0 + tn;
            if (!cvalid || ni >= p.N) continue;
            const size_t off = (((size_t)ni * p.Hout + oh0 + th) * p.Wout + ow0 + tw) * p.Cout + cb;
            float4 o;
            o.x = acc[m][n][0] * p.scale; o.y = acc[m][n][1] * p.scale;
            o.z = acc[m][n][2] * p.scale; o.w = acc[m][n][3] * p.scale;
            if (p.ksplit > 1) {
                atomicAdd(p.y + off + 0, o.x); atomicAdd(p.y + off + 1, o.y);
                atomicAdd(p.y + off + 2, o.z); atomicAdd(p.y + off + 3, o.w);
                continue;
            }
            if (p.mask) {
                float4 f;
                if (p.mask_bytes) f = pg_sign_factors(reinterpret_cast<const unsigned char*>(p.mask)[off >> 2], p.mask_slope);
                else {
                    const float4 mk = *reinterpret_cast<const float4*>(p.mask + off);
                    f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                    mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
                }
                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
            } else {
                o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
                o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
            }
            if (p.yup) {                                         // unpool: four masked copies, y itself is not needed
                const float k = p.up_mul * 0.25f;
                const size_t W2 = (size_t)2 * p.Wout;
                const size_t ubase = (((size_t)ni * 2 * p.Hout + 2 * (oh0 + th)) * W2 + 2 * (ow0 + tw)) * p.Cout + cb;
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const size_t uo = ubase + ((size_t)(d >> 1) * W2 + (d & 1)) * p.Cout;
                    float4 v = make_float4(o.x * k, o.y * k, o.z * k, o.w * k);
                    if (p.upmask) {
                        float4 f;
                        if (p.mask_bytes) f = pg_sign_factors(reinterpret_cast<const unsigned char*>(p.upmask)[uo >> 2], p.mask_slope);
                        else {
                            const float4 mk = *reinterpret_cast<const float4*>(p.upmask + uo);
                            f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                            mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
                        }
                        v.x *= f.x; v.y *= f.y; v.z *= f.z; v.w *= f.w;
                    }
                    *reinterpret_cast<float4*>(p.yup + uo) = v;
                }
                continue;
            }
            if (p.y_bytes) reinterpret_cast<unsigned char*>(p.y)[off >> 2] = pg_sign_byte(o);     // only the sign is kept (pooled output below)
            else if (!(pooling && p.pool_only)) *reinterpret_cast<float4*>(p.y + off) = o;
            if (p.ysigns) p.ysigns[off >> 2] = pg_sign_byte(o);
            ov[n] = o;
        }
        if (pooling) {
            // 2x2 mean inside the wave: the horizontal neighbour is lane^1; the vertical neighbour is lane^TW for
            // tiles <= 8 wide, the next 16-pixel group (TW 16) or the one after (TW 32) otherwise -- launch_conv
            // restricts TW so that this group belongs to the same wave.  Same summation order as pg_avgpool2_fwd.
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                ov[n].x += __shfl_xor(ov[n].x, 1, 64); ov[n].y += __shfl_xor(ov[n].y, 1, 64);
                ov[n].z += __shfl_xor(ov[n].z, 1, 64); ov[n].w += __shfl_xor(ov[n].w, 1, 64);
            }
            bool rowlead[WN];
            if (p.lgTW <= 3) {
#pragma unroll
                for (int n = 0; n < WN; ++n) {
                    ov[n].x += __shfl_xor(ov[n].x, TW, 64); ov[n].y += __shfl_xor(ov[n].y, TW, 64);
                    ov[n].z += __shfl_xor(ov[n].z, TW, 64); ov[n].w += __shfl_xor(ov[n].w, TW, 64);
                    rowlead[n] = (li & TW) == 0;
                }
            } else if (p.lgTW == 4) {
#pragma unroll
                for (int n = 0; n < WN; ++n) rowlead[n] = (n & 1) == 0;
                if constexpr (WN >= 2) {
#pragma unroll
                    for (int n = 0; n < WN; n += 2) {
                        ov[n].x += ov[n + 1].x; ov[n].y += ov[n + 1].y; ov[n].z += ov[n + 1].z; ov[n].w += ov[n + 1].w;
                    }
                }
            } else {
#pragma unroll
                for (int n = 0; n < WN; ++n) rowlead[n] = (n & 2) == 0;
                if constexpr (WN >= 4) {
#pragma unroll
                    for (int n = 0; n < WN; n += 4) {
                        ov[n].x += ov[n + 2].x; ov[n].y += ov[n + 2].y; ov[n].z += ov[n + 2].z; ov[n].w += ov[n + 2].w;
                        ov[n + 1].x += ov[n + 3].x; ov[n + 1].y += ov[n + 3].y; ov[n + 1].z += ov[n + 3].z; ov[n + 1].w += ov[n + 3].w;
                    }
                }
            }
#pragma unroll
            for (int n = 0; n < WN; ++n) {
                const int j = (wave_px * WN + n) * 16 + li;
                const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
                const int ni = n0 + tn;
                if (!cvalid || ni >= p.N || !rowlead[n] || (li & 1)) continue;
                const size_t poff = (((size_t)ni * (p.Hout >> 1) + ((oh0 + th) >> 1)) * (p.Wout >> 1) + ((ow0 + tw) >> 1)) * p.Cout + cb;
                float4 v = make_float4(ov[n].x * 0.25f, ov[n].y * 0.25f, ov[n].z * 0.25f, ov[n].w * 0.25f);
                if (p.pool_other) {
                    const float4 q = *reinterpret_cast<const float4*>(p.pool_other + poff);
                    v.x = fmaf(v.x, p.pool_a, p.pool_b * q.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * q.y);   // same form as
                    v.z = fmaf(v.z, p.pool_a, p.pool_b * q.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * q.w);   // avgpool2_fwd_kernel
                } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
                *reinterpret_cast<float4*>(p.ypool + poff) = v;
            }
        }
    }
}

// Deferred epilogue of a split-K launch: y = mask ? y*lrelu'(mask) : lrelu(y + bias)   (in place)
__global__ __launch_bounds__(256) void conv_epilogue_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                            const float* __restrict__ mask, size_t npix, int Cout,
                                                            float slope, float mask_slope)
{
    const int c4n = Cout >> 2;
    const size_t total = npix * c4n;
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* m4 = reinterpret_cast<const float4*>(mask);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        float4 o = y4[i];
        if (mask) {
            const float4 mk = m4[i];
            o.x *= mk.x > 0.f ? 1.f : mask_slope; o.y *= mk.y > 0.f ? 1.f : mask_slope;
            o.z *= mk.z > 0.f ? 1.f : mask_slope; o.w *= mk.w > 0.f ? 1.f : mask_slope;
        } else {
            if (bias) {
                const float4 bv = *reinterpret_cast<const float4*>(bias + 4 * (i % c4n));
                o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
            }
            o.x = o.x > 0.f ? o.x : o.x * slope; o.y = o.y > 0.f ? o.y : o.y * slope;
            o.z = o.z > 0.f ? o.z : o.z * slope; o.w = o.w > 0.f ? o.w : o.w * slope;
        }
        y4[i] = o;
    }
}

// ------------------------------------------------------------------------------------------
using pgk::WgP;

#ifdef PG_WINO_TRACE
#define PG_WSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (p.trace && lane == 0 && blockIdx.x < 1024 && (tile - t_begin) < 8) \
    p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + (tile - t_begin)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
thread_local unsigned long long* g_wgrad_trace = nullptr;
#else
#define PG_WSTAMP(i) do { } while (0)
#endif

// row stride == 16 (mod 32): the two 32-lane groups of ds_read_b32 hit disjoint banks
template <int B> struct PixStride { static constexpr int value = (B % 32 == 16) ? B : B + 16; };

// dW[tap][co][ci] = sum over pixels: A = gz (row i = cout), B = shifted x (col j = cin), the MFMA
// k index runs over PIXELS (4 per instruction).  One workgroup owns a (BCO x BCI) block of every
// tap and a slice of the pixel tiles; its WAVES_K waves split the pixels of a tile and are reduced
// through LDS before ONE commit per workgroup (plain += when it is the only writer, else atomics).
template <int KS, int WM, int WN, int WAVES_CO, int WAVES_CI, int BPX>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgP p)
{
    constexpr int WAVES_K = 4 / (WAVES_CO * WAVES_CI);
    constexpr int BCO = 16 * WM * WAVES_CO, BCI = 16 * WN * WAVES_CI;
    constexpr int SZ = PixStride<BCO>::value, SX = PixStride<BCI>::value;
    constexpr int TAPS = KS * KS;
    constexpr int ZV = BCO / 4, XV = BCI / 4;
    constexpr int ZPT = (BPX * ZV + 255) / 256;
    constexpr int XMAX = KS == 1 ? BPX : (KS == 3 ? (BPX * 9) / 4 : 16 * BPX);
    constexpr int XPT = (XMAX * XV + 255) / 256;
    extern __shared__ __align__(16) float lds[];

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
    const int HT = TH + KS - 1, WT = TW + KS - 1;
    float* gzt = lds;                        // [BPX][SZ]
    float* xt = lds + BPX * SZ;              // [TN*HT*WT][SX]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_co = wave % WAVES_CO, wave_ci = (wave / WAVES_CO) % WAVES_CI, wave_k = wave / (WAVES_CO * WAVES_CI);
    const int li = lane & 15, kk = lane >> 4;
    const int co0 = blockIdx.y * BCO, ci0 = blockIdx.z * BCI;
    const bool do_bias = (p.db != nullptr) && blockIdx.z == 0 && wave_ci == 0;

    f32x4 acc[TAPS][WM][WN];
    f32x4 accb[WM];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int m = 0; m < WM; ++m)
#pragma unroll
            for (int n = 0; n < WN; ++n) acc[tp][m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < WM; ++m) accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int xH = p.ups ? (p.Hin >> 1) : p.Hin, xW = p.ups ? (p.Win >> 1) : p.Win;
    const int npix = p.TN * HT * WT;

    // ---- per-thread load descriptors: tile-relative coordinates (no div/mod in the tile loop)
    int zq[ZPT], zc[ZPT];                    // pixel-in-tile, cout offset
    int xq[XPT], xc[XPT], xdst[XPT];         // packed (tn,th,tw) of the halo pixel, cin offset, LDS offset
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / ZV, v = idx - q * ZV;
        zq[i] = idx < BPX * ZV ? q : -1;
        zc[i] = co0 + 4 * v;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / XV, v = idx - q * XV;
        const int r2 = (int)__umulhi((unsigned)q, p.mWT), tw = q - r2 * WT;
        const int tn = (int)__umulhi((unsigned)r2, p.mHT), th = r2 - tn * HT;
        xq[i] = q < npix ? ((tn << 20) | (th << 10) | tw) : -1;
        xc[i] = ci0 + 4 * v;
        xdst[i] = q * SX + 4 * v;
    }
    int tapoff[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) tapoff[tp] = ((tp / KS) * WT + (tp % KS)) * SX;

    float4 zreg[ZPT], xreg[XPT];
    auto fetch = [&](int tile) {
        int t = tile;
        const int tw_i = t % p.tilesW; t /= p.tilesW;
        const int th_i = t % p.tilesH; t /= p.tilesH;
        const int n0 = t * p.TN;
        const int oh0 = th_i << p.lgTH, ow0 = tw_i << p.lgTW;
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (zq[i] >= 0) {
                const int q = zq[i];
                const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
                const int n = n0 + tn;
                if (n < p.N && zc[i] < p.Cout)
                    val = *reinterpret_cast<const float4*>(p.gz + (((size_t)n * p.Hout + oh0 + th) * p.Wout + ow0 + tw) * p.Cout + zc[i]);
            }
            zreg[i] = val;
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (xq[i] >= 0) {
                const int tw = xq[i] & 1023, th = (xq[i] >> 10) & 1023, tn = xq[i] >> 20;
                const int n = n0 + tn;
                int ih = oh0 + th - p.pad, iw = ow0 + tw - p.pad;
                if (n < p.N && xc[i] < p.Cin && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
                    if (p.ups) { ih >>= 1; iw >>= 1; }
                    val = *reinterpret_cast<const float4*>(p.x + (((size_t)n * xH + ih) * xW + iw) * p.Cin + xc[i]);
                }
            }
            xreg[i] = val;
        }
    };

    const int t_begin = (int)pg_xcd_remap(blockIdx.x, gridDim.x) * p.tiles_per_block;   // neighbouring tile ranges on one XCD
    const int t_end = min(t_begin + p.tiles_per_block, p.ntiles);
    constexpr int NSTEPS = BPX / 4;

    if (t_begin < t_end) fetch(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
#pragma unroll
        for (int i = 0; i < ZPT; ++i)
            if (zq[i] >= 0) *reinterpret_cast<float4*>(gzt + zq[i] * SZ + (zc[i] - co0)) = zreg[i];
#pragma unroll
        for (int i = 0; i < XPT; ++i)
            if (xq[i] >= 0) *reinterpret_cast<float4*>(xt + xdst[i]) = xreg[i];
        __syncthreads();
        if (tile + 1 < t_end) fetch(tile + 1);            // in flight while the MFMAs below run

        // k-steps of this wave: step = wave_k, wave_k + WAVES_K, ...  (4 pixels each, same tile row)
        auto frag_addr = [&](int step, int& aoff, int& boff) {
            const int q = 4 * step + kk;
            const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
            aoff = q * SZ + wave_co * WM * 16 + li;
            boff = ((tn * HT + th) * WT + tw) * SX + wave_ci * WN * 16 + li;
        };
        float a[2][WM], b[2][TAPS][WN];
        auto load_frags = [&](int step, float (&af)[WM], float (&bf)[TAPS][WN]) {
            int ao, bo; frag_addr(step, ao, bo);
#pragma unroll
            for (int m = 0; m < WM; ++m) af[m] = gzt[ao + m * 16];
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int n = 0; n < WN; ++n) bf[tp][n] = xt[bo + tapoff[tp] + n * 16];
        };
        auto mfmas = [&](const float (&af)[WM], const float (&bf)[TAPS][WN]) {
            if (do_bias) {
#pragma unroll
                for (int m = 0; m < WM; ++m) accb[m] = MFMA16(af[m], 1.0f, accb[m]);
            }
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int m = 0; m < WM; ++m)
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[tp][m][n] = MFMA16(af[m], bf[tp][n], acc[tp][m][n]);
        };
        constexpr int T = NSTEPS / WAVES_K;              // even for every instantiated shape
        static_assert(T % 2 == 0, "k-steps per wave must be even");
        load_frags(wave_k, a[0], b[0]);
        for (int s = 0; s < T; s += 2) {                 // ping-pong: next step's LDS reads under this step's MFMAs
            load_frags(wave_k + (s + 1) * WAVES_K, a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);           // keep the reads ahead of the MFMAs (see conv_igemm_kernel)
            mfmas(a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 2 < T) load_frags(wave_k + (s + 2) * WAVES_K, a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- reduce the WAVES_K partial sums through LDS (tile buffers are free now), then commit once
    if (WAVES_K > 1) {
        // layout: red[(wave_co,wave_ci)][tile index t = (tp*WM+m)*WN+n (+bias tiles)][lane*4+r]
        constexpr int NT = TAPS * WM * WN + WM;
        float* red = lds + (wave_co + WAVES_CO * wave_ci) * NT * 256;
        for (int w = 0; w < WAVES_K; ++w) {
            if (wave_k == w) {
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n) {
                            float4* r4 = reinterpret_cast<float4*>(red + (((tp * WM + m) * WN + n) * 64 + lane) * 4);
                            float4 v = make_float4(acc[tp][m][n][0], acc[tp][m][n][1], acc[tp][m][n][2], acc[tp][m][n][3]);
                            if (w > 0) { const float4 o = *r4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                            if (w == WAVES_K - 1) acc[tp][m][n] = f32x4{v.x, v.y, v.z, v.w}; else *r4 = v;
                        }
#pragma unroll
                for (int m = 0; m < WM; ++m) {
                    float4* r4 = reinterpret_cast<float4*>(red + ((TAPS * WM * WN + m) * 64 + lane) * 4);
                    float4 v = make_float4(accb[m][0], accb[m][1], accb[m][2], accb[m][3]);
                    if (w > 0) { const float4 o = *r4; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    if (w == WAVES_K - 1) accb[m] = f32x4{v.x, v.y, v.z, v.w}; else *r4 = v;
                }
            }
            __syncthreads();
        }
        if (wave_k != WAVES_K - 1) return;
    }

    // commit: C/D fragment row = 4*kk + reg -> cout, col = li -> cin
#pragma unroll
    for (int m = 0; m < WM; ++m) {
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int ci = ci0 + (wave_ci * WN + n) * 16 + li;
            if (ci >= p.Cin) continue;
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wave_co * WM + m) * 16 + 4 * kk + r;
                    if (co < p.Cout) {
                        float* dst = p.dw + ((size_t)(tp * p.Cout + co) * p.Cin + ci);
                        const float v = acc[tp][m][n][r] * p.scale;
                        if (p.atomic) atomicAdd(dst, v); else *dst += v;
                    }
                }
        }
        if (do_bias && li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wave_co * WM + m) * 16 + 4 * kk + r;
                if (co < p.Cout) { if (p.atomic) atomicAdd(p.db + co, accb[m][r]); else p.db[co] += accb[m][r]; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// The two 4x4 layers at the 1x1 <-> 4x4 boundary (GFirstBlock.c1 network.py:47, DLastBlock.c2 :163) are dense
// layers in disguise: every output pixel of the pad-3 conv over a 1x1 input sees exactly ONE tap, and the valid
// 4x4 conv of a 4x4 input is one K = 16*Cin dot product.  Run through the generic halo kernel they execute 16x
// the useful MFMAs (pad-3 case) or leave most CUs idle; these skinny-GEMM kernels stream the 16*Cout*Cin
// weights once, straight from global memory into MFMA operands (no LDS: nothing is reused inside a workgroup).
__device__ __forceinline__ void k4_epilogue(const ConvP& p, const f32x4& acc, size_t off, int cb)
{
    float4 o = make_float4(acc[0] * p.scale, acc[1] * p.scale, acc[2] * p.scale, acc[3] * p.scale);
    if (p.mask) {
        const float4 mk = *reinterpret_cast<const float4*>(p.mask + off);
        o.x *= mk.x > 0.f ? 1.f : p.mask_slope; o.y *= mk.y > 0.f ? 1.f : p.mask_slope;
        o.z *= mk.z > 0.f ? 1.f : p.mask_slope; o.w *= mk.w > 0.f ? 1.f : p.mask_slope;
    } else {
        if (p.bias) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + cb); o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w; }
        o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
        o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
    }
    *reinterpret_cast<float4*>(p.y + off) = o;
}

// Both layers stream 16*Cout*Cin weights (16.8 MB at 512 channels) against a handful of samples: what bounds them is how many
// bytes are in flight, not MFMA or bandwidth.  With one wave per 16-cout weight row block walking its Cin (or its pixel's Cin)
// 16 channels at a time, 512 waves kept ~2 MB in flight: 24 / 36 us per launch (1.2 TFLOP/s).  Now the waves of a workgroup
// split Cin, every wave issues all the loads of an 8-step group before the first MFMA, and the partial sums meet in LDS.
constexpr int K4_DEPTH = 8;                     // 16-channel steps whose loads are issued together

template <int NT>
__device__ __forceinline__ void k4_dot(const float* wrow, const float* const (&xrow)[NT], const bool (&ok)[NT], int cbeg, int cend, f32x4 (&acc)[NT])
{
    for (int c0 = cbeg; c0 < cend; c0 += 16 * K4_DEPTH) {
        float4 a[K4_DEPTH], b[NT][K4_DEPTH];
#pragma unroll
        for (int i = 0; i < K4_DEPTH; ++i) {
            const int c = c0 + 16 * i;
            const bool in = c < cend;
            a[i] = in ? *reinterpret_cast<const float4*>(wrow + c) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < NT; ++t)
                b[t][i] = (in && ok[t]) ? *reinterpret_cast<const float4*>(xrow[t] + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < K4_DEPTH; ++i)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = MFMA16(a[i].x, b[t][i].x, acc[t]); acc[t] = MFMA16(a[i].y, b[t][i].y, acc[t]);
                acc[t] = MFMA16(a[i].z, b[t][i].z, acc[t]); acc[t] = MFMA16(a[i].w, b[t][i].w, acc[t]);
            }
    }
}

// 1x1 -> 4x4 (KS 4, pad 3):  y[n][pix][co] = epi(scale * sum_ci w[15-pix][co][ci] * x[n][ci]).
// One workgroup per (pixel, 16 couts): its four waves take a quarter of Cin each; NT tiles of 16 samples share the weight fragment.
template <int NT>
__global__ __launch_bounds__(256) void conv_k4_expand_kernel(ConvP p)
{
    __shared__ float red[4 * NT * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int tiles_co = p.Cout >> 4;
    const int pix = blockIdx.x / tiles_co, co0 = (blockIdx.x - pix * tiles_co) << 4;
    const int cper = (((p.Cin >> 4) + 3) >> 2) << 4;             // channels per wave, whole 16-channel steps
    const int cbeg = wave * cper, cend = min(p.Cin, cbeg + cper);
    const float* wrow = p.w + ((size_t)(15 - pix) * p.Cout + co0 + li) * p.Cin + 4 * kk;
    for (int nb = blockIdx.y * 16 * NT; nb < p.N; nb += gridDim.y * 16 * NT) {
        f32x4 acc[NT];
        const float* xrow[NT];
        bool ok[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int n = nb + 16 * t + li;
            ok[t] = n < p.N;
            xrow[t] = p.x + (size_t)(ok[t] ? n : 0) * p.Cin + 4 * kk;
        }
        k4_dot<NT>(wrow, xrow, ok, cbeg, cend, acc);
#pragma unroll
        for (int t = 0; t < NT; ++t)
            *reinterpret_cast<float4*>(red + ((wave * NT + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
        __syncthreads();
        if (wave < NT) {                              // wave t finishes tile t (fixed summation order)
            const int t = wave;
            f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(red + ((q * NT + t) * 64 + lane) * 4);
                sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
            }
            const int n = nb + 16 * t + li;
            if (n < p.N) k4_epilogue(p, sum, ((size_t)n * 16 + pix) * p.Cout + co0 + 4 * kk, co0 + 4 * kk);
        }
        __syncthreads();
    }
}

// 4x4 -> 1x1 (KS 4, pad 0):  y[n][co] = epi(scale * sum_pix sum_ci w[pix][co][ci] * x[n][pix][ci]).
// One 16-wave workgroup per (16 couts, NT*16 samples): wave = input pixel, partial sums reduced through LDS.
template <int NT>
__global__ __launch_bounds__(1024) void conv_k4_reduce_kernel(ConvP p)
{
    __shared__ float red[16 * NT * 256];
    const int lane = threadIdx.x & 63, pix = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int co0 = blockIdx.x << 4;
    const int nb = blockIdx.y * 16 * NT;
    const float* wrow = p.w + ((size_t)pix * p.Cout + co0 + li) * p.Cin + 4 * kk;
    f32x4 acc[NT];
    const float* xrow[NT];
    bool ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = nb + 16 * t + li;
        ok[t] = n < p.N;
        xrow[t] = p.x + ((size_t)(ok[t] ? n : 0) * 16 + pix) * p.Cin + 4 * kk;
    }
    k4_dot<NT>(wrow, xrow, ok, 0, p.Cin, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t)
        *reinterpret_cast<float4*>(red + ((pix * NT + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    if (pix < NT) {                                   // wave t finishes tile t (fixed summation order)
        const int t = pix;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < 16; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(red + ((q * NT + t) * 64 + lane) * 4);
            sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
        }
        const int n = nb + 16 * t + li;
        if (n < p.N) k4_epilogue(p, sum, (size_t)n * p.Cout + co0 + 4 * kk, co0 + 4 * kk);
    }
}

// The same layer with the 16 input pixels on 16 workgroups (512 instead of 32 workgroups at 512 couts), the four waves of each on a
// quarter of Cin: partial sums through the stream's scratch (pg_set_workspace), the workgroup that takes the last of a cout block's 16
// tickets adds them in pixel order and runs the epilogue.  Agent-scope (sc1) accesses instead of fences: see conv_wino2_kernel.
template <int NT>
__global__ __launch_bounds__(256) void conv_k4_reduce_split_kernel(ConvP p, float* part, unsigned* count)
{
    __shared__ float red[4 * NT * 256];
    __shared__ unsigned ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int pix = blockIdx.x & 15, cot = blockIdx.x >> 4;
    const int co0 = cot << 4, nb = blockIdx.y * 16 * NT;
    const int blk = blockIdx.y * (p.Cout >> 4) + cot;
    const int cper = (((p.Cin >> 4) + 3) >> 2) << 4;
    const int cbeg = wave * cper, cend = min(p.Cin, cbeg + cper);
    const float* wrow = p.w + ((size_t)pix * p.Cout + co0 + li) * p.Cin + 4 * kk;
    f32x4 acc[NT];
    const float* xrow[NT];
    bool ok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int n = nb + 16 * t + li;
        ok[t] = n < p.N;
        xrow[t] = p.x + ((size_t)(ok[t] ? n : 0) * 16 + pix) * p.Cin + 4 * kk;
    }
    k4_dot<NT>(wrow, xrow, ok, cbeg, cend, acc);
#pragma unroll
    for (int t = 0; t < NT; ++t)
        *reinterpret_cast<float4*>(red + ((wave * NT + t) * 64 + lane) * 4) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
    __syncthreads();
    constexpr int SC1 = 16;
    const __amdgpu_buffer_rsrc_t rp = pg_make_rsrc(part + (size_t)blk * 16 * NT * 256, 16u * NT * 1024u);
    if (wave < NT) {
        const int t = wave;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(red + ((q * NT + t) * 64 + lane) * 4);
            sum[0] += v.x; sum[1] += v.y; sum[2] += v.z; sum[3] += v.w;
        }
        __builtin_amdgcn_raw_buffer_store_b128(pg_u32x4{__float_as_uint(sum[0]), __float_as_uint(sum[1]), __float_as_uint(sum[2]), __float_as_uint(sum[3])},
                                               rp, ((pix * NT + t) * 64 + lane) * 16, 0, SC1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(count + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (ticket != 15u) return;
    if (threadIdx.x == 0) __hip_atomic_store(count + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wave < NT) {
        const int t = wave;
        f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const pg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, ((q * NT + t) * 64 + lane) * 16, 0, SC1);
            sum[0] += __uint_as_float(v[0]); sum[1] += __uint_as_float(v[1]); sum[2] += __uint_as_float(v[2]); sum[3] += __uint_as_float(v[3]);
        }
        const int n = nb + 16 * t + li;
        if (n < p.N) k4_epilogue(p, sum, (size_t)n * p.Cout + co0 + 4 * kk, co0 + 4 * kk);
    }
}

// Weight gradient of both layers: per tap an outer-product GEMM with K = N (the minibatch):
//   EXPAND: dW[tap][co][ci] += scale * sum_n gz[n][15-tap][co] * x[n][ci]        (x: [N][Cin], gz: [N][16][Cout])
//   else  : dW[tap][co][ci] += scale * sum_n gz[n][co]         * x[n][tap][ci]   (x: [N][16][Cin], gz: [N][Cout])
// One wave per (tap, 16 couts, 64 cins): HBM-bound on the 16*Cout*Cin read-modify-write of dW.
template <bool EXPAND>
__global__ __launch_bounds__(256) void conv_k4_wgrad_kernel(WgP p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int tiles_co = p.Cout >> 4, groups_ci = (p.Cin + 63) >> 6;
    int wid = blockIdx.x * 4 + wave;
    const int cig = wid % groups_ci; wid /= groups_ci;
    const int cot = wid % tiles_co; const int tap = wid / tiles_co;
    if (tap >= 16) return;
    const int co0 = cot << 4, ci0 = cig << 6;
    const int gstride = EXPAND ? 16 * p.Cout : p.Cout, goff = EXPAND ? (15 - tap) * p.Cout : 0;
    const int xstride = EXPAND ? p.Cin : 16 * p.Cin, xoff = EXPAND ? 0 : tap * p.Cin;
    const bool do_bias = p.db != nullptr && cig == 0 && (EXPAND || tap == 0);
    f32x4 acc[4], accb = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int n0 = 0; n0 < p.N; n0 += 4) {
        const int n = n0 + kk;
        const bool okn = n < p.N;
        const float a = okn ? p.gz[(size_t)n * gstride + goff + co0 + li] : 0.f;
        float b[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            b[t] = (okn && ci0 + 16 * t + li < p.Cin) ? p.x[(size_t)n * xstride + xoff + ci0 + 16 * t + li] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MFMA16(a, b[t], acc[t]);
        if (do_bias) accb = MFMA16(a, 1.0f, accb);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ci = ci0 + 16 * t + li;
        if (ci >= p.Cin) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = p.dw + ((size_t)(tap * p.Cout + co0 + 4 * kk + r) * p.Cin + ci);
            *dst += acc[t][r] * p.scale;                       // this wave is the only writer of the element
        }
    }
    if (do_bias && li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(p.db + co0 + 4 * kk + r, accb[r]);
    }
}

// ------------------------------------------------------------------------------------------------------
// 3x3 layers with 8 or 16 output channels and <= 32 input channels (the 512^2 / 1024^2 stages and their
// backward-data convs).  A 16x16x4 tile wastes half of its rows on 8 couts and, more importantly, these layers have
// no K loop to pipeline; v_mfma_f32_4x4x1_16B_f32 with block = (cout quad, pixel quad) covers COUT couts x
// 64*4/COUT consecutive pixels of a row per instruction for one (tap, cin), every lane useful.
// Workgroup: TH rows x 32 pixels of one image, whole-K halo tile in LDS (row stride CIN+4 floats: conflict-free
// b128), one barrier, wave w owns TH/4 rows; weights are re-read from LDS per (tap, cin quad) as one b128.
// Optional fused 2x2 average pool of the activated output (see pg_conv2d_pool_nhwc).
template <int COUT, int CIN, int TH>
__global__ __launch_bounds__(256) void conv_thin_kernel(ConvP p)
{
    constexpr int S = CIN + 4, WT = 34, HT = TH + 2, C4 = CIN / 4;
    constexpr int QO = COUT / 4, QP = 16 / QO, PXG = 4 * QP;            // pixels per MFMA group: 32 (8 couts) / 16
    constexpr int GPR = 32 / PXG, G = (TH / 4) * GPR;                   // groups per row, groups per wave
    extern __shared__ __align__(16) float lds[];
    float* xt = lds;                             // [HT][WT][S]
    float* wl = lds + HT * WT * S;               // [9][COUT][CIN]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, j = lane & 3, qo = blk % QO, qp = blk / QO;
    int b = (int)pg_xcd_remap(blockIdx.x, gridDim.x);         // contiguous tile ranges per XCD: halos come from its L2
    const int tw_i = b % (p.Wout >> 5); b /= (p.Wout >> 5);
    const int th_i = b % (p.Hout / TH); const int n = b / (p.Hout / TH);
    const int oh0 = th_i * TH, ow0 = tw_i << 5;
    const int xH = p.ups ? (p.Hin >> 1) : p.Hin, xW = p.ups ? (p.Win >> 1) : p.Win;

    // one image through a raw buffer (bufload.h): out-of-image halo pixels are zero-filled by the hardware, no branch per load
    const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(p.x + (size_t)n * xH * xW * CIN, (unsigned)((size_t)xH * xW * CIN * 4));
    constexpr int NLD = (HT * WT * C4 + 255) / 256;
    const unsigned char* gbase = p.gbytes ? p.gbytes + (size_t)n * p.Hin * p.Win * C4 : nullptr;     // this image's sign bytes
    float4 xv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        const int c4 = e % C4; const int q = e / C4;
        const int tw = q % WT, th = q / WT;
        int ih = oh0 + th - 1, iw = ow0 + tw - 1;
        const bool ok = e < HT * WT * C4 && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        unsigned char gb = 0;
        if (p.gbytes && ok) gb = gbase[(ih * p.Win + iw) * C4 + c4];
        if (p.ups) { ih >>= 1; iw >>= 1; }
        xv[i] = pg_buf_load4(rx, ok ? 4u * (unsigned)((ih * xW + iw) * CIN + 4 * c4) : PG_OOB, 0);
        if (p.gbytes) {                              // pool adjoint in the gather: x 1/4 (x mul) x LeakyReLU' of the finer activation
            const float4 f = pg_sign_factors(gb, p.gslope);
            xv[i].x *= f.x * p.gmul; xv[i].y *= f.y * p.gmul; xv[i].z *= f.z * p.gmul; xv[i].w *= f.w * p.gmul;
        }
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = tid + 256 * i;
        if (e < HT * WT * C4) *reinterpret_cast<float4*>(xt + (e / C4) * S + 4 * (e % C4)) = xv[i];
    }
    for (int e = tid; e < 9 * COUT * C4; e += 256)
        *reinterpret_cast<float4*>(wl + 4 * e) = *reinterpret_cast<const float4*>(p.w + 4 * e);
    __syncthreads();

    f32x4 acc[G], acc2[G];
    int xoff[G];                                 // LDS offset of this lane's pixel in group g (tap 0,0)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int gg = wave * G + g;
        xoff[g] = ((gg / GPR) * WT + (gg % GPR) * PXG + 4 * qp + j) * S;
    }
    const float* wrow = wl + (4 * qo + j) * CIN; // A operand: couts 4*qo + (lane&3)
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const int toff = ((tp / 3) * WT + (tp % 3)) * S;
#pragma unroll
        for (int c4 = 0; c4 < C4; ++c4) {
            const float4 a = *reinterpret_cast<const float4*>(wrow + tp * COUT * CIN + 4 * c4);
            float4 bq[G];
#pragma unroll
            for (int g = 0; g < G; ++g) bq[g] = *reinterpret_cast<const float4*>(xt + xoff[g] + toff + 4 * c4);
            // two accumulation chains per group (even / odd channel of the quad) and the groups interleaved: consecutive MFMAs
            // never share an accumulator (a dependent v_mfma_f32_4x4x1 cannot issue back to back)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, bq[g].x, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc2[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, bq[g].y, acc2[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, bq[g].z, acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc2[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, bq[g].w, acc2[g], 0, 0, 0);
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] += acc2[g];
    // D register r of this lane = out[pixel][cout 4*qo + r]
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + 4 * qo);
    float4 ov[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int gg = wave * G + g;
        const int oy = oh0 + gg / GPR, ox = ow0 + (gg % GPR) * PXG + 4 * qp + j;
        const size_t off = (((size_t)n * p.Hout + oy) * p.Wout + ox) * COUT + 4 * qo;
        float4 o = make_float4(acc[g][0] * p.scale, acc[g][1] * p.scale, acc[g][2] * p.scale, acc[g][3] * p.scale);
        if (p.mask) {
            float4 f;
            if (p.mask_bytes) f = pg_sign_factors(reinterpret_cast<const unsigned char*>(p.mask)[off >> 2], p.mask_slope);
            else {
                const float4 mk = *reinterpret_cast<const float4*>(p.mask + off);
                f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
            }
            o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
        } else {
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
            o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
            o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
            if (p.ysigns) p.ysigns[off >> 2] = pg_sign_byte(o);
        }
        if (p.pnb_y) {                               // adjoint of the previous layer's (LeakyReLU -> PixelNorm), see ConvP
            const float4 yv = *reinterpret_cast<const float4*>(p.pnb_y + off);
            const float4 gv = make_float4(acc[g][0] * p.scale, acc[g][1] * p.scale, acc[g][2] * p.scale, acc[g][3] * p.scale);
            float dt = (gv.x * yv.x + gv.y * yv.y) + (gv.z * yv.z + gv.w * yv.w);
            dt += __shfl_xor(dt, 4, 64);
            if (QO >= 4) dt += __shfl_xor(dt, 8, 64);
            const float rr = p.pnb_r[((size_t)n * p.Hout + oy) * p.Wout + ox], mean = dt / (float)COUT;
            o.x = rr * (gv.x - yv.x * mean) * (yv.x > 0.f ? 1.f : p.mask_slope);
            o.y = rr * (gv.y - yv.y * mean) * (yv.y > 0.f ? 1.f : p.mask_slope);
            o.z = rr * (gv.z - yv.z * mean) * (yv.z > 0.f ? 1.f : p.mask_slope);
            o.w = rr * (gv.w - yv.w * mean) * (yv.w > 0.f ? 1.f : p.mask_slope);
        }
        if (p.pn_r) {                                // PixelNorm over the COUT channels of the pixel: QO lanes (4 apart) share it
            float ssq = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
            ssq += __shfl_xor(ssq, 4, 64);
            if (QO >= 4) ssq += __shfl_xor(ssq, 8, 64);
            const float rr = rsqrtf(ssq / (float)COUT + p.pn_eps);
            o.x *= rr; o.y *= rr; o.z *= rr; o.w *= rr;
            if (qo == 0) p.pn_r[((size_t)n * p.Hout + oy) * p.Wout + ox] = rr;
        }
        if (p.y_bytes) reinterpret_cast<unsigned char*>(p.y)[off >> 2] = pg_sign_byte(o);
        else if (!(p.ypool && p.pool_only)) *reinterpret_cast<float4*>(p.y + off) = o;
        ov[g] = o;
    }
    if (p.ypool) {                               // 2x2 mean: column partner = lane^1, row partner = group g + GPR (same wave)
        static_assert(TH % 8 == 0, "a wave must own complete row pairs");
#pragma unroll
        for (int g = 0; g < G; ++g) {
            ov[g].x += __shfl_xor(ov[g].x, 1, 64); ov[g].y += __shfl_xor(ov[g].y, 1, 64);
            ov[g].z += __shfl_xor(ov[g].z, 1, 64); ov[g].w += __shfl_xor(ov[g].w, 1, 64);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (((g / GPR) & 1) != 0) continue;                  // compile-time: even rows lead
            const int gg = wave * G + g;
            const int oy = oh0 + gg / GPR, ox = ow0 + (gg % GPR) * PXG + 4 * qp + j;
            float4 v = make_float4(((ov[g].x + ov[g + GPR].x)) * 0.25f, ((ov[g].y + ov[g + GPR].y)) * 0.25f,
                                   ((ov[g].z + ov[g + GPR].z)) * 0.25f, ((ov[g].w + ov[g + GPR].w)) * 0.25f);
            if (j & 1) continue;
            const size_t poff = (((size_t)n * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1)) * COUT + 4 * qo;
            if (p.pool_other) {
                const float4 q = *reinterpret_cast<const float4*>(p.pool_other + poff);
                v.x = fmaf(v.x, p.pool_a, p.pool_b * q.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * q.y);
                v.z = fmaf(v.z, p.pool_a, p.pool_b * q.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * q.w);
            } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
            *reinterpret_cast<float4*>(p.ypool + poff) = v;
        }
    }
}

template <int COUT, int CIN, int TH>
int launch_thin(ConvP& p, hipStream_t s)
{
    const size_t smem = ((size_t)(TH + 2) * 34 * (CIN + 4) + 9 * COUT * CIN) * sizeof(float);
    if ((long long)p.Hin * p.Win * CIN * 4 >= (1ll << 31)) return PG_E_UNSUP;                  // 32-bit buffer offsets per image
    auto kern = conv_thin_kernel<COUT, CIN, TH>;
    if (int rc = set_smem(kern, smem)) return rc;
    dim3 grid((unsigned)(p.N * (p.Hout / TH) * (p.Wout >> 5)));
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_thin_kernel<%d, %d, %d>", COUT, CIN, TH);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    return (int)hipGetLastError();
}

int dispatch_thin(ConvP& p, hipStream_t s)
{
    if (g_tune[3] != 20) {                        // row-streaming kernel (conv_strip.hip) where the shape allows; 20: tile kernel (A/B)
        const int rc = pgk::launch_conv_strip(p, s, g_last_kernel, sizeof(g_last_kernel));
        if (rc != PG_E_UNSUP) return rc;
    }
#define THIN(CO_, CI_) if (p.Cout == CO_ && p.Cin == CI_) return launch_thin<CO_, CI_, 8>(p, s);
    THIN(8, 8) THIN(8, 16) THIN(16, 8)
#undef THIN
    return PG_E_UNSUP;
}

inline bool k4_dense_ok(int Cin, int Cout) { return (Cin & 15) == 0 && (Cout & 15) == 0; }

int launch_k4_conv(ConvP& p, hipStream_t s)
{
    if (p.pad == 3) {                                        // 1x1 -> 4x4
        const int nt = p.N <= 16 ? 1 : (p.N <= 32 ? 2 : 4);
        int gy = (p.N + 16 * nt - 1) / (16 * nt); if (gy > 8) gy = 8;
        dim3 grid(16 * (p.Cout >> 4), gy);
        snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_k4_expand_kernel<%d>", nt);
        if (nt == 1) hipLaunchKernelGGL(conv_k4_expand_kernel<1>, grid, dim3(256), 0, s, p);
        else if (nt == 2) hipLaunchKernelGGL(conv_k4_expand_kernel<2>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(conv_k4_expand_kernel<4>, grid, dim3(256), 0, s, p);
    } else {                                                 // 4x4 -> 1x1
        const int nt = p.N <= 16 ? 1 : 2;
        dim3 grid(p.Cout >> 4, (p.N + 16 * nt - 1) / (16 * nt));
        pgk::Workspace ws{};
        const size_t nblk = (size_t)grid.x * grid.y;
        if (g_tune[3] != 21 && nblk <= 256 && nblk <= pgk::WS_TICKETS && pgk::find_workspace(s, ws) &&
            pgk::WS_HEAD + nblk * 16 * nt * 1024 <= ws.bytes) {          // few cout blocks: one workgroup per (block, input pixel)
            dim3 sgrid(grid.x * 16, grid.y);
            snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_k4_reduce_split_kernel<%d>", nt);
            float* part = reinterpret_cast<float*>(ws.ptr + pgk::WS_HEAD);
            unsigned* count = reinterpret_cast<unsigned*>(ws.ptr);
            if (nt == 1) hipLaunchKernelGGL(conv_k4_reduce_split_kernel<1>, sgrid, dim3(256), 0, s, p, part, count);
            else hipLaunchKernelGGL(conv_k4_reduce_split_kernel<2>, sgrid, dim3(256), 0, s, p, part, count);
            return (int)hipGetLastError();
        }
        snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_k4_reduce_kernel<%d>", nt);
        if (nt == 1) hipLaunchKernelGGL(conv_k4_reduce_kernel<1>, grid, dim3(1024), 0, s, p);
        else hipLaunchKernelGGL(conv_k4_reduce_kernel<2>, grid, dim3(1024), 0, s, p);
    }
    return (int)hipGetLastError();
}

int launch_k4_wgrad(WgP& p, hipStream_t s)
{
    const int waves = 16 * (p.Cout >> 4) * ((p.Cin + 63) >> 6);
    dim3 grid((waves + 3) / 4);
    if (p.pad == 3) {
        snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_k4_wgrad_kernel<true>");
        hipLaunchKernelGGL(conv_k4_wgrad_kernel<true>, grid, dim3(256), 0, s, p);
    } else {
        snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_k4_wgrad_kernel<false>");
        hipLaunchKernelGGL(conv_k4_wgrad_kernel<false>, grid, dim3(256), 0, s, p);
    }
    return (int)hipGetLastError();
}

// wt[KS-1-kh][KS-1-kw][ci][co] = w[kh][kw][co][ci]
__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wt, int KS, int Cout, int Cin)
{
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int kh = tap / KS, kw = tap % KS;
    const int otap = (KS - 1 - kh) * KS + (KS - 1 - kw);
    const int ci_b = blockIdx.x * 32, co_b = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int co = co_b + r, ci = ci_b + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)tap * Cout + co) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci_b + r, co = co_b + tx;
        if (ci < Cin && co < Cout) wt[((size_t)otap * Cin + ci) * Cout + co] = tile[tx][r];
    }
}

// All layers of a network in ONE launch (the weights live in one flat buffer, the packed copies in its mirror).
constexpr int PACK_MAX_LAYERS = 32;
struct PackDesc {
    int n;
    int first_block[PACK_MAX_LAYERS + 1];          // prefix sum of the per-layer block counts
    long long off[PACK_MAX_LAYERS];                // element offset of the layer in both flat buffers
    int ks[PACK_MAX_LAYERS], cout[PACK_MAX_LAYERS], cin[PACK_MAX_LAYERS];
};

__global__ void pack_dgrad_batched_kernel(const float* __restrict__ wbase, float* __restrict__ wtbase, PackDesc d)
{
    __shared__ float tile[32][33];
    int l = 0;
    while (l + 1 < d.n && (int)blockIdx.x >= d.first_block[l + 1]) ++l;
    const int KS = d.ks[l], Cout = d.cout[l], Cin = d.cin[l];
    const float* w = wbase + d.off[l];
    float* wt = wtbase + d.off[l];
    int b = blockIdx.x - d.first_block[l];
    const int nbx = (Cin + 31) / 32, nby = (Cout + 31) / 32;
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby; const int tap = b / nby;
    const int kh = tap / KS, kw = tap % KS;
    const int otap = (KS - 1 - kh) * KS + (KS - 1 - kw);
    const int ci_b = bx * 32, co_b = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int co = co_b + r, ci = ci_b + tx;
        tile[r][tx] = (co < Cout && ci < Cin) ? w[((size_t)tap * Cout + co) * Cin + ci] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int ci = ci_b + r, co = co_b + tx;
        if (ci < Cin && co < Cout) wt[((size_t)otap * Cin + ci) * Cout + co] = tile[tx][r];
    }
}

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

struct TileGeom { int lgTW, lgTH, TN, tilesW, tilesH, ntiles; };

inline TileGeom make_geom(int N, int Hout, int Wout, int BPX, int max_tw = 32)
{
    TileGeom g;
    int TW = Wout < max_tw ? Wout : max_tw; if (TW > BPX) TW = BPX;
    while (TW > 4 && BPX / TW < 4 && Hout >= 4) TW >>= 1;      // keep tiles at least 4 rows tall (halo <= 2.25x)
    int TH = BPX / TW; if (TH > Hout) TH = Hout;
    g.lgTW = ilog2(TW); g.lgTH = ilog2(TH);
    g.TN = BPX / (TW * TH);
    g.tilesW = Wout / TW; g.tilesH = Hout / TH;
    g.ntiles = ((N + g.TN - 1) / g.TN) * g.tilesH * g.tilesW;
    return g;
}


template <int KS, int VEC, int WAVES_CO, int WM, int WN>
int launch_conv(ConvP& p, hipStream_t s)
{
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int BCO = 16 * WM * WAVES_CO, BPX = 16 * WN * WAVES_PX, KCP = RowStride<VEC>::value;
    // fused pooling needs the vertical 2x2 partner inside the wave: <= 8-wide tiles always work (lane ^ TW),
    // 16-wide ones need two, 32-wide ones four 16-pixel groups per wave
    const int max_tw = p.ypool ? (WN >= 4 ? 32 : (WN >= 2 ? 16 : 8)) : 32;
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX, max_tw);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH;
    const int HT = (1 << g.lgTH) + KS - 1, WT = (1 << g.lgTW) + KS - 1;
    constexpr int XMAX = halo_max(KS, BPX);
    if (g.TN * HT * WT > XMAX) return PG_E_UNSUP;       // halo larger than the register-prefetch budget
    if ((long long)g.TN * (p.ups ? p.Hin >> 1 : p.Hin) * (p.ups ? p.Win >> 1 : p.Win) * p.Cin * 4 >= (1ll << 31)) return PG_E_UNSUP;   // 32-bit buffer offsets
    p.mWT = (unsigned)((1ull << 32) / (unsigned)WT) + 1u; p.mHT = (unsigned)((1ull << 32) / (unsigned)HT) + 1u;
    const size_t smem = (size_t)(KS * KS * BCO + g.TN * HT * WT) * KCP * sizeof(float);
    auto kern = conv_igemm_kernel<KS, VEC, WAVES_CO, WM, WN>;
    if (int rc = set_smem(kern, smem)) return rc;
    const int ncob = (p.Cout + BCO - 1) / BCO;
    const int nblocks = g.ntiles * ncob;
    const int nchunks = p.Cin / (4 * VEC);
    int ksplit = 1;
    if (g_tune[2] > 0) {
        ksplit = g_tune[2] > nchunks ? nchunks : g_tune[2];
        const int cper = (nchunks + ksplit - 1) / ksplit;
        ksplit = (nchunks + cper - 1) / cper;
    } else if (nblocks < 192 && nchunks >= 4) {           // too few workgroups for 256 CUs: slice K
        ksplit = (512 + nblocks - 1) / nblocks;
        if (ksplit > nchunks) ksplit = nchunks;
        const int cper = (nchunks + ksplit - 1) / ksplit;
        ksplit = (nchunks + cper - 1) / cper;
    }
    if (ksplit > 1 && (p.yup || p.pn_r || p.pnb_y || p.mask_bytes || p.y_bytes || p.ysigns)) return PG_E_UNSUP;   // these epilogues need complete sums
    if ((p.pn_r || p.pnb_y) && (WAVES_CO != 1 || ncob != 1)) return PG_E_UNSUP;  // ... and every cout of a pixel inside one wave
    p.ksplit = ksplit;
    const size_t npix = (size_t)p.N * p.Hout * p.Wout;
    if (ksplit > 1) {
        hipError_t e = hipMemsetAsync(p.y, 0, npix * p.Cout * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(g.ntiles, ncob, ksplit);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_igemm_kernel<%d, %d, %d, %d, %d>", KS, VEC, WAVES_CO, WM, WN);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    if (ksplit > 1) {
        const size_t total = npix * (p.Cout >> 2);
        int eg = (int)((total + 255) / 256); if (eg > 2048) eg = 2048;
        const bool identity = !p.mask && !p.bias && p.slope == 1.0f;
        if (!identity)
            hipLaunchKernelGGL(conv_epilogue_kernel, dim3(eg), dim3(256), 0, s, p.y, p.bias, p.mask, npix, p.Cout,
                               p.slope, p.mask_slope);
    }
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Small-M layers (4x4 .. 16x16 at minibatch 3: a few hundred output pixels against K = 9*512).  The generic kernel
// has to slice K across workgroups there (memset + fp32 atomics + a deferred epilogue launch).  Here the FOUR WAVES of a
// workgroup split K instead: all waves own the same 16-cout x (16*WN)-pixel tile, wave w takes channels
// [64c + 16w, 64c + 16w + 16) of every 64-channel super-chunk (its own slice of the LDS rows), the four partial
// accumulators are summed through LDS and wave 0 runs the fused epilogue -- one launch, no atomics.
template <int WN>
__global__ __launch_bounds__(256) void conv_ksplit_kernel(ConvP p)
{
    // LDS row = 4 wave slices + 4 floats: with a row stride of 96 floats the 16 lanes (li) of a ds_read_b128 group sat on TWO
    // 4-bank groups (96 li mod 64 = 0 / 32: eight-way conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.65); 100 li mod 64
    // walks through all sixteen 4-bank groups
    constexpr int KS = 3, TAPS = 9, BCO = 16, BPX = 16 * WN, KCP = 24, RS = 4 * KCP + 4;
    constexpr int WEL = TAPS * BCO * 16;                        // float4 per weight super-chunk
    constexpr int WPT = (WEL + 255) / 256;
    constexpr int XMAX = BPX <= 16 ? 36 : (BPX * 9) / 4;
    constexpr int XPT = (XMAX * 16 + 255) / 256;
    extern __shared__ __align__(16) float lds[];
    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
    const int HT = TH + KS - 1, WT = TW + KS - 1;
    float* wt = lds;                          // [TAPS][BCO][4][KCP]
    float* xt = lds + TAPS * BCO * RS;        // [TN*HT*WT][4][KCP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    int t = blockIdx.x;
    const int tw_i = t % p.tilesW; t /= p.tilesW;
    const int th_i = t % p.tilesH; t /= p.tilesH;
    const int n0 = t * p.TN;
    const int oh0 = th_i << p.lgTH, ow0 = tw_i << p.lgTW;
    const int co0 = blockIdx.y * BCO;

    int pixbase[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        const int j = n * 16 + li;
        const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
        pixbase[n] = ((tn * HT + th) * WT + tw) * RS + wave * KCP + 4 * kk;
    }
    const int wbase = li * RS + wave * KCP + 4 * kk;

    const int npix = p.TN * HT * WT;
    int wsrc[WPT], wdst[WPT], xsrc[XPT], xdst[XPT], wch[WPT], xch[XPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx >> 4, v = idx & 15;                   // row (tap, cout), float4 index inside the 64 channels
        const int tap = r / BCO, col = r - tap * BCO, co = co0 + col;
        wdst[i] = idx < WEL ? r * RS + (v >> 2) * KCP + 4 * (v & 3) : -1;
        wsrc[i] = (idx < WEL && co < p.Cout) ? (tap * p.Cout + co) * p.Cin : -1;        // element offsets; -1 = zero fill
        wch[i] = 4 * v;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx >> 4, v = idx & 15;
        const int r2 = (int)__umulhi((unsigned)q, p.mWT), tw = q - r2 * WT;
        const int tn = (int)__umulhi((unsigned)r2, p.mHT), th = r2 - tn * HT;
        const int n = n0 + tn;
        const int ih = oh0 + th - p.pad, iw = ow0 + tw - p.pad;
        const bool in_tile = q < npix;
        const bool ok = in_tile && n < p.N && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        xdst[i] = in_tile ? q * RS + (v >> 2) * KCP + 4 * (v & 3) : -1;
        xsrc[i] = ok ? ((n * p.Hin + ih) * p.Win + iw) * p.Cin : -1;
        xch[i] = 4 * v;
    }
    int tapoff[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) tapoff[tp] = ((tp / KS) * WT + (tp % KS)) * RS;

    f32x4 acc[WN];
#pragma unroll
    for (int n = 0; n < WN; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nsuper = (p.Cin + 63) >> 6;
    const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(p.x, (unsigned)((size_t)p.N * p.Hin * p.Win * p.Cin * 4));    // small-M layers: a few MB
    const __amdgpu_buffer_rsrc_t rw = pg_make_rsrc(p.w, (unsigned)((size_t)TAPS * p.Cout * p.Cin * 4));
    float4 wreg[WPT], xreg[XPT];
    auto fetch = [&](int sc) {
        const int k0 = sc << 6;
        // raw buffer loads (bufload.h): a select on the offset instead of a branch around every load
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            wreg[i] = pg_buf_load4(rw, (wsrc[i] >= 0 && k0 + wch[i] < p.Cin) ? 4u * (unsigned)(wsrc[i] + k0 + wch[i]) : PG_OOB, 0);
#pragma unroll
        for (int i = 0; i < XPT; ++i)
            xreg[i] = pg_buf_load4(rx, (xsrc[i] >= 0 && k0 + xch[i] < p.Cin) ? 4u * (unsigned)(xsrc[i] + k0 + xch[i]) : PG_OOB, 0);
    };
    fetch(0);
    for (int sc = 0; sc < nsuper; ++sc) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) if (wdst[i] >= 0) *reinterpret_cast<float4*>(wt + wdst[i]) = wreg[i];
#pragma unroll
        for (int i = 0; i < XPT; ++i) if (xdst[i] >= 0) *reinterpret_cast<float4*>(xt + xdst[i]) = xreg[i];
        __syncthreads();
        if (sc + 1 < nsuper) fetch(sc + 1);
        float a[2][4], b[2][WN][4];
        lds_load<4>(wt + wbase, a[0]);
#pragma unroll
        for (int n = 0; n < WN; ++n) lds_load<4>(xt + pixbase[n] + tapoff[0], b[0][n]);
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
            const int cur = tp & 1, nxt = cur ^ 1;
            if (tp + 1 < TAPS) {
                lds_load<4>(wt + (tp + 1) * BCO * RS + wbase, a[nxt]);
#pragma unroll
                for (int n = 0; n < WN; ++n) lds_load<4>(xt + pixbase[n] + tapoff[tp + 1], b[nxt][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                if constexpr (WN == 1) {                      // two accumulation chains: no adjacent dependent MFMAs
                    if (s4 & 1) acc_odd = MFMA16(a[cur][s4], b[cur][0][s4], acc_odd);
                    else acc[0] = MFMA16(a[cur][s4], b[cur][0][s4], acc[0]);
                } else {
#pragma unroll
                    for (int n = 0; n < WN; ++n) acc[n] = MFMA16(a[cur][s4], b[cur][n][s4], acc[n]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if constexpr (WN == 1) acc[0] += acc_odd;
    // ---- sum the four K slices through LDS; wave 0 finishes
    float* red = lds;                                           // [3][WN][64][4]
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < WN; ++n)
            *reinterpret_cast<float4*>(red + (((wave - 1) * WN + n) * 64 + lane) * 4) = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
    }
    __syncthreads();
    if (wave != 0) return;
    const int cb = co0 + 4 * kk;
    if (cb >= p.Cout) return;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + cb);
#pragma unroll
    for (int n = 0; n < WN; ++n) {
        float4 sacc = make_float4(acc[n][0], acc[n][1], acc[n][2], acc[n][3]);
#pragma unroll
        for (int w = 0; w < 3; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(red + ((w * WN + n) * 64 + lane) * 4);
            sacc.x += v.x; sacc.y += v.y; sacc.z += v.z; sacc.w += v.w;
        }
        const int j = n * 16 + li;
        const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
        const int ni = n0 + tn;
        if (ni >= p.N) continue;
        const size_t off = (((size_t)ni * p.Hout + oh0 + th) * p.Wout + ow0 + tw) * p.Cout + cb;
        float4 o = make_float4(sacc.x * p.scale, sacc.y * p.scale, sacc.z * p.scale, sacc.w * p.scale);
        if (p.mask) {
            const float4 mk = *reinterpret_cast<const float4*>(p.mask + off);
            o.x *= mk.x > 0.f ? 1.f : p.mask_slope; o.y *= mk.y > 0.f ? 1.f : p.mask_slope;
            o.z *= mk.z > 0.f ? 1.f : p.mask_slope; o.w *= mk.w > 0.f ? 1.f : p.mask_slope;
        } else {
            o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
            o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
            o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
        }
        *reinterpret_cast<float4*>(p.y + off) = o;
    }
}

template <int WN>
int launch_ksplit(ConvP& p, hipStream_t s)
{
    constexpr int BPX = 16 * WN;
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH;
    const int HT = (1 << g.lgTH) + 2, WT = (1 << g.lgTW) + 2;
    constexpr int XMAX = BPX <= 16 ? 36 : (BPX * 9) / 4;
    if (g.TN * HT * WT > XMAX) return PG_E_UNSUP;
    if ((long long)p.N * p.Hin * p.Win * p.Cin * 4 >= (1ll << 31) || (long long)9 * p.Cout * p.Cin * 4 >= (1ll << 31)) return PG_E_UNSUP;   // 32-bit buffer offsets
    p.mWT = (unsigned)((1ull << 32) / (unsigned)WT) + 1u; p.mHT = (unsigned)((1ull << 32) / (unsigned)HT) + 1u;
    size_t smem = (size_t)(9 * 16 + g.TN * HT * WT) * 100 * sizeof(float);      // RS of the kernel
    const size_t red = (size_t)3 * WN * 256 * sizeof(float);
    if (red > smem) smem = red;
    auto kern = conv_ksplit_kernel<WN>;
    if (int rc = set_smem(kern, smem)) return rc;
    p.ksplit = 1;
    dim3 grid(g.ntiles, (p.Cout + 15) / 16);
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_ksplit_kernel<%d>", WN);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    return (int)hipGetLastError();
}

// Tile-shape selection.  A workgroup (4 waves, one per SIMD) runs for  L = chunks x (MFMA cycles per chunk + per-chunk
// overhead) + fixed cycles  when it has the CU to itself; r workgroups are resident per CU (LDS / VGPR limited) and
// share the matrix pipes, so a batch of r workgroups takes max(L, r x MFMA cycles) and the launch takes
// (batches over 256 CUs) x that.  Small tiles have a longer MFMA-free fraction per workgroup but far more residents,
// which is what wins on the 256/512-channel layers at minibatch 3; the constants were fitted to a sweep of every
// candidate over every layer shape of the schedule (tools/sweeps/sweep_conv_all.py, tools/sweeps/fit_cost_model.py: 0.7 % regret).
// Shapes with < 192 workgroups are K-split in launch_conv (memset + atomics + deferred epilogue: fixed penalty).
struct TileCand { int bpx, bco; };

template <int KS, int VEC>
int dispatch_conv(ConvP& p, hipStream_t s)
{
    if constexpr (KS == 4) {
        return launch_conv<KS, VEC, 4, 1, 1>(p, s);               // 16-tap halo: keep the pixel tile small
    } else {
        if constexpr (VEC == 4) {
            const long long Mpx = (long long)p.N * p.Hout * p.Wout;
            if (!p.ups && !p.ypool && !p.yup && !p.pn_r && !p.pnb_y && p.Cin >= 128 && g_tune[0] < 0 && g_tune[3] != 8 &&
                ((g_tune[3] == 9 && Mpx <= 2304) || (g_tune[3] != 9 && Mpx <= 576))) {
                const int rc = Mpx <= 256 ? launch_ksplit<1>(p, s) : launch_ksplit<2>(p, s);
                if (rc != PG_E_UNSUP) return rc;
            }
        }
        static const TileCand cands[] = {{256, 16}, {128, 64}, {128, 32}, {128, 16}, {64, 64}, {64, 32}, {64, 16}, {16, 64}};
        // registers per lane of each instantiation (hipcc 7.2, KS = 3): residency = min(LDS, 512 / vgpr)
        static const int vgpr4[] = {160, 200, 128, 100, 164, 100, 68, 92};
        static const int vgpr2[] = {120, 144, 92, 76, 120, 76, 52, 60};
        static const int vgpr1[] = {88, 112, 72, 52, 88, 56, 32, 48};
        const int* vg = VEC == 4 ? vgpr4 : (VEC == 2 ? vgpr2 : vgpr1);
        constexpr int KCP = RowStride<VEC>::value;
        const int nchunks = p.Cin / (4 * VEC);
        double best = 1e30; int bi = 1;
        for (int i = 0; i < 8; ++i) {
            const TileCand c = cands[i];
            if (c.bpx == 256 && p.Cout > 16) continue;            // 256-pixel tiles only exist for <= 16 couts
            if (c.bco > 16 && p.Cout <= 16) continue;
            if (c.bco > 32 && p.Cout <= 32) continue;
            if ((p.pn_r || p.pnb_y) && (c.bco < p.Cout || i == 1 || i == 4 || i == 5 || i == 7)) continue;   // fused PixelNorm: one wave row of couts
            const TileGeom g = make_geom(p.N, p.Hout, p.Wout, c.bpx);
            const int halo = g.TN * ((1 << g.lgTH) + KS - 1) * ((1 << g.lgTW) + KS - 1);
            const long long lds = (long long)(KS * KS * c.bco + halo) * KCP * 4;
            long long r = 160 * 1024 / lds;
            if (r > 512 / vg[i]) r = 512 / vg[i];
            if (r > 8) r = 8;
            if (r < 1) r = 1;
            const long long blocks = (long long)g.ntiles * ((p.Cout + c.bco - 1) / c.bco);
            long long ks = 1;
            if (blocks < 192 && nchunks >= 4) {                   // same rule as launch_conv
                ks = (512 + blocks - 1) / blocks;
                if (ks > nchunks) ks = nchunks;
                const long long cp = (nchunks + ks - 1) / ks;
                ks = (nchunks + cp - 1) / cp;
            }
            const long long cper = (nchunks + ks - 1) / ks;
            const long long wgs = blocks * ks;
            const double mfma_chunk = (double)(c.bpx / 16) * (c.bco / 16) / 4.0 * VEC * KS * KS * 32.0;
            const double mfma_wg = mfma_chunk * (double)cper;
            const double L = (double)cper * (mfma_chunk + 300.0) + 1500.0 + (ks > 1 ? 5.0 * c.bpx * c.bco + 2000.0 : 0.0);
            const long long full = wgs / (256 * r), rem = wgs % (256 * r);
            double cost = (double)full * (L > r * mfma_wg ? L : r * mfma_wg);
            if (rem) {
                const double rr = (double)((rem + 255) / 256);
                cost += L > rr * mfma_wg ? L : rr * mfma_wg;
            }
            if (ks > 1) cost += 8000.0;
            if (cost < best) { best = cost; bi = i; }
        }
        if (g_tune[0] >= 0) bi = g_tune[0];
        switch (bi) {
            case 0: return launch_conv<KS, VEC, 1, 1, 4>(p, s);
            case 1: return launch_conv<KS, VEC, 2, 2, 4>(p, s);
            case 2: return launch_conv<KS, VEC, 1, 2, 2>(p, s);
            case 3: return launch_conv<KS, VEC, 1, 1, 2>(p, s);
            case 4: return launch_conv<KS, VEC, 4, 1, 4>(p, s);
            case 5: return launch_conv<KS, VEC, 2, 1, 2>(p, s);
            case 6: return launch_conv<KS, VEC, 1, 1, 1>(p, s);
            default: return launch_conv<KS, VEC, 4, 1, 1>(p, s);
        }
    }
}

// Generic tile kernel only, and only when it runs without split-K (used by the fused unpool epilogue).
int dispatch_conv_generic_nosplit(ConvP& p, hipStream_t s);

template <int KS>
int dispatch_conv_vec(ConvP& p, hipStream_t s)
{
    if ((p.Cin & 15) == 0) return dispatch_conv<KS, 4>(p, s);
    if ((p.Cin & 7) == 0) return dispatch_conv<KS, 2>(p, s);
    return dispatch_conv<KS, 1>(p, s);
}

int dispatch_conv_generic_nosplit(ConvP& p, hipStream_t s) { return dispatch_conv_vec<3>(p, s); }

template <int KS, int WM, int WN, int WAVES_CO, int WAVES_CI, int BPX>
int launch_wgrad(WgP& p, hipStream_t s)
{
    constexpr int WAVES_K = 4 / (WAVES_CO * WAVES_CI);
    constexpr int BCO = 16 * WM * WAVES_CO, BCI = 16 * WN * WAVES_CI;
    constexpr int SZ = PixStride<BCO>::value, SX = PixStride<BCI>::value;
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH; p.ntiles = g.ntiles;
    const int HT = (1 << g.lgTH) + KS - 1, WT = (1 << g.lgTW) + KS - 1;
    constexpr int XMAX = KS == 1 ? BPX : (KS == 3 ? (BPX * 9) / 4 : 16 * BPX);
    if (g.TN * HT * WT > XMAX) return PG_E_UNSUP;
    p.mWT = (unsigned)((1ull << 32) / (unsigned)WT) + 1u; p.mHT = (unsigned)((1ull << 32) / (unsigned)HT) + 1u;
    size_t smem = ((size_t)BPX * SZ + (size_t)g.TN * HT * WT * SX) * sizeof(float);
    const size_t red = WAVES_K > 1 ? (size_t)WAVES_CO * WAVES_CI * (KS * KS * WM * WN + WM) * 256 * sizeof(float) : 0;
    if (red > smem) smem = red;
    const int gy = (p.Cout + BCO - 1) / BCO, gz_ = (p.Cin + BCI - 1) / BCI;
    int chunks = (512 + gy * gz_ - 1) / (gy * gz_);        // ~512 workgroups: fills 256 CUs twice over while
    if (g_tune[2] > 0) chunks = g_tune[2];                 // (tuning sweep)
    if (chunks > g.ntiles) chunks = g.ntiles;              // keeping the commit traffic (chunks x |dW|) small
    if (chunks < 1) chunks = 1;
    p.tiles_per_block = (g.ntiles + chunks - 1) / chunks;
    chunks = (g.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    p.atomic = 1;        // fire-and-forget L2 atomics even for a sole writer: a load-add-store commit serialises on the load latency (+5 us per launch)
    auto kern = conv_wgrad_kernel<KS, WM, WN, WAVES_CO, WAVES_CI, BPX>;
    if (int rc = set_smem(kern, smem)) return rc;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_wgrad_kernel<%d, %d, %d, %d, %d, %d>", KS, WM, WN, WAVES_CO, WAVES_CI, BPX);
    hipLaunchKernelGGL(kern, dim3(chunks, gy, gz_), dim3(256), smem, s, p);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------
// Weight gradient of the 8-channel layers (8->8, 8->16, 16->8 at 1024^2): a 16x16x4 MFMA tile would be 75 % / 50 %
// zero padding.  v_mfma_f32_4x4x1_16B_f32 computes SIXTEEN independent 4x4 outer products per instruction at the
// same FLOP rate: the CO x CI outer product of one pixel is NB = (CO/4)*(CI/4) blocks, so one instruction takes
// 16/NB pixels (4 for 8x8, 2 for 8x16) with every lane doing useful work.  Lane l: block b = l>>2, A row / B col
// = l&3; D register r of lane 4b+j is element [r][j] of block b.  Blocks of the same (co-quad, ci-quad) but
// different pixel slot are separate accumulators, summed by xor-shuffles before the workgroup reduction.
// FIX: the tile is 16 x 4 pixels of one image (every layer this kernel serves from 16 x 16 maps up): the k-step -> LDS address map is
// then a per-lane base plus compile-time constants, i.e. immediate offsets of the ds_reads instead of ~12 VALU instructions per k-step
// (rocprofv3 SQ_INSTS_VALU per wave, 8->16 @1024^2 n9: 9.2 k non-MFMA VALU next to 10.4 k MFMAs before).
template <int CO, int CI, int BPX, bool FIX>
__global__ __launch_bounds__(256) void conv_wgrad_thin_kernel(WgP p)
{
    constexpr int KS = 3, TAPS = 9;
    constexpr int QO = CO / 4, QI = CI / 4, NB = QO * QI;
    constexpr int PPM = NB <= 16 ? 16 / NB : 1;                  // pixels per MFMA
    constexpr int GQ = NB <= 16 ? 1 : NB / 16;                   // MFMAs (groups of 16 blocks) per pixel and tap
    static_assert(NB <= 16 || (16 % QI) == 0, "the B operand must be shared by the block groups");
    constexpr int SZ = PixStride<CO>::value, SX = PixStride<CI>::value;
    constexpr int ZV = CO / 4, XV = CI / 4;
    constexpr int ZPT = (BPX * ZV + 255) / 256;
    constexpr int XMAX = (BPX * 9) / 4;
    constexpr int XPT = (XMAX * XV + 255) / 256;
    extern __shared__ __align__(16) float lds[];

    static_assert(!FIX || BPX == 64, "the fixed geometry is 16 x 4 pixels");
    const int TW = FIX ? 16 : 1 << p.lgTW, TH = FIX ? 4 : 1 << p.lgTH;
    const int HT = TH + KS - 1, WT = TW + KS - 1;
    float* gzt = lds;                        // [BPX][SZ]
    float* xt = lds + BPX * SZ;              // [TN*HT*WT][SX]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int blk = lane >> 2, i4 = lane & 3;
    const int hi = blk % QI, slot = NB <= 16 ? blk / NB : 0;     // group q of this lane: cout quad (16q + blk) / QI
    int ho[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) ho[q] = ((16 * q + blk) / QI) % QO;
    const bool do_bias = p.db != nullptr;

    f32x4 acc[GQ][TAPS];
    float bsum[GQ];
#pragma unroll
    for (int q = 0; q < GQ; ++q) {
        bsum[q] = 0.f;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[q][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int npix = p.TN * HT * WT;
    const int xH = p.ups ? (p.Hin >> 1) : p.Hin, xW = p.ups ? (p.Win >> 1) : p.Win;
    // Per-thread load descriptors, computed ONCE: byte offsets relative to the tile origin (the per-tile part of every address
    // is a wave-uniform scalar) and, for the halo pixels of x, which tile edge they sit on.  The phase trace of round 2
    // (tools/exp/wgrad_trace.py) showed the per-tile address arithmetic of the previous version (integer divisions of the tile
    // index, per-load multiplies and range checks) costing 1400 of the 3700 cycles a tile took.
    int zq[ZPT], zc[ZPT];
    int xdst[XPT];
    unsigned zrel[ZPT], brel[ZPT];
    int xrel[XPT], xedge[XPT];                                   // xedge: bit 0 top, 1 bottom, 2 left, 3 right halo; -1 = unused slot
    const int zH = p.gbytes ? (p.Hout >> 1) : p.Hout, zW = p.gbytes ? (p.Wout >> 1) : p.Wout;
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / ZV, v = idx - q * ZV;
        zq[i] = idx < BPX * ZV ? q : -1;
        zc[i] = 4 * v;
        const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
        const int zh = p.gbytes ? (th >> 1) : th, zw = p.gbytes ? (tw >> 1) : tw;
        zrel[i] = zq[i] >= 0 ? 4u * (unsigned)(((tn * zH + zh) * zW + zw) * CO + 4 * v) : PG_OOB;
        brel[i] = (unsigned)(((tn * p.Hout + th) * p.Wout + tw) * (CO / 4) + v);
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / XV, v = idx - q * XV;
        const int r2 = (int)__umulhi((unsigned)q, p.mWT), tw = q - r2 * WT;
        const int tn = (int)__umulhi((unsigned)r2, p.mHT), th = r2 - tn * HT;
        xdst[i] = q * SX + 4 * v;
        int ih = th - p.pad, iw = tw - p.pad;
        if (p.ups) { ih >>= 1; iw >>= 1; }                       // nearest-x2 upsample fused into the gather (tile origins are even)
        xrel[i] = 4 * (((tn * xH + ih) * xW + iw) * CI + 4 * v);
        xedge[i] = q < npix ? ((th < p.pad ? 1 : 0) | (th >= TH + p.pad ? 2 : 0) | (tw < p.pad ? 4 : 0) | (tw >= TW + p.pad ? 8 : 0)) : -1;
    }
    int tapoff[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) tapoff[tp] = ((tp / KS) * WT + (tp % KS)) * SX;
    int abase[GQ], bbase;                                        // FIX: LDS offsets of this lane's fragments at k-step 0
    {
        const int q0 = PPM * wave + slot;
#pragma unroll
        for (int g = 0; g < GQ; ++g) abase[g] = q0 * SZ + 4 * ho[g] + i4;
        bbase = q0 * SX + 4 * hi + i4;
    }

    float4 zreg[ZPT], xreg[XPT];
    unsigned char zb[ZPT];                   // sign bytes of the prefetched gz values (pool adjoint in the gather)
#pragma unroll
    for (int i = 0; i < ZPT; ++i) zb[i] = 0;
    const size_t zimg = (size_t)zH * zW * CO, ximg = (size_t)xH * xW * CI;
    int f_tw = 0, f_th = 0, f_n = 0;         // tile coordinates of the NEXT fetch (tiles are fetched in order: no divisions per tile)
    auto fetch_seek = [&](int tile) {
        int t = tile;
        f_tw = t % p.tilesW; t /= p.tilesW;
        f_th = t % p.tilesH; f_n = t / p.tilesH;
    };
    auto fetch = [&]() {
        const int n0 = f_n * p.TN;
        const int oh0 = f_th << p.lgTH, ow0 = f_tw << p.lgTW;
        // raw buffers over the TN images of this tile (bufload.h): PG_OOB / beyond-the-records = zero fill, no branch per load
        const int nimg = min(p.TN, p.N - n0);
        const __amdgpu_buffer_rsrc_t rz = pg_make_rsrc(p.gz + (size_t)n0 * zimg, (unsigned)((size_t)nimg * zimg * 4));
        const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(p.x + (size_t)n0 * ximg, (unsigned)((size_t)nimg * ximg * 4));
        const unsigned zorg = 4u * (unsigned)((((p.gbytes ? oh0 >> 1 : oh0) * zW) + (p.gbytes ? ow0 >> 1 : ow0)) * CO);
        const int xorg = 4 * ((((p.ups ? oh0 >> 1 : oh0) * xW) + (p.ups ? ow0 >> 1 : ow0)) * CI);
        // tile edges that coincide with the image border: their halo pixels are outside the image
        const int border = (oh0 == 0 ? 1 : 0) | (oh0 + TH >= p.Hout ? 2 : 0) | (ow0 == 0 ? 4 : 0) | (ow0 + TW >= p.Wout ? 8 : 0);
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            zreg[i] = pg_buf_load4(rz, zrel[i], zorg);
            if (p.gbytes) {
                // the byte is applied when the prefetched value is stored to LDS (next iteration): a multiply here would
                // wait for the load and serialise the register prefetch
                const bool ok = zq[i] >= 0 && (zq[i] >> (p.lgTW + p.lgTH)) < nimg;
                zb[i] = ok ? p.gbytes[((size_t)n0 * p.Hout + oh0) * p.Wout * (CO / 4) + (size_t)ow0 * (CO / 4) + brel[i]] : (unsigned char)0;
            }
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bool ok = xedge[i] >= 0 && (xedge[i] & border) == 0;
            xreg[i] = pg_buf_load4(rx, ok ? (unsigned)(xrel[i] + xorg) : PG_OOB, 0);
        }
        if (++f_tw == p.tilesW) { f_tw = 0; if (++f_th == p.tilesH) { f_th = 0; ++f_n; } }
    };

    const int t_begin = (int)pg_xcd_remap(blockIdx.x, gridDim.x) * p.tiles_per_block;   // neighbouring tile ranges on one XCD
    const int t_end = min(t_begin + p.tiles_per_block, p.ntiles);
    constexpr int NSTEPS = BPX / PPM, T = NSTEPS / 4;            // k-steps per tile / per wave

    fetch_seek(t_begin);
    if (t_begin < t_end) fetch();
    for (int tile = t_begin; tile < t_end; ++tile) {
        PG_WSTAMP(0);
#pragma unroll
        for (int i = 0; i < ZPT; ++i)
            if (zq[i] >= 0) {
                float4 v = zreg[i];
                if (p.gbytes) {
                    const float4 f = pg_sign_factors(zb[i], p.gslope);
                    v.x *= f.x * p.gmul; v.y *= f.y * p.gmul; v.z *= f.z * p.gmul; v.w *= f.w * p.gmul;
                }
                *reinterpret_cast<float4*>(gzt + zq[i] * SZ + zc[i]) = v;
            }
#pragma unroll
        for (int i = 0; i < XPT; ++i)
            if (xedge[i] >= 0) *reinterpret_cast<float4*>(xt + xdst[i]) = xreg[i];
        PG_WSTAMP(1);
        __syncthreads();
        PG_WSTAMP(2);
        if (tile + 1 < t_end) fetch();
        PG_WSTAMP(3);

        auto load_frags = [&](int kstep, float (&af)[GQ], float (&bf)[TAPS]) {      // k-step of this wave: step = wave + 4 kstep
            if constexpr (FIX) {
                // pixel q = q0 + 4 PPM kstep with q0 = PPM wave + slot < 4 PPM <= 16: q0 stays inside tile row 0, kstep walks along the
                // row (16 / (4 PPM) steps) and then down: every offset below is a per-lane base + a compile-time constant
                constexpr int PER_ROW = 16 / (4 * PPM);
                const int th = kstep / PER_ROW, dw_ = (kstep % PER_ROW) * 4 * PPM;
#pragma unroll
                for (int g = 0; g < GQ; ++g) af[g] = gzt[abase[g] + (th * 16 + dw_) * SZ];
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) bf[tp] = xt[bbase + ((th + tp / KS) * 18 + dw_ + tp % KS) * SX];
            } else {
                const int q = PPM * (wave + 4 * kstep) + slot;
                const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
#pragma unroll
                for (int g = 0; g < GQ; ++g) af[g] = gzt[q * SZ + 4 * ho[g] + i4];
                const int bo = ((tn * HT + th) * WT + tw) * SX + 4 * hi + i4;
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) bf[tp] = xt[bo + tapoff[tp]];
            }
        };
        auto mfmas = [&](const float (&af)[GQ], const float (&bf)[TAPS]) {
#pragma unroll
            for (int g = 0; g < GQ; ++g) {
                bsum[g] += af[g];
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) acc[g][tp] = __builtin_amdgcn_mfma_f32_4x4x1f32(af[g], bf[tp], acc[g][tp], 0, 0, 0);
            }
        };
        float a[2][GQ], b[2][TAPS];
        static_assert(T % 2 == 0, "k-steps per wave must be even");
        load_frags(0, a[0], b[0]);
#pragma unroll
        for (int s2 = 0; s2 < T; s2 += 2) {
            load_frags(s2 + 1, a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (s2 + 2 < T) load_frags(s2 + 2, a[0], b[0]);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a[1], b[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PG_WSTAMP(4);
        __syncthreads();
        PG_WSTAMP(5);
    }

    // ---- sum the pixel slots (lanes 4*NB apart), then the 4 waves through LDS, then ONE commit per workgroup
#pragma unroll
    for (int g = 0; g < GQ; ++g) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[g][tp][r];
                if (PPM >= 2) v += __shfl_xor(v, 32, 64);
                if (PPM >= 4) v += __shfl_xor(v, 16, 64);
                acc[g][tp][r] = v;
            }
        if (PPM >= 2) bsum[g] += __shfl_xor(bsum[g], 32, 64);
        if (PPM >= 4) bsum[g] += __shfl_xor(bsum[g], 16, 64);
    }
    constexpr int NL = NB <= 16 ? 4 * NB : 64;                   // lanes holding distinct results
    constexpr int NE = GQ * (TAPS * 4 + 1);                      // values per lane: [group][tap*4 + r | bias]
    float* red = lds;                                            // [wave][NE][NL]
    if (lane < NL) {
#pragma unroll
        for (int g = 0; g < GQ; ++g) {
#pragma unroll
            for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[(wave * NE + g * (TAPS * 4 + 1) + tp * 4 + r) * NL + lane] = acc[g][tp][r];
            red[(wave * NE + g * (TAPS * 4 + 1) + TAPS * 4) * NL + lane] = bsum[g];
        }
    }
    __syncthreads();
    // one thread per element sums the four waves and commits
    for (int e = tid; e < NE * NL; e += 256) {
        const int l = e % NL, ve = e / NL;
        const int g = ve / (TAPS * 4 + 1), slot_e = ve % (TAPS * 4 + 1);   // slot_e = tp*4 + r, or TAPS*4 for the bias
        const float v = (red[e] + red[e + NE * NL]) + (red[e + 2 * NE * NL] + red[e + 3 * NE * NL]);
        const int b_ = l >> 2, j = l & 3;
        const int hi_ = b_ % QI, ho_ = ((16 * g + b_) / QI) % QO;
        if (slot_e < TAPS * 4) {
            const int tp = slot_e >> 2, r = slot_e & 3;
            float* dst = p.dw + ((size_t)(tp * CO + 4 * ho_ + r) * CI + 4 * hi_ + j);
            if (p.atomic) atomicAdd(dst, v * p.scale); else *dst += v * p.scale;
        } else if (do_bias && hi_ == 0) {                        // lane (ho, hi = 0, i) carries sum_p gz[p][4*ho + i]
            float* dst = p.db + 4 * ho_ + j;
            if (p.atomic) atomicAdd(dst, v); else *dst += v;
        }
    }
}

template <int BPX>
int launch_wgrad_thin(WgP& p, hipStream_t s)
{
    if (g_tune[1] != 20) {                        // row-streaming kernel (conv_strip.hip) where the shape allows; 20: tile kernel (A/B)
        const int rc = pgk::launch_wgrad_strip(p, s, g_last_kernel, sizeof(g_last_kernel));
        if (rc != PG_E_UNSUP) return rc;
    }
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH; p.ntiles = g.ntiles;
    const int HT = (1 << g.lgTH) + 2, WT = (1 << g.lgTW) + 2;
    if (g.TN * HT * WT > (BPX * 9) / 4) return PG_E_UNSUP;
    if ((long long)g.TN * p.Hout * p.Wout * (p.Cout > p.Cin ? p.Cout : p.Cin) * 4 >= (1ll << 31)) return PG_E_UNSUP;     // 32-bit buffer offsets
    p.mWT = (unsigned)((1ull << 32) / (unsigned)WT) + 1u; p.mHT = (unsigned)((1ull << 32) / (unsigned)HT) + 1u;
    const int sz = p.Cout == 16 ? 16 : p.Cout + 16, sx = p.Cin == 16 ? 16 : p.Cin + 16;      // PixStride<C>
    size_t smem = ((size_t)BPX * sz + (size_t)g.TN * HT * WT * sx) * sizeof(float);
    const int nb = (p.Cout / 4) * (p.Cin / 4);
    const size_t red = (size_t)4 * (nb <= 16 ? 1 : nb / 16) * 37 * (nb <= 16 ? 4 * nb : 64) * sizeof(float);   // [wave][NE][NL]
    if (red > smem) smem = red;
    int chunks = 1024; if (chunks > g.ntiles) chunks = g.ntiles;
    p.tiles_per_block = (g.ntiles + chunks - 1) / chunks;
    chunks = (g.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    p.atomic = chunks > 1 ? 1 : 0;
    snprintf(g_last_kernel, sizeof(g_last_kernel), "conv_wgrad_thin_kernel<%d, %d, %d>", p.Cout, p.Cin, BPX);
#ifdef PG_WINO_TRACE
    p.trace = g_wgrad_trace;
#endif
    static const int fix_env = getenv("PG_WGRAD_THIN_FIX") ? atoi(getenv("PG_WGRAD_THIN_FIX")) : 1;
    const bool fix = BPX == 64 && g.lgTW == 4 && g.lgTH == 2 && g.TN == 1 && fix_env;
#define THIN(CO_, CI_) { if (fix) { auto kern = conv_wgrad_thin_kernel<CO_, CI_, BPX, BPX == 64>; if (int rc = set_smem(kern, smem)) return rc; \
                             hipLaunchKernelGGL(kern, dim3(chunks), dim3(256), smem, s, p); } \
                         else { auto kern = conv_wgrad_thin_kernel<CO_, CI_, BPX, false>; if (int rc = set_smem(kern, smem)) return rc; \
                             hipLaunchKernelGGL(kern, dim3(chunks), dim3(256), smem, s, p); } }
    if (p.Cout == 8 && p.Cin == 8) THIN(8, 8)
    else if (p.Cout == 16 && p.Cin == 8) THIN(16, 8)
    else if (p.Cout == 8 && p.Cin == 16) THIN(8, 16)
    else if (p.Cout == 32 && p.Cin == 16) THIN(32, 16)
    else if (p.Cout == 16 && p.Cin == 32) THIN(16, 32)
    else return PG_E_UNSUP;
#undef THIN
    return (int)hipGetLastError();
}

template <int KS>
int dispatch_wgrad(WgP& p, hipStream_t s)
{
    if constexpr (KS == 4) {
        return launch_wgrad<KS, 1, 1, 2, 2, 16>(p, s);                                        // 32x32 block, 16-px tiles
    } else {
        const long long M = (long long)p.N * p.Hout * p.Wout;
        if (M <= 32) return launch_wgrad<KS, 1, 1, 2, 2, 16>(p, s);
        if constexpr (KS == 3) {
            // measured (tools/sweeps/sweep_wgrad_thin.py): block-MFMA wins on 32 -> 16 always, on 16 -> 32 below ~1.5 M pixels
            if (g_tune[1] != 8 && ((p.Cout == 16 && p.Cin == 32) || (p.Cout == 32 && p.Cin == 16 && M < 1500000)))
                return launch_wgrad_thin<64>(p, s);
        }
        if (p.Cout <= 16 && p.Cin <= 16) {
            if constexpr (KS == 3) {
                // 8-channel sides: the 16x16x4 tile would be 50-75 % padding -> 4x4x1 block MFMA kernel
                if (g_tune[1] != 8 && (p.Cout == 8 || p.Cin == 8) && (p.Cout == 8 || p.Cout == 16) && (p.Cin == 8 || p.Cin == 16))
                    return launch_wgrad_thin<64>(p, s);
            }
            return launch_wgrad<KS, 1, 1, 1, 1, 128>(p, s);     // 16x16 block
        }
        if constexpr (KS == 3) {
            switch (g_tune[1]) {                                                               // tuning sweep only
                case 1: return launch_wgrad<KS, 2, 1, 1, 1, 128>(p, s);
                case 3: return launch_wgrad<KS, 2, 1, 2, 1, 64>(p, s);
                default: break;
            }
        }
        if constexpr (KS == 3) {
            // measured (tools/sweeps/sweep_wgrad.py): with >= ~4e8 MACs per tap the 64-cout block (two K-waves) wins on
            // >= 64 input channels and 128-pixel tiles win on the narrow layers; small launches keep 32x16 / 64 px
            const double macs = (double)M * p.Cout * p.Cin;
            if (g_tune[1] < 0 && macs >= 4e8) {
                if (p.Cin >= 64) return launch_wgrad<KS, 2, 1, 2, 1, 64>(p, s);
                return launch_wgrad<KS, 2, 1, 1, 1, 128>(p, s);
            }
        }
        return launch_wgrad<KS, 2, 1, 1, 1, 64>(p, s);     // 32(cout) x 16(cin) block, 4 waves split the pixels
    }
}

}  // namespace

extern "C" int pg_avgpool2_fwd(const float* x, const float* other, float* y, int N, int H, int W, int C,
                               float a, float b, pg_stream_t stream);

extern "C" int pg_avgpool2_bwd(const float* gy, const float* mask, float* gx, int N, int H, int W, int C,
                               float mul, float mask_slope, pg_stream_t stream);
extern "C" int pg_pixelnorm_fwd(const float* x, float* y, float* r, int64_t P, int C, float eps, pg_stream_t stream);
extern "C" int pg_pixelnorm_lrelu_bwd(const float* gy, const float* y, const float* r, float* gz,
                                      int64_t P, int C, float slope, pg_stream_t stream);

static int conv2d_impl(const float* x, const float* w, const float* bias, const float* mask, float* y,
                       float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
                       int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                       float scale, float slope, float mask_slope, pg_stream_t stream,
                       float* yup = nullptr, const float* upmask = nullptr, float up_mul = 1.f,
                       float* pn_r = nullptr, float pn_eps = 0.f, const float* pnb_y = nullptr, const float* pnb_r = nullptr)
{
    if (!x || !w || !y || N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((Cin & 3) || (Cout & 3)) return PG_E_ALIGN;
    const int flags = ups;                                  // PG_FLAG_*: bit 0 = nearest x2 upsample of x
    ups = flags & PG_FLAG_UPSAMPLE;
    ConvP p;
    p.mask_bytes = (flags & PG_FLAG_MASK_BYTES) ? 1 : 0; p.y_bytes = (flags & PG_FLAG_Y_BYTES) ? 1 : 0;
    p.ysigns = nullptr;
    p.gbytes = nullptr; p.gmul = 1.f; p.gslope = 1.f;
    if (flags & PG_FLAG_SIGNS_OUT) {                        // forward mode: the (otherwise unused) mask argument is the byte output
        if (!mask || p.mask_bytes) return PG_E_ARG;
        p.ysigns = reinterpret_cast<unsigned char*>(const_cast<float*>(mask));
        mask = nullptr;
    }
    p.x = x; p.w = w; p.bias = bias; p.mask = mask; p.y = y;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout; p.KS = KS; p.pad = pad; p.ups = ups;
    p.Hout = Hin + 2 * pad - KS + 1; p.Wout = Win + 2 * pad - KS + 1;
    if (p.Hout <= 0 || p.Wout <= 0 || !is_pow2(p.Hout) || !is_pow2(p.Wout)) return PG_E_UNSUP;
    if (ups && ((Hin | Win) & 1)) return PG_E_ARG;
    if (ypool && ((p.Hout | p.Wout) & 1)) return PG_E_ARG;
    // 32-bit element offsets inside the kernels
    if ((long long)N * Hin * Win * Cin >= (1ll << 31) || (long long)N * p.Hout * p.Wout * Cout >= (1ll << 31) ||
        (long long)KS * KS * Cout * Cin >= (1ll << 31)) return PG_E_UNSUP;
    p.scale = scale; p.slope = slope; p.mask_slope = mask_slope;
    p.ksplit = 1;
    const bool fuse_pool = ypool != nullptr && KS == 3 && g_tune[3] != 3;
    p.ypool = fuse_pool ? ypool : nullptr; p.pool_other = pool_other; p.pool_a = pool_a; p.pool_b = pool_b;
    p.pool_only = pool_only;
    // the unpool epilogue exists in the generic tile kernel only (no split-K): everything else unpools in a second pass
    const bool fuse_up = yup != nullptr && KS == 3 && g_tune[3] != 10;
    p.yup = nullptr; p.upmask = upmask; p.up_mul = up_mul;
    p.pn_r = nullptr; p.pn_eps = pn_eps; p.pnb_y = nullptr; p.pnb_r = pnb_r;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (p.mask_bytes || p.y_bytes || p.ysigns) {
        // sign-byte activations exist in the epilogues of the generic tile kernel (no split-K, no second pass) and of the
        // 8-cout block-MFMA kernel only: PG_E_UNSUP tells the caller to redo the layer with fp32 masks
        if (KS != 3 || (p.y_bytes && !ypool) || pn_r || pnb_y) return PG_E_UNSUP;
        p.ypool = ypool;
        p.yup = yup;
        const bool thin_plain = !p.y_bytes && !yup && !ypool && ((Cout == 8 && (Cin == 8 || Cin == 16)) || (Cout == 16 && Cin == 8 && mask));
        const bool thin_pool = g_tune[3] != 17 && !yup && ypool && Cout == 16 && Cin == 8;      // 8->16 + pool (forward: sign bytes out; tangent: masked): +8..14 % over the generic tile kernel (tools/sweeps/bench_thin16pool.py)
        const bool thin_b = pad == 1 && (thin_plain || thin_pool) && (p.Wout & 31) == 0 && (p.Hout & 7) == 0 && g_tune[3] != 2;
        if (thin_b) return dispatch_thin(p, s);
        return dispatch_conv_generic_nosplit(p, s);
    }
    if (pnb_y && pnb_r && KS == 3 && Cout <= 32 && g_tune[3] != 12) {
        p.pnb_y = pnb_y;
        const bool thin_pn = pad == 1 && Cout == 8 && (Cin == 8 || Cin == 16) && (p.Wout & 31) == 0 && (p.Hout & 7) == 0 && g_tune[3] != 2;
        rc = thin_pn ? dispatch_thin(p, s) : dispatch_conv_generic_nosplit(p, s);
        if (rc == 0) return 0;
        if (rc != PG_E_UNSUP) return rc;
        p.pnb_y = nullptr;
    }
    if (pn_r && KS == 3 && Cout <= 32 && !mask && g_tune[3] != 11) {   // fused PixelNorm: thin kernel (8 couts) or one-row generic tiles
        p.pn_r = pn_r;
        const bool thin_pn = pad == 1 && Cout == 8 && (Cin == 8 || Cin == 16) && (p.Wout & 31) == 0 && (p.Hout & 7) == 0 && g_tune[3] != 2;
        rc = thin_pn ? dispatch_thin(p, s) : dispatch_conv_generic_nosplit(p, s);
        if (rc == 0) return 0;
        if (rc != PG_E_UNSUP) return rc;
        p.pn_r = nullptr;
    }
    if (fuse_up) {
        p.yup = yup;
        rc = dispatch_conv_generic_nosplit(p, s);
        if (rc == 0) return 0;
        if (rc != PG_E_UNSUP) return rc;
        p.yup = nullptr;
    }
    // measured (tools/sweeps/sweep_thin8.py): 1.5-1.7x on 8 couts; on 16 couts only the masked 8->16 launch gains (the
    // 16x16x4 tile has no padding there), 32 input channels lose -> those stay on the generic kernel
    const bool thin_ok = KS == 3 && pad == 1 && ((Cout == 8 && (Cin == 8 || Cin == 16)) || (Cout == 16 && Cin == 8 && mask && !ypool)) &&
                         (p.Wout & 31) == 0 && (p.Hout & 7) == 0 && g_tune[3] != 2;
    if (thin_ok) {
        rc = dispatch_thin(p, s);                       // pools in its own epilogue when p.ypool is set
        if (rc) return rc;
        if (ypool && !p.ypool)
            return pg_avgpool2_fwd(y, pool_other, ypool, N, p.Hout >> 1, p.Wout >> 1, Cout, pool_a, pool_b, stream);
        if (yup) return pg_avgpool2_bwd(y, upmask, yup, N, p.Hout, p.Wout, Cout, up_mul, mask_slope, stream);
        if (pn_r) return pg_pixelnorm_fwd(y, y, pn_r, (int64_t)N * p.Hout * p.Wout, Cout, pn_eps, stream);
        if (pnb_y) return pg_pixelnorm_lrelu_bwd(y, pnb_y, pnb_r, y, (int64_t)N * p.Hout * p.Wout, Cout, mask_slope, stream);
        return 0;
    }
    if (KS == 4 && !ups && k4_dense_ok(Cin, Cout) && g_tune[3] != 1 &&
        ((pad == 3 && Hin == 1 && Win == 1) || (pad == 0 && Hin == 4 && Win == 4)))
        rc = launch_k4_conv(p, s);
    else switch (KS) {
        case 1: rc = dispatch_conv_vec<1>(p, s); break;
        case 3: rc = dispatch_conv_vec<3>(p, s); break;
        case 4: rc = dispatch_conv_vec<4>(p, s); break;
        default: return PG_E_UNSUP;
    }
    if (rc) return rc;
    if (ypool && !(fuse_pool && p.ksplit == 1))          // split-K / non-3x3 launches pool in a second pass over y
        return pg_avgpool2_fwd(y, pool_other, ypool, N, p.Hout >> 1, p.Wout >> 1, Cout, pool_a, pool_b, stream);
    if (yup) return pg_avgpool2_bwd(y, upmask, yup, N, p.Hout, p.Wout, Cout, up_mul, mask_slope, stream);
    if (pn_r) return pg_pixelnorm_fwd(y, y, pn_r, (int64_t)N * p.Hout * p.Wout, Cout, pn_eps, stream);
    if (pnb_y) return pg_pixelnorm_lrelu_bwd(y, pnb_y, pnb_r, y, (int64_t)N * p.Hout * p.Wout, Cout, mask_slope, stream);
    return 0;
}

extern "C" int pg_conv2d_nhwc(const float* x, const float* w, const float* bias, const float* mask, float* y,
                              int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                              float scale, float slope, float mask_slope, pg_stream_t stream)
{
    return conv2d_impl(x, w, bias, mask, y, nullptr, nullptr, 1.f, 0.f, 0, N, Hin, Win, Cin, Cout, KS, pad, ups,
                       scale, slope, mask_slope, stream);
}

extern "C" int pg_conv2d_pixelnorm_nhwc(const float* x, const float* w, const float* bias, float* y, float* r,
                                        int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                                        float scale, float slope, float eps, pg_stream_t stream)
{
    if (!r) return PG_E_ARG;
    return conv2d_impl(x, w, bias, nullptr, y, nullptr, nullptr, 1.f, 0.f, 0, N, Hin, Win, Cin, Cout, KS, pad, ups,
                       scale, slope, 0.2f, stream, nullptr, nullptr, 1.f, r, eps);
}

extern "C" int pg_conv2d_pnbwd_nhwc(const float* x, const float* w, const float* ysaved, const float* r, float* y,
                                    int N, int Hin, int Win, int Cin, int Cout, int KS, int pad,
                                    float scale, float slope, pg_stream_t stream)
{
    if (!ysaved) return PG_E_ARG;
    return conv2d_impl(x, w, nullptr, nullptr, y, nullptr, nullptr, 1.f, 0.f, 0, N, Hin, Win, Cin, Cout, KS, pad, 0,
                       scale, 1.0f, slope, stream, nullptr, nullptr, 1.f, nullptr, 0.f, ysaved, r);
}

extern "C" int pg_conv2d_unpool_nhwc(const float* x, const float* w, const float* upmask, float* y, float* yup,
                                     int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int flags,
                                     float scale, float up_mul, float mask_slope, pg_stream_t stream)
{
    if (!yup) return PG_E_ARG;
    return conv2d_impl(x, w, nullptr, nullptr, y, nullptr, nullptr, 1.f, 0.f, 0, N, Hin, Win, Cin, Cout, KS, pad, flags & PG_FLAG_MASK_BYTES,
                       scale, 1.0f, mask_slope, stream, yup, upmask, up_mul);
}

extern "C" int pg_conv2d_pool_nhwc(const float* x, const float* w, const float* bias, const float* mask, float* y,
                                   float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
                                   int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                                   float scale, float slope, float mask_slope, pg_stream_t stream)
{
    if (!ypool) return PG_E_ARG;
    return conv2d_impl(x, w, bias, mask, y, ypool, pool_other, pool_a, pool_b, pool_only, N, Hin, Win, Cin, Cout, KS, pad, ups,
                       scale, slope, mask_slope, stream);
}

extern "C" int pg_conv2d_wgrad_nhwc(const float* x, const float* gz, float* dw, float* db,
                                    int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                                    float scale, pg_stream_t stream)
{
    if (!x || !gz || !dw || N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((Cin & 3) || (Cout & 3)) return PG_E_ALIGN;
    WgP p;
    p.x = x; p.gz = gz; p.dw = dw; p.db = db;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout; p.pad = pad; p.ups = ups;
    p.Hout = Hin + 2 * pad - KS + 1; p.Wout = Win + 2 * pad - KS + 1;
    if (p.Hout <= 0 || p.Wout <= 0 || !is_pow2(p.Hout) || !is_pow2(p.Wout)) return PG_E_UNSUP;
    if (ups && ((Hin | Win) & 1)) return PG_E_ARG;
    p.scale = scale;
    p.gbytes = nullptr; p.gmul = 1.f; p.gslope = 1.f;
    hipStream_t s = (hipStream_t)stream;
    if (KS == 4 && !ups && k4_dense_ok(Cin, Cout) && g_tune[3] != 1 &&
        ((pad == 3 && Hin == 1 && Win == 1) || (pad == 0 && Hin == 4 && Win == 4)))
        return launch_k4_wgrad(p, s);
    switch (KS) {
        case 1: return dispatch_wgrad<1>(p, s);
        case 3: return dispatch_wgrad<3>(p, s);
        case 4: return dispatch_wgrad<4>(p, s);
        default: return PG_E_UNSUP;
    }
}

// Backward-data conv / weight gradient of a DBlock's c2 layer whose incoming gradient is the POOL ADJOINT of the coarser
// block's gradient g: instead of materialising gz2 = gmul * upsample2(g) * lrelu'(a2) (16 channels at 1024^2: the largest
// tensor of the backward sweep, written once and read twice), both consumers evaluate it in their input gathers from g
// (a quarter of the pixels) and the sign bytes of a2.  Block-MFMA kernels of the 8/16-channel layers only (PG_E_UNSUP otherwise).
extern "C" int pg_conv2d_unpooled_nhwc(const float* g, const float* w, const unsigned char* gbytes, float gmul, float gslope,
                                       const float* mask, float* y, int N, int Hin, int Win, int Cin, int Cout, int flags,
                                       float scale, float mask_slope, pg_stream_t stream)
{
    if (!g || !w || !gbytes || !y || N <= 0 || Hin <= 0 || Win <= 0) return PG_E_ARG;
    if ((Hin | Win) & 1) return PG_E_ARG;
    if (!(Cout == 8 && (Cin == 8 || Cin == 16)) || (Win & 31) || (Hin & 7) || !is_pow2(Hin) || !is_pow2(Win)) return PG_E_UNSUP;
    if ((long long)N * Hin * Win * Cin >= (1ll << 31)) return PG_E_UNSUP;
    ConvP p;
    p.x = g; p.w = w; p.bias = nullptr; p.mask = mask; p.y = y;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout; p.KS = 3; p.pad = 1; p.ups = 1;
    p.Hout = Hin; p.Wout = Win;
    p.scale = scale; p.slope = 1.f; p.mask_slope = mask_slope; p.ksplit = 1;
    p.ypool = nullptr; p.pool_other = nullptr; p.pool_a = 1.f; p.pool_b = 0.f; p.pool_only = 0;
    p.yup = nullptr; p.upmask = nullptr; p.up_mul = 1.f;
    p.pn_r = nullptr; p.pn_eps = 0.f; p.pnb_y = nullptr; p.pnb_r = nullptr;
    p.mask_bytes = (flags & PG_FLAG_MASK_BYTES) ? 1 : 0; p.y_bytes = 0; p.ysigns = nullptr;
    p.gbytes = gbytes; p.gmul = gmul; p.gslope = gslope;
    return dispatch_thin(p, (hipStream_t)stream);
}

extern "C" int pg_conv2d_wgrad_unpooled_nhwc(const float* x, const float* g, const unsigned char* gbytes, float gmul, float gslope,
                                             float* dw, float* db, int N, int Hin, int Win, int Cin, int Cout,
                                             float scale, pg_stream_t stream)
{
    if (!x || !g || !gbytes || !dw || N <= 0 || Hin <= 0 || Win <= 0) return PG_E_ARG;
    if ((Hin | Win) & 1) return PG_E_ARG;
    if (!((Cout == 16 && Cin == 8) || (Cout == 8 && Cin == 8) || (Cout == 8 && Cin == 16)) || !is_pow2(Hin) || !is_pow2(Win) || Hin < 8 || Win < 8)
        return PG_E_UNSUP;
    WgP p;
    p.x = x; p.gz = g; p.dw = dw; p.db = db;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout; p.pad = 1; p.ups = 0;
    p.Hout = Hin; p.Wout = Win;
    p.scale = scale;
    p.gbytes = gbytes; p.gmul = gmul; p.gslope = gslope;
    return launch_wgrad_thin<64>(p, (hipStream_t)stream);
}

// A DBlock's first conv with the block's fromRGB layer evaluated in its input gather (reference network.py:145 in front of :33-36):
//   x0 = lrelu(rgb_scale * conv1x1(img, rgb_w) + rgb_b)  (never written; its sign bytes -> x_signs),  y = lrelu(scale * conv3x3(x0, w) + bias)
// for forward passes whose fromRGB output is not needed in fp32 afterwards (no weight gradient of this conv follows: the G step's pass
// through D).  The 8 -> 8 layer of the 1024^2 stage (row-streaming kernel); PG_E_UNSUP otherwise.
extern "C" int pg_conv2d_fromrgb_nhwc(const float* img, const float* rgb_w, const float* rgb_b, float rgb_scale, float rgb_slope,
                                      unsigned char* x_signs, const float* w, const float* bias, float* y, unsigned char* y_signs,
                                      int N, int C, int H, int W, int Cmid, int Cout, float scale, float slope, pg_stream_t stream)
{
    if (!img || !rgb_w || !w || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cmid <= 0 || Cout <= 0) return PG_E_ARG;
    return pgk::launch_conv_strip_fromrgb(img, rgb_w, rgb_b, rgb_scale, rgb_slope, x_signs, w, bias, y, y_signs, N, C, H, W, Cmid, Cout,
                                          scale, slope, (hipStream_t)stream, g_last_kernel, sizeof(g_last_kernel));
}

// The generator's last conv (+ bias + LeakyReLU + PixelNorm, network.py:33-41) with the block's toRGB layer (network.py:49, :138 at
// alpha = 1) in the same epilogue: the normalised activation is written for the backward pass as before and the image
//   img[n][c][h][w] = t_scale * sum_co t_w[c][co] * y[n][h][w][co] + t_b[c]
// leaves with it, instead of a second launch that reads y back.  8 -> 8 on strip-sized maps (the 1024^2 stage); PG_E_UNSUP otherwise.
extern "C" int pg_conv2d_pixelnorm_torgb_nhwc(const float* x, const float* w, const float* bias, float* y, float* r,
                                              const float* t_w, const float* t_b, float t_scale, float* img,
                                              int N, int C, int H, int W, int Cin, int Cout, float scale, float slope, float eps,
                                              pg_stream_t stream)
{
    if (!x || !w || !y || !r || !t_w || !img || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    return pgk::launch_conv_strip_pn_torgb(x, w, bias, y, r, t_w, t_b, t_scale, img, N, C, H, W, Cin, Cout, scale, slope, eps,
                                           (hipStream_t)stream, g_last_kernel, sizeof(g_last_kernel));
}

// The entry block's backward-data conv (adjoint of c1, x LeakyReLU' of fromRGB's output from its sign bytes) with fromRGB's own
// backward-data (the adjoint of the 1x1 conv of network.py:145) in the same epilogue: the gradient with respect to the IMAGE leaves
// with -- or, y == NULL, instead of -- the 8-channel gradient gf, which only fromRGB's weight gradient reads afterwards.
// Also (img, rgb_dw[, rgb_db] given): fromRGB's WEIGHT gradient accumulated in the same epilogue (one commit per workgroup) -- in the batched
// adjoint sweep nobody else reads the 8-channel gradient, so it is not written at all there (y == NULL).
extern "C" int pg_conv2d_masked_fromrgb_bwd_nhwc(const float* gz, const float* wt, const unsigned char* mask_bytes, float mask_slope, float* y,
                                                 const float* rgb_w, float rgb_scale, float* gimg,
                                                 const float* img, float* rgb_dw, float* rgb_db,
                                                 int N, int C, int H, int W, int Cin, int Cout, float scale, pg_stream_t stream)
{
    if (!gz || !wt || !mask_bytes || N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((!gimg && !rgb_dw) || (gimg && !rgb_w) || (rgb_dw && !img) || (rgb_db && !rgb_dw)) return PG_E_ARG;
    return pgk::launch_conv_strip_masked_rgb_bwd(gz, wt, mask_bytes, mask_slope, y, rgb_w, rgb_scale, gimg, img, rgb_dw, rgb_db,
                                                 N, C, H, W, Cin, Cout, scale, (hipStream_t)stream, g_last_kernel, sizeof(g_last_kernel));
}

extern "C" const char* pg_debug_last_conv_kernel(void) { return g_last_kernel; }
#ifdef PG_WINO_TRACE
extern "C" int pg_debug_wgrad_trace(void* buf) { g_wgrad_trace = (unsigned long long*)buf; return 0; }
#endif

extern "C" int pg_debug_set_tuning(int key, int value)
{
    if (key < 0 || key >= 4) return PG_E_ARG;
    g_tune[key] = value;
    return 0;
}

extern "C" int pg_pack_dgrad_weights_batched(const float* wbase, float* wtbase, int nlayers, const int64_t* off,
                                             const int* ks, const int* cout, const int* cin, pg_stream_t stream)
{
    if (!wbase || !wtbase || nlayers <= 0 || !off || !ks || !cout || !cin) return PG_E_ARG;
    for (int l0 = 0; l0 < nlayers; l0 += PACK_MAX_LAYERS) {
        PackDesc d;
        d.n = nlayers - l0 < PACK_MAX_LAYERS ? nlayers - l0 : PACK_MAX_LAYERS;
        int total = 0;
        for (int l = 0; l < d.n; ++l) {
            const int i = l0 + l;
            if (ks[i] <= 0 || cout[i] <= 0 || cin[i] <= 0 || off[i] < 0) return PG_E_ARG;
            d.first_block[l] = total;
            d.off[l] = off[i]; d.ks[l] = ks[i]; d.cout[l] = cout[i]; d.cin[l] = cin[i];
            total += ((cin[i] + 31) / 32) * ((cout[i] + 31) / 32) * ks[i] * ks[i];
        }
        d.first_block[d.n] = total;
        hipLaunchKernelGGL(pack_dgrad_batched_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, wbase, wtbase, d);
    }
    return (int)hipGetLastError();
}

extern "C" int pg_pack_dgrad_weights(const float* w, float* wt, int KS, int Cout, int Cin, pg_stream_t stream)
{
    if (!w || !wt || KS <= 0 || Cout <= 0 || Cin <= 0) return PG_E_ARG;
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, KS * KS);
    hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, KS, Cout, Cin);
    return (int)hipGetLastError();
}
