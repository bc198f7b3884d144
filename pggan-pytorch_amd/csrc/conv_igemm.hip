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
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pggan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

template <int VEC> __device__ __forceinline__ void lds_load(const float* p, float (&o)[VEC]);
template <> __device__ __forceinline__ void lds_load<4>(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void lds_load<2>(const float* p, float (&o)[2]) {
    float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[1] = v.y;
}
template <> __device__ __forceinline__ void lds_load<1>(const float* p, float (&o)[1]) { o[0] = *p; }

struct ConvP {
    const float* x; const float* w; const float* bias; const float* mask; float* y;
    int N, Hin, Win, Cin, Cout, Hout, Wout, KS, pad, ups;
    float scale, slope, mask_slope;
    int lgTW, lgTH, TN, tilesW, tilesH;
};

// One workgroup (4 waves) computes BCO couts x BPX output pixels; the pixel tile is
// TN images x TH x TW (powers of two) so that the (KS-1)-halo of the input is staged once in LDS
// and every tap is a shifted read of the same tile.
template <int VEC, int WAVES_CO, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvP p)
{
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int BCO = 16 * WM * WAVES_CO;
    constexpr int KC = 4 * VEC, KCP = KC + 4;
    extern __shared__ __align__(16) float lds[];

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
    const int HT = TH + p.KS - 1, WT = TW + p.KS - 1;
    const int taps = p.KS * p.KS;
    float* wt = lds;                          // [taps][BCO][KCP]
    float* xt = lds + taps * BCO * KCP;       // [TN][HT][WT][KCP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_co = wave % WAVES_CO, wave_px = wave / WAVES_CO;
    const int li = lane & 15, kk = lane >> 4;

    int t = blockIdx.x;
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

    f32x4 acc[WM][WN];
#pragma unroll
    for (int m = 0; m < WM; ++m)
#pragma unroll
        for (int n = 0; n < WN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int xH = p.ups ? (p.Hin >> 1) : p.Hin, xW = p.ups ? (p.Win >> 1) : p.Win;
    const int wrows = taps * BCO;
    const int npix = p.TN * HT * WT;

    for (int k0 = 0; k0 < p.Cin; k0 += KC) {
        for (int idx = tid; idx < wrows * VEC; idx += 256) {
            const int r = idx / VEC, v = idx - r * VEC;
            const int tap = r / BCO, col = r - tap * BCO, co = co0 + col;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co < p.Cout)
                val = *reinterpret_cast<const float4*>(p.w + ((size_t)(tap * p.Cout + co) * p.Cin + k0 + 4 * v));
            *reinterpret_cast<float4*>(wt + r * KCP + 4 * v) = val;
        }
        for (int idx = tid; idx < npix * VEC; idx += 256) {
            const int q = idx / VEC, v = idx - q * VEC;
            const int tw = q % WT, r2 = q / WT, th = r2 % HT, tn = r2 / HT;
            const int n = n0 + tn;
            int ih = oh0 + th - p.pad, iw = ow0 + tw - p.pad;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.N && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
                if (p.ups) { ih >>= 1; iw >>= 1; }
                val = *reinterpret_cast<const float4*>(p.x + (((size_t)n * xH + ih) * xW + iw) * p.Cin + k0 + 4 * v);
            }
            *reinterpret_cast<float4*>(xt + q * KCP + 4 * v) = val;
        }
        __syncthreads();
        for (int kh = 0; kh < p.KS; ++kh) {
            for (int kw = 0; kw < p.KS; ++kw) {
                const int tap = kh * p.KS + kw;
                float a[WM][VEC], b[WN][VEC];
#pragma unroll
                for (int m = 0; m < WM; ++m) lds_load<VEC>(wt + tap * BCO * KCP + wbase[m], a[m]);
#pragma unroll
                for (int n = 0; n < WN; ++n) lds_load<VEC>(xt + pixbase[n] + (kh * WT + kw) * KCP, b[n]);
#pragma unroll
                for (int s = 0; s < VEC; ++s)
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n) acc[m][n] = MFMA16(a[m][s], b[n][s], acc[m][n]);
            }
        }
        __syncthreads();
    }

    // epilogue: lane holds couts cb..cb+3 of pixel j
#pragma unroll
    for (int m = 0; m < WM; ++m) {
        const int cb = co0 + (wave_co * WM + m) * 16 + 4 * kk;
        if (cb >= p.Cout) continue;
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + cb);
#pragma unroll
        for (int n = 0; n < WN; ++n) {
            const int j = (wave_px * WN + n) * 16 + li;
            const int tw = j & (TW - 1), th = (j >> p.lgTW) & (TH - 1), tn = j >> (p.lgTW + p.lgTH);
            const int ni = n0 + tn;
            if (ni >= p.N) continue;
            const size_t off = (((size_t)ni * p.Hout + oh0 + th) * p.Wout + ow0 + tw) * p.Cout + cb;
            float4 o;
            o.x = acc[m][n][0] * p.scale; o.y = acc[m][n][1] * p.scale;
            o.z = acc[m][n][2] * p.scale; o.w = acc[m][n][3] * p.scale;
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
}

// ------------------------------------------------------------------------------------------
struct WgP {
    const float* x; const float* gz; float* dw; float* db;
    int N, Hin, Win, Cin, Cout, Hout, Wout, pad, ups;
    float scale;
    int lgTW, lgTH, TN, tilesW, tilesH, ntiles, tiles_per_block;
};

// dW[tap][co][ci] = sum over pixels: A = gz (row i = cout), B = shifted x (col j = cin), the MFMA
// k index runs over PIXELS (4 per instruction).  One workgroup owns a (BCO x BCI) block of every
// tap and a slice of the pixel tiles; partial sums are committed with fp32 atomics.
template <int KS, int WM, int WN, int WAVES_CO, int WAVES_CI>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgP p)
{
    constexpr int WAVES_K = 4 / (WAVES_CO * WAVES_CI);
    constexpr int BCO = 16 * WM * WAVES_CO, BCI = 16 * WN * WAVES_CI;
    constexpr int SZ = BCO + 16, SX = BCI + 16;      // row strides == 16 (mod 32): conflict-free b32 reads
    constexpr int TAPS = KS * KS;
    extern __shared__ __align__(16) float lds[];

    const int TW = 1 << p.lgTW, TH = 1 << p.lgTH;
    const int HT = TH + KS - 1, WT = TW + KS - 1;
    const int TPIX = p.TN << (p.lgTW + p.lgTH);
    float* gzt = lds;                        // [TPIX][SZ]
    float* xt = lds + TPIX * SZ;             // [TN*HT*WT][SX]

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
    constexpr int ZV = BCO / 4, XV = BCI / 4;

    const int t_begin = blockIdx.x * p.tiles_per_block;
    const int t_end = min(t_begin + p.tiles_per_block, p.ntiles);
    for (int tile = t_begin; tile < t_end; ++tile) {
        int t = tile;
        const int tw_i = t % p.tilesW; t /= p.tilesW;
        const int th_i = t % p.tilesH; t /= p.tilesH;
        const int n0 = t * p.TN;
        const int oh0 = th_i << p.lgTH, ow0 = tw_i << p.lgTW;

        for (int idx = tid; idx < TPIX * ZV; idx += 256) {
            const int q = idx / ZV, v = idx - q * ZV;
            const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
            const int n = n0 + tn, co = co0 + 4 * v;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.N && co < p.Cout)
                val = *reinterpret_cast<const float4*>(p.gz + (((size_t)n * p.Hout + oh0 + th) * p.Wout + ow0 + tw) * p.Cout + co);
            *reinterpret_cast<float4*>(gzt + q * SZ + 4 * v) = val;
        }
        for (int idx = tid; idx < npix * XV; idx += 256) {
            const int q = idx / XV, v = idx - q * XV;
            const int tw = q % WT, r2 = q / WT, th = r2 % HT, tn = r2 / HT;
            const int n = n0 + tn, ci = ci0 + 4 * v;
            int ih = oh0 + th - p.pad, iw = ow0 + tw - p.pad;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < p.N && ci < p.Cin && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
                if (p.ups) { ih >>= 1; iw >>= 1; }
                val = *reinterpret_cast<const float4*>(p.x + (((size_t)n * xH + ih) * xW + iw) * p.Cin + ci);
            }
            *reinterpret_cast<float4*>(xt + q * SX + 4 * v) = val;
        }
        __syncthreads();
        const int nsteps = TPIX >> 2;
        for (int step = wave_k; step < nsteps; step += WAVES_K) {
            const int q = 4 * step + kk;
            const int tw = q & (TW - 1), th = (q >> p.lgTW) & (TH - 1), tn = q >> (p.lgTW + p.lgTH);
            float a[WM];
#pragma unroll
            for (int m = 0; m < WM; ++m) a[m] = gzt[q * SZ + (wave_co * WM + m) * 16 + li];
            if (do_bias) {
#pragma unroll
                for (int m = 0; m < WM; ++m) accb[m] = MFMA16(a[m], 1.0f, accb[m]);
            }
            const float* xrow = xt + ((tn * HT + th) * WT + tw) * SX + wave_ci * WN * 16 + li;
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    float b[WN];
#pragma unroll
                    for (int n = 0; n < WN; ++n) b[n] = xrow[(kh * WT + kw) * SX + n * 16];
#pragma unroll
                    for (int m = 0; m < WM; ++m)
#pragma unroll
                        for (int n = 0; n < WN; ++n)
                            acc[kh * KS + kw][m][n] = MFMA16(a[m], b[n], acc[kh * KS + kw][m][n]);
                }
        }
        __syncthreads();
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
                    if (co < p.Cout)
                        atomicAdd(p.dw + ((size_t)(tp * p.Cout + co) * p.Cin + ci), acc[tp][m][n][r] * p.scale);
                }
        }
        if (do_bias && li == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wave_co * WM + m) * 16 + 4 * kk + r;
                if (co < p.Cout) atomicAdd(p.db + co, accb[m][r]);
            }
        }
    }
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

inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

struct TileGeom { int lgTW, lgTH, TN, tilesW, tilesH, ntiles; };

inline TileGeom make_geom(int N, int Hout, int Wout, int BPX)
{
    TileGeom g;
    int TW = Wout < 32 ? Wout : 32; if (TW > BPX) TW = BPX;
    int TH = BPX / TW; if (TH > Hout) TH = Hout;
    g.lgTW = ilog2(TW); g.lgTH = ilog2(TH);
    g.TN = BPX / (TW * TH);
    g.tilesW = Wout / TW; g.tilesH = Hout / TH;
    g.ntiles = ((N + g.TN - 1) / g.TN) * g.tilesH * g.tilesW;
    return g;
}

template <int VEC, int WAVES_CO, int WM, int WN>
int launch_conv(ConvP& p, hipStream_t s)
{
    constexpr int WAVES_PX = 4 / WAVES_CO;
    constexpr int BCO = 16 * WM * WAVES_CO, BPX = 16 * WN * WAVES_PX, KCP = 4 * VEC + 4;
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH;
    const int HT = (1 << g.lgTH) + p.KS - 1, WT = (1 << g.lgTW) + p.KS - 1;
    const size_t smem = (size_t)(p.KS * p.KS * BCO + g.TN * HT * WT) * KCP * sizeof(float);
    if (smem > 160 * 1024) return PG_E_UNSUP;
    auto kern = conv_igemm_kernel<VEC, WAVES_CO, WM, WN>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    dim3 grid(g.ntiles, (p.Cout + BCO - 1) / BCO);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, p);
    return (int)hipGetLastError();
}

template <int VEC>
int dispatch_conv(ConvP& p, hipStream_t s)
{
    const long long M = (long long)p.N * p.Hout * p.Wout;
    if (p.KS == 4 || M <= 64) {                       // few output pixels: weight-streaming shapes
        if (M <= 16) return launch_conv<VEC, 4, 1, 1>(p, s);
        return launch_conv<VEC, 4, 1, 4>(p, s);
    }
    if (p.Cout <= 16) return launch_conv<VEC, 1, 1, 4>(p, s);
    if (p.Cout <= 32) return launch_conv<VEC, 1, 2, 2>(p, s);
    return launch_conv<VEC, 2, 2, 4>(p, s);
}

template <int KS, int WM, int WN, int WAVES_CO, int WAVES_CI>
int launch_wgrad(WgP& p, hipStream_t s)
{
    constexpr int BCO = 16 * WM * WAVES_CO, BCI = 16 * WN * WAVES_CI;
    constexpr int SZ = BCO + 16, SX = BCI + 16;
    const long long M = (long long)p.N * p.Hout * p.Wout;
    const int BPX = (KS == 4 || M <= 32) ? 16 : 64;
    TileGeom g = make_geom(p.N, p.Hout, p.Wout, BPX);
    p.lgTW = g.lgTW; p.lgTH = g.lgTH; p.TN = g.TN; p.tilesW = g.tilesW; p.tilesH = g.tilesH; p.ntiles = g.ntiles;
    const int HT = (1 << g.lgTH) + KS - 1, WT = (1 << g.lgTW) + KS - 1;
    const size_t smem = ((size_t)BPX * SZ + (size_t)g.TN * HT * WT * SX) * sizeof(float);
    if (smem > 160 * 1024) return PG_E_UNSUP;
    const int gy = (p.Cout + BCO - 1) / BCO, gz_ = (p.Cin + BCI - 1) / BCI;
    int chunks = (2048 + gy * gz_ - 1) / (gy * gz_);       // aim for ~2048 workgroups
    if (chunks > g.ntiles) chunks = g.ntiles;
    if (chunks < 1) chunks = 1;
    p.tiles_per_block = (g.ntiles + chunks - 1) / chunks;
    chunks = (g.ntiles + p.tiles_per_block - 1) / p.tiles_per_block;
    auto kern = conv_wgrad_kernel<KS, WM, WN, WAVES_CO, WAVES_CI>;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(kern, dim3(chunks, gy, gz_), dim3(256), smem, s, p);
    return (int)hipGetLastError();
}

template <int KS>
int dispatch_wgrad(WgP& p, hipStream_t s)
{
    if (KS == 4) return launch_wgrad<KS, 1, 1, 2, 2>(p, s);
    if (p.Cout <= 16 && p.Cin <= 16) return launch_wgrad<KS, 1, 1, 1, 1>(p, s);
    if (p.Cout < 64 || p.Cin < 64) return launch_wgrad<KS, (KS == 4 ? 1 : 2), (KS == 4 ? 1 : 2), 1, 1>(p, s);
    return launch_wgrad<KS, (KS == 4 ? 1 : 2), (KS == 4 ? 1 : 2), 2, 2>(p, s);
}

}  // namespace

extern "C" int pg_conv2d_nhwc(const float* x, const float* w, const float* bias, const float* mask, float* y,
                              int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                              float scale, float slope, float mask_slope, pg_stream_t stream)
{
    if (!x || !w || !y || N <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((Cin & 3) || (Cout & 3)) return PG_E_ALIGN;
    if (KS != 1 && KS != 3 && KS != 4) return PG_E_UNSUP;
    ConvP p;
    p.x = x; p.w = w; p.bias = bias; p.mask = mask; p.y = y;
    p.N = N; p.Hin = Hin; p.Win = Win; p.Cin = Cin; p.Cout = Cout; p.KS = KS; p.pad = pad; p.ups = ups;
    p.Hout = Hin + 2 * pad - KS + 1; p.Wout = Win + 2 * pad - KS + 1;
    if (p.Hout <= 0 || p.Wout <= 0 || !is_pow2(p.Hout) || !is_pow2(p.Wout)) return PG_E_UNSUP;
    if (ups && ((Hin | Win) & 1)) return PG_E_ARG;
    p.scale = scale; p.slope = slope; p.mask_slope = mask_slope;
    hipStream_t s = (hipStream_t)stream;
    if ((Cin & 15) == 0) return dispatch_conv<4>(p, s);
    if ((Cin & 7) == 0) return dispatch_conv<2>(p, s);
    return dispatch_conv<1>(p, s);
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
    hipStream_t s = (hipStream_t)stream;
    switch (KS) {
        case 1: return dispatch_wgrad<1>(p, s);
        case 3: return dispatch_wgrad<3>(p, s);
        case 4: return dispatch_wgrad<4>(p, s);
        default: return PG_E_UNSUP;
    }
}

extern "C" int pg_pack_dgrad_weights(const float* w, float* wt, int KS, int Cout, int Cin, pg_stream_t stream)
{
    if (!w || !wt || KS <= 0 || Cout <= 0 || Cin <= 0) return PG_E_ARG;
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, KS * KS);
    hipLaunchKernelGGL(pack_dgrad_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt, KS, Cout, Cin);
    return (int)hipGetLastError();
}
