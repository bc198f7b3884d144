// Winograd F(2x2, 3x3) WEIGHT GRADIENT for the wide 3x3 layers, fp32 on v_mfma_f32_16x16x4_f32.
//
//   forward   Y  = A^T [ (G g G^T) (.) (B^T d B) ] A
//   gradient  dg = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G
//
// so the 16 Winograd-domain products M[xi][co][ci] = sum over 2x2 output tiles of Z[xi][tile][co] * V[xi][tile][ci] are
// sixteen GEMMs with K = tiles: 16 MFMAs per 4 tiles (16 pixels) instead of 36 for the direct formulation.  A workgroup
// owns a 32(cout) x 32(cin) block of dW (2 x 2 waves, 16 x 16 each, all four waves walk the same tiles: no reduction),
// stages a region of 32 tiles (16x8 pixels: gz rows + x halo rows) in LDS, every lane transforms ITS (tile, channel)
// values in registers: Z from the 2x2 gz values (A operand: row = cout, k = tile), V from the 4x4 x patch (B operand:
// col = cin, k = tile).  M stays in 16 accumulators; G^T M G is lane-local at the end and commits 9 taps with atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pggan_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

namespace {

struct WWP {
    const float* x; const float* gz; float* dw; float* db;
    int N, H, W, Cin, Cout, ups;
    float scale;
    int blocksW, blocksH, nregions, regions_per_block;     // region = one image x 4 x 8 tiles (8 x 16 pixels)
};

constexpr int RTW = 8, RTH = 4;                // tiles per region: 8 wide x 4 high = 32 tiles = 16 x 8 pixels
constexpr int PW = 2 * RTW, PH = 2 * RTH;      // 16 x 8 output pixels
constexpr int HW_ = PW + 2, HH_ = PH + 2;      // 18 x 10 input halo pixels
constexpr int SC = 40;                         // LDS pixel stride (floats) for 32 channels: the 4 tiles of a k-step (2 px apart) land 16 banks apart

__global__ __launch_bounds__(256) void conv_wino_wgrad_kernel(WWP p)
{
    __shared__ __align__(16) float gzt[PH * PW * SC];      // [128 px][32 co]
    __shared__ __align__(16) float xt[HH_ * HW_ * SC];     // [180 px][32 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
    const int wave_co = wave & 1, wave_ci = wave >> 1;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;
    const bool do_bias = p.db != nullptr && blockIdx.z == 0 && wave_ci == 0;

    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // load descriptors: gz 128 px x 8 float4, x 180 px x 8 float4
    constexpr int ZPT = (PH * PW * 8) / 256;               // 4
    constexpr int XPT = (HH_ * HW_ * 8 + 255) / 256;       // 6
    float4 zreg[ZPT], xreg[XPT];
    auto fetch = [&](int region) {
        int r = region;
        const int bw = r % p.blocksW; r /= p.blocksW;
        const int bh = r % p.blocksH; const int n = r / p.blocksH;
        const int oy0 = bh * PH, ox0 = bw * PW;
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            const int idx = tid + 256 * i;
            const int q = idx >> 3, v = idx & 7;
            const int py = q / PW, px = q - py * PW;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (co0 + 4 * v < p.Cout)
                val = *reinterpret_cast<const float4*>(p.gz + (((size_t)n * p.H + oy0 + py) * p.W + ox0 + px) * p.Cout + co0 + 4 * v);
            zreg[i] = val;
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + 256 * i;
            const int q = idx >> 3, v = idx & 7;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < HH_ * HW_) {
                const int py = q / HW_, px = q - py * HW_;
                int ih = oy0 + py - 1, iw = ox0 + px - 1;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W && ci0 + 4 * v < p.Cin) {
                    if (p.ups) { ih >>= 1; iw >>= 1; }
                    val = *reinterpret_cast<const float4*>(p.x + (((size_t)n * xH + ih) * xW + iw) * p.Cin + ci0 + 4 * v);
                }
            }
            xreg[i] = val;
        }
    };
    const int r_begin = blockIdx.x * p.regions_per_block;
    const int r_end = min(r_begin + p.regions_per_block, p.nregions);
    if (r_begin < r_end) fetch(r_begin);
    for (int region = r_begin; region < r_end; ++region) {
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(gzt + (idx >> 3) * SC + 4 * (idx & 7)) = zreg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + 256 * i;
            if ((idx >> 3) < HH_ * HW_) *reinterpret_cast<float4*>(xt + (idx >> 3) * SC + 4 * (idx & 7)) = xreg[i];
        }
        __syncthreads();
        if (region + 1 < r_end) fetch(region + 1);
        // 32 tiles = 8 k-steps of 4 tiles; lane (li, kk): tile 4*step + kk, A channel co = wave_co*16 + li, B channel ci = wave_ci*16 + li
#pragma unroll 2
        for (int step = 0; step < 8; ++step) {
            const int t = 4 * step + kk;
            const int ttx = t & (RTW - 1), tty = t >> 3;
            const float* gp = gzt + ((2 * tty) * PW + 2 * ttx) * SC + wave_co * 16 + li;
            const float y00 = gp[0], y01 = gp[SC], y10 = gp[PW * SC], y11 = gp[(PW + 1) * SC];
            bsum += (y00 + y01) + (y10 + y11);
            // Z = A dY A^T,  A = [[1,0],[1,1],[1,-1],[0,-1]]
            const float c0a = y00, c0b = y01;                 // rows of (A dY): r0 = y0., r1 = y0. + y1., r2 = y0. - y1., r3 = -y1.
            const float c1a = y00 + y10, c1b = y01 + y11;
            const float c2a = y00 - y10, c2b = y01 - y11;
            const float c3a = -y10, c3b = -y11;
            float z[16];
            z[0] = c0a; z[1] = c0a + c0b; z[2] = c0a - c0b; z[3] = -c0b;
            z[4] = c1a; z[5] = c1a + c1b; z[6] = c1a - c1b; z[7] = -c1b;
            z[8] = c2a; z[9] = c2a + c2b; z[10] = c2a - c2b; z[11] = -c2b;
            z[12] = c3a; z[13] = c3a + c3b; z[14] = c3a - c3b; z[15] = -c3b;
            // V = B^T d B from the 4x4 patch of x
            const float* xp = xt + ((2 * tty) * HW_ + 2 * ttx) * SC + wave_ci * 16 + li;
            float d[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) d[a][c] = xp[(a * HW_ + c) * SC];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
                d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float t0 = d[a][0] - d[a][2], t1 = d[a][1] + d[a][2], t2 = d[a][2] - d[a][1], t3 = d[a][1] - d[a][3];
                d[a][0] = t0; d[a][1] = t1; d[a][2] = t2; d[a][3] = t3;
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[xi] = MFMA16(z[xi], d[xi >> 2][xi & 3], acc[xi]);
        }
        __syncthreads();
    }
    // dg = G^T M G, lane-local: acc[xi][r] is M[xi] for cout co0 + wave_co*16 + 4*kk + r, cin ci0 + wave_ci*16 + li
    const int ci = ci0 + wave_ci * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + wave_co * 16 + 4 * kk + r;
        float m[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi >> 2][xi & 3] = acc[xi][r];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
            const float s = 0.5f * (m[1][j] + m[2][j]), dlt = 0.5f * (m[1][j] - m[2][j]);
            t[0][j] = m[0][j] + s; t[1][j] = dlt; t[2][j] = s + m[3][j];
        }
        if (co < p.Cout && ci < p.Cin) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float s = 0.5f * (t[a][1] + t[a][2]), dlt = 0.5f * (t[a][1] - t[a][2]);
                const float g0 = t[a][0] + s, g1 = dlt, g2 = s + t[a][3];
                float* dst = p.dw + ((size_t)(a * 3) * p.Cout + co) * p.Cin + ci;
                atomicAdd(dst, g0 * p.scale);
                atomicAdd(dst + (size_t)p.Cout * p.Cin, g1 * p.scale);
                atomicAdd(dst + (size_t)2 * p.Cout * p.Cin, g2 * p.scale);
            }
        }
    }
    if (do_bias) {                                            // lane (li = co, kk): sum over its tiles; fold the 4 kk lanes
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + wave_co * 16 + li;
        if (kk == 0 && co < p.Cout) atomicAdd(p.db + co, bsum);
    }
}

thread_local char g_ww_last[64] = "";
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" const char* pg_debug_last_wino_wgrad_kernel(void) { return g_ww_last; }

extern "C" int pg_conv2d_wgrad_wino_nhwc(const float* x, const float* gz, float* dw, float* db,
                                         int N, int H, int W, int Cin, int Cout, int ups, float scale, pg_stream_t stream)
{
    if (!x || !gz || !dw || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((Cin & 3) || (Cout & 3)) return PG_E_ALIGN;
    if (!pow2(H) || !pow2(W) || H < PH || W < PW) return PG_E_UNSUP;
    if (ups && ((H | W) & 1)) return PG_E_ARG;
    if ((long long)N * H * W * Cin >= (1ll << 31) || (long long)N * H * W * Cout >= (1ll << 31)) return PG_E_UNSUP;
    WWP p;
    p.x = x; p.gz = gz; p.dw = dw; p.db = db;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ups = ups; p.scale = scale;
    p.blocksW = W / PW; p.blocksH = H / PH;
    p.nregions = N * p.blocksW * p.blocksH;
    const int gy = (Cout + 31) / 32, gz_ = (Cin + 31) / 32;
    // ~512 workgroups: best of a 256/384/512/1024 sweep (tools/sweep_wino_wgrad.py); more workgroups pay for
    // themselves in the per-workgroup G^T M G commit (9216 atomics each), fewer leave CUs idle.
    static const int target = [] { const char* t = getenv("PG_WW_TARGET"); return t ? atoi(t) : 512; }();
    int chunks = (target + gy * gz_ - 1) / (gy * gz_);
    if (chunks > p.nregions) chunks = p.nregions;
    if (chunks < 1) chunks = 1;
    p.regions_per_block = (p.nregions + chunks - 1) / chunks;
    chunks = (p.nregions + p.regions_per_block - 1) / p.regions_per_block;
    snprintf(g_ww_last, sizeof(g_ww_last), "conv_wino_wgrad_kernel");
    hipLaunchKernelGGL(conv_wino_wgrad_kernel, dim3(chunks, gy, gz_), dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}
