// Winograd F(2x2, 3x3) WEIGHT GRADIENT for the wide 3x3 layers, fp32 on v_mfma_f32_16x16x4_f32.
//
//   forward   Y  = A^T [ (G g G^T) (.) (B^T d B) ] A
//   gradient  dg = G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G
//
// so the 16 Winograd-domain products M[xi][co][ci] = sum over 2x2 output tiles of Z[xi][tile][co] * V[xi][tile][ci] are
// sixteen GEMMs with K = tiles: 16 MFMAs per 4 tiles (16 pixels) instead of 36 for the direct formulation.  A workgroup
// owns a 16*NCO(cout) x 16*NCI(cin) block of dW: NCO x NCI waves of 16 x 16 each; with 32 x 32 blocks all four waves walk
// the same tiles (no reduction), the 16-channel layers (NCO or NCI = 1) split the tiles of a region over the spare waves
// instead and every wave commits its own partial sum.  The workgroup
// stages a region of 32 tiles (16x8 pixels: gz rows + x halo rows) in LDS, every lane transforms ITS (tile, channel)
// values in registers: Z from the 2x2 gz values (A operand: row = cout, k = tile), V from the 4x4 x patch (B operand:
// col = cin, k = tile).  M stays in 16 accumulators; G^T M G is lane-local at the end and commits 9 taps with atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pggan_hip.h"
#include "bufload.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#ifndef PG_WW_SCHED
#define PG_WW_SCHED 2            // 3: both cout halves' transforms first, then one stream of 32 MFMAs (A/B builds)
#endif
#ifndef PG_WW_SCHED_NP
#define PG_WW_SCHED_NP 1            // the same barriers in the 16 x 16 / 16 x 32 / 32 x 16 forms: +6 ... 9 % alone (140 -> 149 TF at n12 @512 16->32), neutral in the step
#endif

namespace {

struct WWP {
    const float* x; const float* gz; float* dw; float* db;
    // optional second batch (x2, gz2) of the SAME layer summed into the same launch: the regions of both batches form one
    // sequence (first batch: regions [0, nregions1)), so dW is committed once per workgroup for both (the commit is 35 % of a
    // small-batch launch).  db_batches: bit 0 / 1 = the first / second batch contributes to db.
    const float* x2; const float* gz2;
    int nregions1, db_batches;
    int N, H, W, Cin, Cout, ups;
    float scale;
    int blocksW, blocksH, nregions, regions_per_block;     // region = one image x 4 x 8 tiles (8 x 16 pixels)
    int lgBW, lgBH;                                         // log2 of blocksW / blocksH (H, W are powers of two)
#ifdef PG_WINO_TRACE
    unsigned long long* trace;                             // [workgroup][wave][region < 8][8] s_memtime stamps (tools/exp/wgrad_trace.py)
#endif
};

#ifdef PG_WINO_TRACE
#define PG_RSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (p.trace && lane == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 1024 && (region - r_begin) < 8) \
    p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + (region - r_begin)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
thread_local unsigned long long* g_ww_trace = nullptr;
#else
#define PG_RSTAMP(i) do { } while (0)
#endif

constexpr int RTW = 8, RTH = 4;                // tiles per region: 8 wide x 4 high = 32 tiles = 16 x 8 pixels
constexpr int PW = 2 * RTW, PH = 2 * RTH;      // 16 x 8 output pixels
constexpr int HW_ = PW + 2, HH_ = PH + 2;      // 18 x 10 input halo pixels
// LDS pixel stride (floats): the 4 tiles of a k-step (2 px apart) land 16 banks apart -- 40 for 32 channels, 24 for 16
constexpr int stride_for(int nw) { return nw == 2 ? 40 : 24; }

// The two operand transforms of a k-step, with the signs arranged so that neither needs a negation: Z' = S Z S and V' = S V S for
// S = diag(1, 1, 1, -1) have the same element-wise product as Z = A dY A^T and V = B^T d B, Z' is sums and differences of the four
// gz values only, and V' absorbs the signs by swapping the operands of the row-3 / column-3 subtractions.  (The ISA of the first
// form had 8 v_xor sign flips per 32 MFMAs; VALU cycles and MFMA cycles of co-resident waves add on gfx950,
// profiles/r04_mfma_valu_share.txt.)  12 VALU for Z', 32 for V'.
// dW commit: buffer_atomic_add_f32 with the whole offset in the VGPR (the flat atomicAdd cost a v_mad_u64_u32 per address, 72 per lane
// of the pair kernel; a register soffset is avoided for the reason given at st4 in conv_wino_strip.hip)
__device__ __forceinline__ void pg_atomic_add(__amdgpu_buffer_rsrc_t r, unsigned voff, float v)
{
    __builtin_amdgcn_raw_ptr_buffer_atomic_fadd_f32(v, r, (int)voff, 0, 0);
}
__device__ __forceinline__ void wino_zprime(const float* gp, int stride, int row, float (&z)[16])
{
    const float y00 = gp[0], y01 = gp[stride], y10 = gp[row], y11 = gp[row + stride];
    const float c1a = y00 + y10, c1b = y01 + y11, c2a = y00 - y10, c2b = y01 - y11;
    z[0] = y00; z[1] = y00 + y01; z[2] = y00 - y01; z[3] = y01;
    z[4] = c1a; z[5] = c1a + c1b; z[6] = c1a - c1b; z[7] = c1b;            // z[5] = y00 + y01 + y10 + y11: the tile's bias sum
    z[8] = c2a; z[9] = c2a + c2b; z[10] = c2a - c2b; z[11] = c2b;
    z[12] = y10; z[13] = y10 + y11; z[14] = y10 - y11; z[15] = y11;
}
__device__ __forceinline__ void wino_vprime(const float* xp, int stride, int row, float (&d)[4][4])
{
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[a][c] = xp[a * row + c * stride];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[3][c] - d[1][c];
        d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float t0 = d[a][0] - d[a][2], t1 = d[a][1] + d[a][2], t2 = d[a][2] - d[a][1], t3 = d[a][3] - d[a][1];
        d[a][0] = t0; d[a][1] = t1; d[a][2] = t2; d[a][3] = t3;
    }
}

template <int NCO, int NCI>
__global__ __launch_bounds__(256) void conv_wino_wgrad_kernel(WWP p)
{
    constexpr int SZ = stride_for(NCO), SX = stride_for(NCI);
    constexpr int KSPL = 4 / (NCO * NCI);                  // waves sharing one 16 x 16 block: they split the k-steps
    constexpr int VZ = 4 * NCO, VX = 4 * NCI;              // float4 per pixel
    __shared__ __align__(16) float smem[PH * PW * SZ + HH_ * HW_ * SX];
    float* gzt = smem;                                     // [128 px][16*NCO co]
    float* xt = smem + PH * PW * SZ;                       // [180 px][16*NCI ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int wave_co = wave % NCO, wave_ci = (wave / NCO) % NCI, wk = wave / (NCO * NCI);
    const int co0 = blockIdx.y * (16 * NCO), ci0 = blockIdx.z * (16 * NCI);
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W, xsh = p.ups ? 1 : 0;
    const bool do_bias = p.db != nullptr && blockIdx.z == 0 && wave_ci == 0;
    const __amdgpu_buffer_rsrc_t rdw = pg_make_rsrc(p.dw, 36u * (unsigned)(p.Cout * p.Cin));     // the commit: buffer atomics on 32-bit offsets
    const unsigned tapbytes = 4u * (unsigned)(p.Cout * p.Cin);

    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;

    // load descriptors: gz 128 px x VZ float4, x 180 px x VX float4
    constexpr int ZPT = (PH * PW * VZ) / 256;              // 4 (2)
    constexpr int XPT = (HH_ * HW_ * VX + 255) / 256;      // 6 (3)
    // per-thread load descriptors (region-invariant part), raw buffer loads per image (bufload.h): PG_OOB = zero fill
    unsigned zoff[ZPT];
    int xpy[XPT], xpx[XPT], xcv[XPT];
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VZ, v = idx % VZ;
        const int py = q / PW, px = q - py * PW;
        zoff[i] = co0 + 4 * v < p.Cout ? 4u * (unsigned)((py * p.W + px) * p.Cout + 4 * v) : PG_OOB;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VX, v = idx % VX;
        const int py = q / HW_, px = q - py * HW_;
        const bool live = q < HH_ * HW_ && ci0 + 4 * v < p.Cin;
        xpy[i] = live ? py - 1 : -(1 << 20);                // a row far outside every image: the bounds test below fails
        xpx[i] = px - 1;
        // byte offset of the lane's pixel relative to the region's first pixel (negative for the halo row / column above / left of it).
        // Region origins are even, so (oy0 + py - 1) >> ups == (oy0 >> ups) + ((py - 1) >> ups) with an arithmetic shift: the per-region
        // part of the address is one scalar, and no multiplication is left in the per-region code (it compiled to v_mad_u64_u32 +
        // v_mul_lo_u32 behind two divergent branches per load)
        xcv[i] = 4 * ((((py - 1) >> xsh) * xW + ((px - 1) >> xsh)) * p.Cin + 4 * v);
    }
    const unsigned zimg = (unsigned)((size_t)p.H * p.W * p.Cout * 4), ximg = (unsigned)((size_t)xH * xW * p.Cin * 4);
    float4 zreg[ZPT], xreg[XPT];
    auto fetch = [&](int region) {
        const bool second = region >= p.nregions1;
        int r = second ? region - p.nregions1 : region;
        const float* xb = second ? p.x2 : p.x;
        const float* gb = second ? p.gz2 : p.gz;
        const int bw = r & (p.blocksW - 1); r >>= p.lgBW;          // (powers of two: no runtime divisions per region)
        const int bh = r & (p.blocksH - 1); const int n = r >> p.lgBH;
        const int oy0 = bh * PH, ox0 = bw * PW;
        const __amdgpu_buffer_rsrc_t rz = pg_make_rsrc(gb + (size_t)n * p.H * p.W * p.Cout + co0, zimg - 4u * (unsigned)co0);
        const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(xb + (size_t)n * xH * xW * p.Cin + ci0, ximg - 4u * (unsigned)ci0);
        const unsigned zbase = 4u * (unsigned)((oy0 * p.W + ox0) * p.Cout);
#pragma unroll
        for (int i = 0; i < ZPT; ++i) zreg[i] = pg_buf_load4(rz, zoff[i], zbase);
        const int xbase = 4 * (((oy0 >> xsh) * xW + (ox0 >> xsh)) * p.Cin);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bool ok = (unsigned)(oy0 + xpy[i]) < (unsigned)p.H && (unsigned)(ox0 + xpx[i]) < (unsigned)p.W;
            xreg[i] = pg_buf_load4(rx, ok ? (unsigned)(xbase + xcv[i]) : PG_OOB, 0);
        }
    };
    const int r_begin = (int)pg_xcd_remap(blockIdx.x, gridDim.x) * p.regions_per_block;   // neighbouring region ranges on one XCD
    const int r_end = min(r_begin + p.regions_per_block, p.nregions);
    if (r_begin < r_end) fetch(r_begin);
    for (int region = r_begin; region < r_end; ++region) {
        PG_RSTAMP(0);
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(gzt + (idx / VZ) * SZ + 4 * (idx % VZ)) = zreg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + 256 * i;
            if (idx / VX < HH_ * HW_) *reinterpret_cast<float4*>(xt + (idx / VX) * SX + 4 * (idx % VX)) = xreg[i];
        }
        PG_RSTAMP(1);
        __syncthreads();
        PG_RSTAMP(2);
        if (region + 1 < r_end) fetch(region + 1);
        const bool bias_here = (p.db_batches >> (region >= p.nregions1 ? 1 : 0)) & 1;      // (workgroup-uniform)
        PG_RSTAMP(3);
        // 32 tiles = 8 k-steps of 4 tiles; lane (li, kk): tile 4*step + kk, A channel co = wave_co*16 + li, B channel ci = wave_ci*16 + li.
        // step = wk + KSPL s (wave-uniform, unrolled): tile row step >> 1, tile column 4 (step & 1) + kk -- one base per lane + immediates
        const float* gq = gzt + (2 * kk) * SZ + wave_co * 16 + li;
        const float* xq = xt + (2 * kk) * SX + wave_ci * 16 + li;
#pragma unroll
        for (int s_ = 0; s_ < 8 / KSPL; ++s_) {
            const int step = wk + KSPL * s_;
            const int tty = step >> 1, tx0 = 4 * (step & 1);
            float z[16], d[4][4];
            wino_zprime(gq + ((2 * tty) * PW + 2 * tx0) * SZ, SZ, PW * SZ, z);
            if (bias_here) bsum += z[5];
            wino_vprime(xq + ((2 * tty) * HW_ + 2 * tx0) * SX, SX, HW_ * SX, d);
#if PG_WW_SCHED_NP
            __builtin_amdgcn_sched_barrier(0);                // (as in the pair kernel below)
#endif
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) acc[xi] = MFMA16(z[xi], d[xi >> 2][xi & 3], acc[xi]);
#if PG_WW_SCHED_NP
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        PG_RSTAMP(4);
        __syncthreads();
        PG_RSTAMP(5);
    }
    // dg = G^T M G, lane-local: acc[xi][r] is M[xi] for cout co0 + wave_co*16 + 4*kk + r, cin ci0 + wave_ci*16 + li
    float g[4][9];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float m[4][4];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) m[xi >> 2][xi & 3] = acc[xi][r];
        float t[3][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                         // G^T = [[1,.5,.5,0],[0,.5,-.5,0],[0,.5,.5,1]]
            const float s = 0.5f * (m[1][j] + m[2][j]), dlt = 0.5f * (m[1][j] - m[2][j]);
            t[0][j] = m[0][j] + s; t[1][j] = dlt; t[2][j] = s + m[3][j];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float s = 0.5f * (t[a][1] + t[a][2]), dlt = 0.5f * (t[a][1] - t[a][2]);
            g[r][3 * a] = t[a][0] + s; g[r][3 * a + 1] = dlt; g[r][3 * a + 2] = s + t[a][3];
        }
    }
    if (KSPL > 1) {                                           // waves that split the tiles of one block: fold them in LDS, wave wk = 0 commits
        static_assert((KSPL - 1) * NCO * NCI * 36 * 64 <= PH * PW * SZ + HH_ * HW_ * SX, "reduction slots reuse the tile buffers");
        const int blk = wave % (NCO * NCI);
        if (wk > 0) {
            float* dst = smem + ((wk - 1) * (NCO * NCI) + blk) * (36 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 9; ++v) dst[(9 * r + v) * 64] = g[r][v];
        }
        __syncthreads();
        if (wk > 0) {
            if (do_bias) {                                    // the bias partial still goes out from every wave
                bsum += __shfl_xor(bsum, 16, 64);
                bsum += __shfl_xor(bsum, 32, 64);
                const int co = co0 + wave_co * 16 + li;
                if (kk == 0 && co < p.Cout) atomicAdd(p.db + co, bsum);
            }
            return;
        }
#pragma unroll
        for (int k = 1; k < KSPL; ++k) {
            const float* src = smem + ((k - 1) * (NCO * NCI) + blk) * (36 * 64) + lane;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 9; ++v) g[r][v] += src[(9 * r + v) * 64];
        }
    }
    const int ci = ci0 + wave_ci * 16 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = co0 + wave_co * 16 + 4 * kk + r;
        if (co < p.Cout && ci < p.Cin) {
            const unsigned off = 4u * (unsigned)(co * p.Cin + ci);
#pragma unroll
            for (int v = 0; v < 9; ++v) pg_atomic_add(rdw, off + (unsigned)v * tapbytes, g[r][v] * p.scale);
        }
    }
    if (do_bias) {                                            // lane (li = co, kk): sum over its tiles; fold the 4 kk lanes
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        const int co = co0 + wave_co * 16 + li;
        if (kk == 0 && co < p.Cout) atomicAdd(p.db + co, bsum);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 32 x 32 blocks, second mapping: a wave owns BOTH cout halves of one cin half (two 16 x 16 accumulator sets) and the two wave
// pairs split the k-steps of a region.  Per 32 MFMAs a wave then transforms Z for 32 couts (24 VALU, 8 reads) and V for 16 cins
// (32 VALU, 16 reads): 56 VALU + 24 LDS reads per 32 MFMAs instead of 2 x (44 + 20) -- the phase trace put the k-steps of the
// first mapping at 1460 cycles per 16 MFMAs per wave (512 of MFMA issue), i.e. bound by the transform and fragment traffic every
// wave repeats.  The wave pairs fold their sums through LDS at the end (as the 16-channel variants do).
__global__ __launch_bounds__(256, 2) void conv_wino_wgrad_pair_kernel(WWP p)
{
    constexpr int SZ = stride_for(2), SX = stride_for(2);
    constexpr int VZ = 8, VX = 8;                          // float4 per pixel
    __shared__ __align__(16) float smem[PH * PW * SZ + HH_ * HW_ * SX];
    float* gzt = smem;                                     // [128 px][32 co]
    float* xt = smem + PH * PW * SZ;                       // [180 px][32 ci]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;
    const int wave_ci = wave & 1, wk = wave >> 1;
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.z * 32;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W, xsh = p.ups ? 1 : 0;
    const bool do_bias = p.db != nullptr && blockIdx.z == 0 && wave_ci == 0;
    const __amdgpu_buffer_rsrc_t rdw = pg_make_rsrc(p.dw, 36u * (unsigned)(p.Cout * p.Cin));     // the commit: buffer atomics on 32-bit offsets
    const unsigned tapbytes = 4u * (unsigned)(p.Cout * p.Cin);

    f32x4 acc[2][16];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[c][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};

    constexpr int ZPT = (PH * PW * VZ) / 256;              // 4
    constexpr int XPT = (HH_ * HW_ * VX + 255) / 256;      // 6
    unsigned zoff[ZPT];
    int xpy[XPT], xpx[XPT], xcv[XPT];
#pragma unroll
    for (int i = 0; i < ZPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VZ, v = idx % VZ;
        const int py = q / PW, px = q - py * PW;
        zoff[i] = co0 + 4 * v < p.Cout ? 4u * (unsigned)((py * p.W + px) * p.Cout + 4 * v) : PG_OOB;
    }
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VX, v = idx % VX;
        const int py = q / HW_, px = q - py * HW_;
        const bool live = q < HH_ * HW_ && ci0 + 4 * v < p.Cin;
        xpy[i] = live ? py - 1 : -(1 << 20);
        xpx[i] = px - 1;
        xcv[i] = 4 * ((((py - 1) >> xsh) * xW + ((px - 1) >> xsh)) * p.Cin + 4 * v);      // (see conv_wino_wgrad_kernel)
    }
    const unsigned zimg = (unsigned)((size_t)p.H * p.W * p.Cout * 4), ximg = (unsigned)((size_t)xH * xW * p.Cin * 4);
    float4 zreg[ZPT], xreg[XPT];
    auto fetch = [&](int region) {
        const bool second = region >= p.nregions1;
        int r = second ? region - p.nregions1 : region;
        const float* xb = second ? p.x2 : p.x;
        const float* gb = second ? p.gz2 : p.gz;
        const int bw = r & (p.blocksW - 1); r >>= p.lgBW;          // (powers of two: no runtime divisions per region)
        const int bh = r & (p.blocksH - 1); const int n = r >> p.lgBH;
        const int oy0 = bh * PH, ox0 = bw * PW;
        const __amdgpu_buffer_rsrc_t rz = pg_make_rsrc(gb + (size_t)n * p.H * p.W * p.Cout + co0, zimg - 4u * (unsigned)co0);
        const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(xb + (size_t)n * xH * xW * p.Cin + ci0, ximg - 4u * (unsigned)ci0);
        const unsigned zbase = 4u * (unsigned)((oy0 * p.W + ox0) * p.Cout);
#pragma unroll
        for (int i = 0; i < ZPT; ++i) zreg[i] = pg_buf_load4(rz, zoff[i], zbase);
        const int xbase = 4 * (((oy0 >> xsh) * xW + (ox0 >> xsh)) * p.Cin);
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const bool ok = (unsigned)(oy0 + xpy[i]) < (unsigned)p.H && (unsigned)(ox0 + xpx[i]) < (unsigned)p.W;
            xreg[i] = pg_buf_load4(rx, ok ? (unsigned)(xbase + xcv[i]) : PG_OOB, 0);
        }
    };
    const int r_begin = (int)pg_xcd_remap(blockIdx.x, gridDim.x) * p.regions_per_block;
    const int r_end = min(r_begin + p.regions_per_block, p.nregions);
    if (r_begin < r_end) fetch(r_begin);
    for (int region = r_begin; region < r_end; ++region) {
#pragma unroll
        for (int i = 0; i < ZPT; ++i) {
            const int idx = tid + 256 * i;
            *reinterpret_cast<float4*>(gzt + (idx / VZ) * SZ + 4 * (idx % VZ)) = zreg[i];
        }
#pragma unroll
        for (int i = 0; i < XPT; ++i) {
            const int idx = tid + 256 * i;
            if (idx / VX < HH_ * HW_) *reinterpret_cast<float4*>(xt + (idx / VX) * SX + 4 * (idx % VX)) = xreg[i];
        }
        __syncthreads();
        if (region + 1 < r_end) fetch(region + 1);
        const bool bias_here = (p.db_batches >> (region >= p.nregions1 ? 1 : 0)) & 1;
        // step = wk + 2 s (wave-uniform, unrolled): tile row s, tile column 4 wk + kk -- one base per lane + immediates
        const float* gq = gzt + (2 * (4 * wk + kk)) * SZ + li;
        const float* xq = xt + (2 * (4 * wk + kk)) * SX + wave_ci * 16 + li;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            float d[4][4];
            wino_vprime(xq + (2 * s_) * HW_ * SX, SX, HW_ * SX, d);          // B operand, shared by both cout halves
#if PG_WW_SCHED == 3
            float z2[2][16];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                wino_zprime(gq + (2 * s_) * PW * SZ + h * 16, SZ, PW * SZ, z2[h]);
                if (bias_here) bsum[h] += z2[h][5];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc[h][xi] = MFMA16(z2[h][xi], d[xi >> 2][xi & 3], acc[h][xi]);
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float z[16];
                wino_zprime(gq + (2 * s_) * PW * SZ + h * 16, SZ, PW * SZ, z);
                if (bias_here) bsum[h] += z[5];
                // Scheduling barriers around the MFMAs of a k-step: without them hipcc threads the transforms' additions through the MFMA
                // stream and pays an s_nop for every VALU result an MFMA reads at once (24 in the loop, 1 with the barriers): -5.3 % over
                // the layer set alone (151 -> 164 TF algorithmic at n12 @32 256->512), -0.07 ms per 1024^2 step (round 5)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int xi = 0; xi < 16; ++xi) acc[h][xi] = MFMA16(z[xi], d[xi >> 2][xi & 3], acc[h][xi]);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        __syncthreads();
    }
    // dg = G^T M G, lane-local; the second wave pair hands its sums over through LDS (the tile buffers are free now)
    static_assert(2 * 2 * 36 * 64 <= PH * PW * SZ + HH_ * HW_ * SX, "reduction slots reuse the tile buffers");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float g[4][9];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float m[4][4];
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) m[xi >> 2][xi & 3] = acc[h][xi][r];
            float t[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float s2 = 0.5f * (m[1][j] + m[2][j]), dlt = 0.5f * (m[1][j] - m[2][j]);
                t[0][j] = m[0][j] + s2; t[1][j] = dlt; t[2][j] = s2 + m[3][j];
            }
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float s2 = 0.5f * (t[a][1] + t[a][2]), dlt = 0.5f * (t[a][1] - t[a][2]);
                g[r][3 * a] = t[a][0] + s2; g[r][3 * a + 1] = dlt; g[r][3 * a + 2] = s2 + t[a][3];
            }
        }
        float* slot = smem + ((wave_ci * 2 + h) * 36) * 64 + lane;
        if (wk == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int v = 0; v < 9; ++v) slot[(9 * r + v) * 64] = g[r][v];
        }
        __syncthreads();
        if (wk == 0) {
            const int ci = ci0 + wave_ci * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + h * 16 + 4 * kk + r;
                if (co < p.Cout && ci < p.Cin) {
                    const unsigned off = 4u * (unsigned)(co * p.Cin + ci);
#pragma unroll
                    for (int v = 0; v < 9; ++v) pg_atomic_add(rdw, off + (unsigned)v * tapbytes, (g[r][v] + slot[(9 * r + v) * 64]) * p.scale);
                }
            }
        }
        __syncthreads();                                  // (the slots are reused by nobody, but keep the two halves' barriers paired)
    }
    if (do_bias) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float b = bsum[h];
            b += __shfl_xor(b, 16, 64);
            b += __shfl_xor(b, 32, 64);
            const int co = co0 + h * 16 + li;
            if (kk == 0 && co < p.Cout) atomicAdd(p.db + co, b);
        }
    }
}


thread_local char g_ww_last[64] = "";
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

}  // namespace

extern "C" const char* pg_debug_last_wino_wgrad_kernel(void) { return g_ww_last; }
#ifdef PG_WINO_TRACE
extern "C" int pg_debug_wino_wgrad_trace(void* buf) { g_ww_trace = (unsigned long long*)buf; return 0; }
#endif

extern "C" int pg_conv2d_wgrad_wino2_nhwc(const float* x, const float* gz, int N, const float* x2, const float* gz2, int N2,
                                          float* dw, float* db, int db_batches,
                                          int H, int W, int Cin, int Cout, int ups, float scale, pg_stream_t stream)
{
    if (!x || !gz || !dw || N <= 0 || N2 < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if (N2 > 0 && (!x2 || !gz2)) return PG_E_ARG;
    if (db_batches & ~3) return PG_E_ARG;
    if (!db) db_batches = 0;
    if (N2 > 0 && ((long long)N2 * H * W * Cin >= (1ll << 31) || (long long)N2 * H * W * Cout >= (1ll << 31))) return PG_E_UNSUP;
    if ((Cin & 3) || (Cout & 3)) return PG_E_ALIGN;
    if (!pow2(H) || !pow2(W) || H < PH || W < PW) return PG_E_UNSUP;
    if (ups && ((H | W) & 1)) return PG_E_ARG;
    if ((long long)N * H * W * Cin >= (1ll << 31) || (long long)N * H * W * Cout >= (1ll << 31)) return PG_E_UNSUP;
    if ((long long)H * W * Cin * 4 >= (1ll << 31) || (long long)H * W * Cout * 4 >= (1ll << 31)) return PG_E_UNSUP;      // 32-bit buffer offsets per image
    if ((long long)36 * Cout * Cin >= (1ll << 31)) return PG_E_UNSUP;                                                   // ... and into dW
    WWP p;
    p.x = x; p.gz = gz; p.dw = dw; p.db = db_batches ? db : nullptr;
    p.x2 = N2 > 0 ? x2 : x; p.gz2 = N2 > 0 ? gz2 : gz; p.db_batches = db_batches;
    p.N = N + N2; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ups = ups; p.scale = scale;
    p.blocksW = W / PW; p.blocksH = H / PH;
    p.lgBW = 0; while ((1 << p.lgBW) < p.blocksW) ++p.lgBW;
    p.lgBH = 0; while ((1 << p.lgBH) < p.blocksH) ++p.lgBH;
    p.nregions1 = N * p.blocksW * p.blocksH;
    p.nregions = (N + N2) * p.blocksW * p.blocksH;
#ifdef PG_WINO_TRACE
    p.trace = g_ww_trace;
#endif
    const int nco = Cout <= 16 ? 1 : 2, nci = Cin <= 16 ? 1 : 2;
    const int gy = (Cout + 16 * nco - 1) / (16 * nco), gz_ = (Cin + 16 * nci - 1) / (16 * nci);
    // ~384-512 workgroups: best of a 256/384/512/1024 sweep (tools/sweeps/sweep_wino_wgrad.py); more workgroups pay for
    // themselves in the per-workgroup G^T M G commit (9216 atomics each), fewer leave CUs idle.  Round 2 measured the commit by
    // leaving it out: 5 % of the launch on n9 @64 128->256 (141 -> 134 us) but 35 % on n3 @128 64->64 (39 -> 25 us), where 128
    // workgroups add to the same 36 K addresses; the phase trace (tools/exp/wgrad_trace.py ... wino) puts a region at 14.1 k
    // cycles, 11.7 k of them the 8 k-steps (20 ds_read_b32 + ~50 VALU + 16 MFMAs each: 1460 cycles per step at two waves per
    // SIMD, 512 of MFMA issue), 2.4 k staging: the kernel is bound by its compute phase and its commit, not by staging.
    // (round 4, inside the train step where the kernel shares the CUs with the main stream's convs; ms per step, 384 | 512: 1024^2 stage (launches
    //  of 3 + 9 images) 10.573 / 10.573 | 10.607 / 10.617; 512^2 and 256^2 stages equal; 128^2 stage (16 + 48 images) 21.17 / 21.16 | 20.97 / 21.01;
    //  256 and 768 lose at 1024^2: 10.66 / 10.75.  Alone on the device 512 is 1-2 % faster than 384.)
    static const int target_env = [] { const char* t = getenv("PG_WW_TARGET"); return t ? atoi(t) : 0; }();
    static const int target3_env = [] { const char* t = getenv("PG_WW_TARGET3"); return t ? atoi(t) : 0; }();        // launches of <= 4 images (the G step's): sweeps
    const int target = (target3_env > 0 && N + N2 <= 4) ? target3_env : target_env > 0 ? target_env : (N + N2 <= 12 ? 320 : N + N2 <= 24 ? 384 : 512);
    // (round 5, after the k-step got faster; same-box pairs: 1024^2 stage 320 | 384: 10.286 | 10.308 ms over five pairs, 288 equal, 256 +0.08;
    //  512^2 stage (launches of 6 and 24 images) 384 | 512: 13.44 | 13.54; 256^2 stage (14 and 56 images) 512 | 384 | 768: 23.0 | 23.1 | 23.1)
    int chunks = (target + gy * gz_ - 1) / (gy * gz_);
    if (chunks > p.nregions) chunks = p.nregions;
    if (chunks < 1) chunks = 1;
    p.regions_per_block = (p.nregions + chunks - 1) / chunks;
    chunks = (p.nregions + p.regions_per_block - 1) / p.regions_per_block;
    snprintf(g_ww_last, sizeof(g_ww_last), "conv_wino_wgrad_kernel<%d, %d>", nco, nci);
    const dim3 grid(chunks, gy, gz_);
    // measured (tools/sweeps/sweep_wino_wgrad.py, PG_WW_PAIR=0/1/2): the pair mapping wins 8-9 % from ~9 regions per workgroup on
    // (n9 @64 128->256: 137 -> 126 us) and loses up to 15 % at 3 (its extra fold through LDS); 1: built-in choice, 0 / 2: never / always
    static const int pair_env = getenv("PG_WW_PAIR") ? atoi(getenv("PG_WW_PAIR")) : 1;
    if (nco == 2 && nci == 2 && (pair_env == 2 || (pair_env == 1 && p.regions_per_block >= 6))) {
        snprintf(g_ww_last, sizeof(g_ww_last), "conv_wino_wgrad_pair_kernel");
        hipLaunchKernelGGL(conv_wino_wgrad_pair_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
    }
    else if (nco == 2 && nci == 2) hipLaunchKernelGGL((conv_wino_wgrad_kernel<2, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (nco == 1 && nci == 2) hipLaunchKernelGGL((conv_wino_wgrad_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else if (nco == 2 && nci == 1) hipLaunchKernelGGL((conv_wino_wgrad_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv_wino_wgrad_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
    return (int)hipGetLastError();
}

extern "C" int pg_conv2d_wgrad_wino_nhwc(const float* x, const float* gz, float* dw, float* db,
                                         int N, int H, int W, int Cin, int Cout, int ups, float scale, pg_stream_t stream)
{
    return pg_conv2d_wgrad_wino2_nhwc(x, gz, N, nullptr, nullptr, 0, dw, db, 1, H, W, Cin, Cout, ups, scale, stream);
}
