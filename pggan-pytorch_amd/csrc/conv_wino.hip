// Winograd F(2x2, 3x3) convolution for the wide 3x3 layers (>= 32 input channels), fp32 on v_mfma_f32_16x16x4_f32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A        per 2x2 output tile, 4x4 input patch d, 3x3 filter g
//
// 16 multiplies per 2x2 outputs instead of 36: the sixteen element-wise products become sixteen independent
// [Cout x Cin] x [Cin x tiles] GEMMs, i.e. 2.25x fewer MFMAs than the direct implicit GEMM.  Numerically this is a
// re-association of the same fp32 sums (transform coefficients +-1 and 1/2), well inside the 1e-3 parity bar
// (measured ~1e-6 relative); it replaces F.conv2d network.py:33-36 for those layers exactly like the direct kernel.
//
// Layout: U[16][Cout][Cin] transformed weights (pg_wino_transform_weights, once per weight version); workgroup =
// 16 couts x 64 tiles (4 waves x 16 tiles), K in chunks of 16 channels: raw input halo region and the U chunk in
// LDS, each lane loads ITS tile's 4x4 patch (16 b128 reads), transforms it in registers (32 float4 adds) and feeds
// 64 MFMAs; the output transform is lane-local because a lane holds all 16 products of its (tile, 4 couts).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <mutex>
#include <vector>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pggan_hip.h"
#include "bufload.h"
#include "convp.h"

#include "wino_epi.h"

namespace {

using namespace pgw;

template <int VEC> struct WRow { static constexpr int value = VEC == 4 ? 24 : 12; };   // LDS row stride (floats), conflict-free b128 / b64
constexpr int XMAX = 400;                      // halo pixels per workgroup: 18x18 (8x8 tiles) .. 4 x 10x10 (8x8 images)

__global__ __launch_bounds__(256) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin)
{
    const size_t n = (size_t)Cout * Cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        // thread i -> (pack, cout, channel of the pack): consecutive threads WRITE consecutive floats of every Winograd position
        // (the reads are 32-byte pieces; with i -> (cout, cin) the sixteen stores of a wave were 8 x 32-byte pieces each)
        const size_t pk = i / ((size_t)8 * Cout), rr = i - pk * 8 * Cout;
        const size_t co = rr >> 3, ci = 8 * pk + (rr & 7), src = co * Cin + ci;
        float g[3][3], t[4][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = w[(size_t)(a * 3 + b) * n + src];
#pragma unroll
        for (int b = 0; b < 3; ++b) {                     // t = G g
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
            t[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
            t[3][b] = g[2][b];
        }
        float* ub = u + wino_u_index(0, co, ci, Cout);
        const size_t xs = (size_t)Cout * 8;               // stride between Winograd positions inside a pack
#pragma unroll
        for (int a = 0; a < 4; ++a) {                     // U = t G^T
            ub[(size_t)(4 * a + 0) * xs] = t[a][0];
            ub[(size_t)(4 * a + 1) * xs] = 0.5f * ((t[a][0] + t[a][1]) + t[a][2]);
            ub[(size_t)(4 * a + 2) * xs] = 0.5f * ((t[a][0] - t[a][1]) + t[a][2]);
            ub[(size_t)(4 * a + 3) * xs] = t[a][2];
        }
    }
}


template <int VEC>
__global__ __launch_bounds__(256) void conv_wino_kernel(WinoP p)
{
    constexpr int KC = 4 * VEC, KCP = WRow<VEC>::value;
    constexpr int KXP = VEC == 4 ? 20 : KCP;                // pixel stride of the input region (see the lane -> tile map below)
    constexpr int XPT = (XMAX * VEC + 255) / 256;           // float4 per thread for the halo region (KC channels)
    constexpr int UPT = VEC;                                 // 16 xi x 16 couts x VEC float4 / 256 threads
    typedef float fragv __attribute__((ext_vector_type(VEC)));
    extern __shared__ __align__(16) float lds[];
    float* ut = lds;                                         // [16 xi][16 co][KCP]
    float* xt = lds + 16 * 16 * KCP;                         // [TN][HT][WT][KXP]
    const int TTW = 1 << p.lgTW, TTH = 1 << p.lgTH;
    const int WT = 2 * TTW + 2, HT = 2 * TTH + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
#ifdef PG_WINO_TRACE
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + 0) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
    // 1-D grid, cout blocks minor: after the XCD remap (bufload.h) the workgroups that share an input region (same tiles,
    // different couts) run back to back on ONE XCD, so the region comes from HBM once and from that L2 afterwards; each XCD
    // walks a contiguous range of tile blocks (halo rows / columns shared with the neighbours stay in its L2 too).
    // Layers whose Winograd weights do not fit an L2 (> 2 MB: the 256/512-channel layers) keep the tile-minor order
    // instead, where all concurrently running workgroups share one 16-cout slice of U.
    int b = (int)pg_xcd_remap(blockIdx.x, gridDim.x);
    int cob;
    if (p.cout_minor) { cob = b % p.ncob; b /= p.ncob; }
    else { const int ntb = (int)gridDim.x / p.ncob; cob = b / ntb; b -= cob * ntb; }
    const int bw = b % p.blocksW; b /= p.blocksW;
    const int bh = b % p.blocksH; b /= p.blocksH;
    const int n0 = b * p.TN;
    const int ty0 = bh << p.lgTH, tx0 = bw << p.lgTW;        // first tile of the block
    const int co0 = cob * 16;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;
    const int npix = p.TN * HT * WT;

    // this lane's tile inside the block
    // ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32): with 8 tiles per row the map
    // li -> (li&3) + 4*(li>>3) + 8*(bit2 ^ bit3) gives every group one full row of 8 tiles per kk; tiles are 2 pixels =
    // KXP/2 = 10 sixteen-byte slots apart, so the 8 tiles cover the even (kk 0) / odd (kk 1) slots of the 256-byte bank row
    // exactly once (stride 24 puts tiles t and t+4 on the same banks: 2-way conflicts on all 16 patch reads).
    const int lt = p.lgTW == 3 ? (li & 3) + 4 * (li >> 3) + 8 * (((li >> 2) ^ (li >> 3)) & 1) : li;
    const int t = wave * 16 + lt;
    const int ttx = t & (TTW - 1), tty = (t >> p.lgTW) & (TTH - 1), ttn = t >> (p.lgTW + p.lgTH);
    const int pbase = ((ttn * HT + 2 * tty) * WT + 2 * ttx) * KXP + VEC * kk;

    // buffer resources: U whole, x from the first image of this workgroup (the host checks both spans stay below 2 GiB)
    const size_t img = (size_t)xH * xW * p.Cin;
    const int nimg = min(p.TN, p.N - n0);
    const __amdgpu_buffer_rsrc_t rx = pg_make_rsrc(p.x + (size_t)n0 * img, (unsigned)(nimg * img * 4));
    const __amdgpu_buffer_rsrc_t ru = pg_make_rsrc(p.u, (unsigned)((size_t)16 * p.Cout * p.Cin * 4));
    unsigned xsrc[XPT], usrc[UPT];
    int xdst[XPT], udst[UPT];
#pragma unroll
    for (int i = 0; i < XPT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / VEC, v = idx - q * VEC;
        const int r2 = (int)__umulhi((unsigned)q, p.mWT), tw = q - r2 * WT;
        const int tn = (int)__umulhi((unsigned)r2, p.mHT), th = r2 - tn * HT;
        const int n = n0 + tn;
        int ih = 2 * ty0 + th - 1, iw = 2 * tx0 + tw - 1;
        const bool in_tile = q < npix;
        const bool ok = in_tile && n < p.N && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        if (p.ups) { ih >>= 1; iw >>= 1; }
        xdst[i] = in_tile ? q * KXP + 4 * v : -1;
        xsrc[i] = ok ? 4u * (unsigned)(((tn * xH + ih) * xW + iw) * p.Cin + 4 * v) : PG_OOB;
    }
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
        const int idx = tid + 256 * i;
        const int r = idx / VEC, v = idx - r * VEC;          // r = xi*16 + co
        const int xi = r >> 4, co = co0 + (r & 15);
        udst[i] = r * KCP + 4 * v;
        usrc[i] = co < p.Cout ? 4u * (unsigned)((((v >> 1) * 16 + xi) * p.Cout + co) * 8 + 4 * (v & 1)) : PG_OOB;   // packs k0/8 (+1): wino_u_index
    }
    const unsigned upack = 4u * 16u * 8u * (unsigned)p.Cout;                                      // bytes of one 8-channel pack

    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 xreg[XPT], ureg[UPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < UPT; ++i) ureg[i] = pg_buf_load4(ru, usrc[i], (unsigned)(k0 >> 3) * upack);
#pragma unroll
        for (int i = 0; i < XPT; ++i) xreg[i] = pg_buf_load4(rx, xsrc[i], 4u * (unsigned)k0);
    };
    fetch(0);
    for (int k0 = 0; k0 < p.Cin; k0 += KC) {
        PG_STAMP(0);
#pragma unroll
        for (int i = 0; i < UPT; ++i) *reinterpret_cast<float4*>(ut + udst[i]) = ureg[i];
#pragma unroll
        for (int i = 0; i < XPT; ++i) if (xdst[i] >= 0) *reinterpret_cast<float4*>(xt + xdst[i]) = xreg[i];
        PG_STAMP(1);
        __syncthreads();
        PG_STAMP(2);
        if (k0 + KC < p.Cin) fetch(k0 + KC);
        PG_STAMP(3);

        // the patch as pairs of floats (the compiler still emits scalar v_add_f32: forcing v_pk_add_f32 through inline asm
        // needs an s_nop per instruction for the VALU->MFMA hazard the assembler cannot see, and measured slower)
        v2f d[4][4][VEC / 2];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const fragv t = *reinterpret_cast<const fragv*>(xt + pbase + (a * WT + c) * KXP);
#pragma unroll
                for (int h = 0; h < VEC / 2; ++h) d[a][c][h] = v2f{t[2 * h], t[2 * h + 1]};
            }
        // V = B^T d B, in place
#pragma unroll
        for (int h = 0; h < VEC / 2; ++h) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const v2f t0 = d[0][c][h] - d[2][c][h], t1 = d[1][c][h] + d[2][c][h];
                const v2f t2 = d[2][c][h] - d[1][c][h], t3 = d[1][c][h] - d[3][c][h];
                d[0][c][h] = t0; d[1][c][h] = t1; d[2][c][h] = t2; d[3][c][h] = t3;
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const v2f t0 = d[a][0][h] - d[a][2][h], t1 = d[a][1][h] + d[a][2][h];
                const v2f t2 = d[a][2][h] - d[a][1][h], t3 = d[a][1][h] - d[a][3][h];
                d[a][0][h] = t0; d[a][1][h] = t1; d[a][2][h] = t2; d[a][3][h] = t3;
            }
        }
        PG_STAMP(4);
        // MFMA order: the four Winograd positions of a row are interleaved, so that two MFMAs on the SAME accumulator are
        // never adjacent (v_mfma_f32_16x16x4_f32 issues every 32 cycles but a dependent accumulate needs 40, and any other
        // instruction between two dependent MFMAs costs ~43 more: MI355X_MICROARCH.md "Per-instruction cycle constants").
        fragv af[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[0][j] = *reinterpret_cast<const fragv*>(ut + (j * 16 + li) * KCP + VEC * kk);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < 4) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    af[nxt][j] = *reinterpret_cast<const fragv*>(ut + (((g + 1) * 4 + j) * 16 + li) * KCP + VEC * kk);
            }
#pragma unroll
            for (int s4 = 0; s4 < VEC; ++s4)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 * g + j] = MFMA16(af[cur][j][s4], d[g][j][s4 >> 1][s4 & 1], acc[4 * g + j]);
        }
        PG_STAMP(5);
        __syncthreads();
        PG_STAMP(6);
    }

    wino_epilogue(p, acc, co0 + 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
#ifdef PG_WINO_TRACE
    __builtin_amdgcn_sched_barrier(0);
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + 1) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
}


// ------------------------------------------------------------------------------------------------------------------
// Second-generation Winograd conv: input-transform reuse + LDS-DMA staging (round-2 rebuild).
//
// What the phase trace of conv_wino_kernel<4> showed (tools/exp/wino_trace.py, n9 @64 128->256, cycles per 16-channel chunk
// and wave, two workgroups per CU): LDS store of the staged registers 1200 (ds_write_b128 moves 79 B/clk/CU), barrier 800,
// fetch issue 520, patch reads + input transform 1000, 64 MFMAs 3100 (2048 if the pipe were free), barrier 300: the matrix
// pipe is busy 58 % even on the best layers because 3900 of 7000 cycles per chunk are staging / transform work that every
// 16-cout workgroup repeats for the same input region.
//
// This kernel:
//   * one workgroup owns 64 tiles x 16*NCB couts (NCB = 2: the transformed input V is computed ONCE per 32 couts, half the
//     patch reads / transforms / input staging per MFMA; 32-channel layers have all their couts in one workgroup, so their
//     input region is fetched once instead of once per 16 couts);
//   * K chunks of 8 channels, staged global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs (that is
//     what makes 128 accumulator registers + two waves per SIMD fit), no ds_write pass, out-of-image pixels zero-filled by
//     the buffer bounds check; double-buffered, ONE barrier per chunk;
//   * the LDS image is "quad-planar": plane q holds channels 4q..4q+3 of every pixel (resp. every (xi, cout) row of U)
//     as consecutive 16-byte slots — the layout LDS-DMA can write (wave-uniform base + lane * 16) — and region rows are
//     shifted by one slot on every second tile row, so that the 16 tiles a wave reads with one ds_read_b64 (2 tile rows x 8
//     tiles, 2 pixels apart) cover the 256-byte bank row exactly once: conflict-free patch and fragment reads.
//   * XK = 2: the input region is staged 16 channels at a time (every second chunk), i.e. 64 contiguous bytes per pixel instead
//     of 32: every 128-byte line of the activations then comes from L2 twice instead of four times.
//   * XS: DMA instructions per thread for one input region = 256-slot (4 KB) units of an X buffer.  The usual region (one image,
//     8 x 8 tiles: 18 x 19 slots x 2 planes = 684) fits XS = 3: 2 x (12 + 8) KB = 40 KB of LDS per workgroup, FOUR workgroups per
//     CU (124 VGPRs allow four waves per SIMD) instead of three with XS = 4.
template <int NCB, int XK, int XS, bool KSP = false, int EPI = EPI_GENERIC>
__global__ __launch_bounds__(256, 2) void conv_wino2_kernel(WinoP p)      // 2 waves per SIMD: <= 256 VGPRs + AGPRs (NCB = 4 at one wave per SIMD: 1.2-1.5x slower, tools/exp/rejected/wino2_64_couts_per_workgroup.txt)
{
    static_assert(!KSP || (NCB == 1 && XK == 1), "K split: 16 couts per workgroup, 8-channel input staging");
    static_assert(EPI == EPI_GENERIC || (NCB == 1 && XK == 1), "specialised epilogues: 16 couts per workgroup");
    constexpr int KC = 8;
    constexpr int XPL = 2 * XK;                              // channel-quad planes of the staged input region
    static_assert(XK == 1 ? (XS == 3 || XS == 4) : XS == 7, "input region: <= 768 / 1024 slots (XK = 1), 1792 (XK = 2)");
    constexpr int US = 2 * NCB;                              // ... for a U chunk: 2 planes x 16 xi x 16*NCB rows / 256
    constexpr int XSLOTS = XS * 256, USLOTS = US * 256;
    constexpr int XBYTES = XSLOTS * 16, UBYTES = USLOTS * 16; // LDS: [X 0][X 1][U 0][U 1]
    typedef float v2 __attribute__((ext_vector_type(2)));
    extern __shared__ __align__(16) float lds[];
    const int TTW = 1 << p.lgTW, TTH = 1 << p.lgTH;
    const int WT = 2 * TTW + 2, HT = 2 * TTH + 2, WTP = WT + 1;
    const int npixp = p.TN * HT * WTP;                       // slots of one input plane
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, kk = lane >> 4;
#ifdef PG_WINO_TRACE
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + 0) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
    int b = (int)pg_xcd_remap(blockIdx.x, gridDim.x);
    int cob, ks = 0;
    if constexpr (KSP) { const int q = (int)__umulhi((unsigned)b, p.mKs); ks = b - q * p.ksplit; b = q; }   // the splits of a block are adjacent
    const int blk = b;
    // no runtime divisions (five of them were ~100 of the ~450 fixed VALU instructions of a wave): blocksW / blocksH are powers of
    // two, the cout-block split is a magic multiplication (exact for b * divisor < 2^32, checked by the host)
    if (p.ncob == 1) cob = 0;
    else if (p.cout_minor) { const int q = (int)__umulhi((unsigned)b, p.mDiv); cob = b - q * p.ncob; b = q; }
    else { cob = (int)__umulhi((unsigned)b, p.mDiv); b -= cob * p.ntb; }
    const int bw = b & (p.blocksW - 1); b >>= p.lgBW;
    const int bh = b & (p.blocksH - 1); b >>= p.lgBH;
    const int n0 = b * p.TN;
    const int ty0 = bh << p.lgTH, tx0 = bw << p.lgTW;
    const int co0 = cob * 16 * NCB;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;

    const int t = wave * 16 + li;
    const int ttx = t & (TTW - 1), tty = (t >> p.lgTW) & (TTH - 1), ttn = t >> (p.lgTW + p.lgTH);
    // byte offset of patch element (a, c) inside a buffer: plane (kk >> 1), slot of pixel (2 tty + a, 2 ttx + c), half (kk & 1)
    int prow[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int y = 2 * tty + a;
        prow[a] = ((kk >> 1) * npixp + (ttn * HT + y) * WTP + 2 * ttx + ((y >> 1) & 1)) * 16 + (kk & 1) * 8;    // + plane pair of the chunk
    }
    const int ubyte = 2 * XBYTES + ((kk >> 1) * 256 * NCB + li) * 16 + (kk & 1) * 8;      // + buffer, + (xi * 16 * NCB + cb * 16) * 16

    const size_t img = (size_t)xH * xW * p.Cin;
    const int nimg = min(p.TN, p.N - n0);
    // DMA descriptors: instruction i of wave w fills slots [(i*4 + w)*64, +64); lane l supplies slot (i*4 + w)*64 + l
    unsigned xsrc[XS], usrc[US];
#pragma unroll
    for (int i = 0; i < XS; ++i) {
        const int sl = (i * 4 + wave) * 64 + lane;
        int q = 0;
#pragma unroll
        for (int j = 1; j < XPL; ++j) q += sl >= j * npixp ? 1 : 0;
        const int pi = sl - q * npixp;
        const int row = (int)__umulhi((unsigned)pi, p.mWT), xs = pi - row * WTP;        // mWT: magic reciprocal of WTP here
        const int tn = (int)__umulhi((unsigned)row, p.mHT), y = row - tn * HT;
        const int x = xs - ((y >> 1) & 1);
        int ih = 2 * ty0 + y - 1, iw = 2 * tx0 + x - 1;
        const bool ok = pi < npixp && (unsigned)x < (unsigned)WT && n0 + tn < p.N &&
                        (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        if (p.ups) { ih >>= 1; iw >>= 1; }
        xsrc[i] = ok ? 4u * (unsigned)(((tn * xH + ih) * xW + iw) * p.Cin + 4 * q) : PG_OOB;
    }
#pragma unroll
    for (int i = 0; i < US; ++i) {
        const int sl = (i * 4 + wave) * 64 + lane;
        const int q = sl / (256 * NCB), r = sl - q * 256 * NCB;                          // r = xi * 16*NCB + co
        const int xi = r / (16 * NCB), co = co0 + r - xi * 16 * NCB;
        usrc[i] = co < p.Cout ? 4u * (unsigned)((xi * p.Cout + co) * 8 + 4 * q) : PG_OOB;        // inside the 8-channel pack k0/8 (wino_u_index)
    }
    const unsigned upack = 4u * 16u * 8u * (unsigned)p.Cout;
    // LDS-DMA through inline asm: hipcc orders every later LDS read behind an LDS-DMA it can see (s_waitcnt vmcnt(0) before the
    // next ds_read), which would drain the prefetch of the NEXT chunk before the MFMAs of this one; the copies are ordered by hand
    // instead (the vmcnt(0) + barrier at the top of every chunk).  M0 = LDS byte address of the wave's 1 KiB destination.
    auto rsrc_words = [](const void* base, unsigned bytes) {        // the raw buffer descriptor pg_make_rsrc builds, as four SGPR words
        const unsigned long long a = (unsigned long long)base;
        return pg_u32x4{(unsigned)a, (unsigned)(a >> 32) & 0xffffu, bytes, 0x00020000u};
    };
    const pg_u32x4 rxs = rsrc_words(p.x + (size_t)n0 * img, (unsigned)(nimg * img * 4));
    const pg_u32x4 rus = rsrc_words(p.u, (unsigned)((size_t)16 * p.Cout * p.Cin * 4));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto dma16 = [&](const pg_u32x4& rs, unsigned voff, unsigned soff, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(rs), "s"(soff), "s"(dst) : "memory");
    };
    const unsigned wbase = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    auto dma_u = [&](int k0) {                                // U chunk k0 .. k0+7 -> U buffer (k0 / 8) & 1
        const unsigned dst = wbase + 2 * XBYTES + (unsigned)((k0 >> 3) & 1) * UBYTES, usoff = (unsigned)(k0 >> 3) * upack;
#pragma unroll
        for (int i = 0; i < US; ++i) dma16(rus, usrc[i], usoff, dst + i * 4096);
    };
    auto dma_x = [&](int k0) {                                // input channels k0 .. k0 + 8 XK - 1 -> X buffer (k0 / (8 XK)) & 1
        const unsigned dst = wbase + (unsigned)((k0 / (KC * XK)) & 1) * XBYTES, soff = 4u * (unsigned)k0;
#pragma unroll
        for (int i = 0; i < XS; ++i)
            if ((i * 4) * 64 < XPL * npixp) dma16(rxs, xsrc[i], soff, dst + i * 4096);     // (workgroup-uniform: skips unused instructions)
    };

    f32x4 acc[NCB][16];                                      // first written by the first chunk
    const int kbeg = KSP ? ks * p.kcper * KC : 0;
    const int kend = KSP ? min(p.Cin, kbeg + p.kcper * KC) : p.Cin;

    dma_u(kbeg);
    dma_x(kbeg);
    // One K chunk.  The first one starts its accumulators from the MFMA's constant-zero C operand instead of 64 zeroed registers
    // (the fixed per-workgroup instruction count is what bounds the 16/32-channel layers: rocprofv3 SQ_INSTS_VALU per wave).
    auto chunk = [&](const int k0, auto first) {
        PG_STAMP(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA of the chunk has landed ...
        PG_STAMP(1);
        __syncthreads();
        PG_STAMP(2);                                      // ... everyone's has, and nobody still reads the other buffer
        // explicit LDS address space: a volatile access through a generic pointer would become a flat load (vmcnt + lgkmcnt)
        typedef __attribute__((address_space(3))) const char* lds_cptr;
        typedef __attribute__((address_space(3))) const volatile v2* lds_v2ptr;
        const int sub = (k0 >> 3) % XK;                           // which 8-channel half of the staged region this chunk uses
        const lds_cptr xb = (lds_cptr)lds + ((k0 / (KC * XK)) & 1) * XBYTES + sub * 2 * npixp * 16;
        v2 d[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) d[a][c] = *(lds_v2ptr)(xb + prow[a] + c * 16);   // volatile: keep ds_read_b64 (a merged ds_read2_b64 is half rate)
        __builtin_amdgcn_sched_barrier(0);
        if (k0 + KC < kend) dma_u(k0 + KC);                   // in flight under the transform and the MFMAs below (issuing them
        if (sub == 0 && k0 + KC * XK < kend) dma_x(k0 + KC * XK);   // before the patch reads instead: step 11.39 / 11.50 vs 11.32 / 11.43 ms)
        __builtin_amdgcn_sched_barrier(0);
        PG_STAMP(3);
        // V = B^T d B, in place
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v2 t0 = d[0][c] - d[2][c], t1 = d[1][c] + d[2][c], t2 = d[2][c] - d[1][c], t3 = d[1][c] - d[3][c];
            d[0][c] = t0; d[1][c] = t1; d[2][c] = t2; d[3][c] = t3;
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const v2 t0 = d[a][0] - d[a][2], t1 = d[a][1] + d[a][2], t2 = d[a][2] - d[a][1], t3 = d[a][1] - d[a][3];
            d[a][0] = t0; d[a][1] = t1; d[a][2] = t2; d[a][3] = t3;
        }
        PG_STAMP(4);
        // Scheduling barrier between the input transform and the MFMAs: without it hipcc threads the transform's additions through the
        // MFMA stream (46 VALU, a mix of v_pk_add_f32 and scalar adds, one or two per MFMA gap); with it the transform is 32 v_pk_add_f32
        // up front and the 32 MFMAs follow back to back with only the U fragment reads between them: -1.1 ... -1.7 % over the layer set
        // alone, -0.045 ms per 1024^2 step (round 5, three same-box pairs).  Found through s_setprio, which acts as such a barrier; the
        // priority itself changes nothing (s_setprio 0 in its place measures the same), nor does loading the first row's U fragments
        // before the transform (8 more VGPRs: past the 128 of four waves per SIMD, +3 % with the spills).
        __builtin_amdgcn_sched_barrier(0);
        const lds_cptr ub = (lds_cptr)lds + ubyte + ((k0 >> 3) & 1) * UBYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                         // one row of Winograd positions at a time: 4 x NCB accumulators interleaved
            v2 af[NCB][4];
#pragma unroll
            for (int c = 0; c < NCB; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) af[c][j] = *(lds_v2ptr)(ub + (((g * 4 + j) * NCB + c) * 16) * 16);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
                        {
                            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                            acc[c][4 * g + j] = MFMA16(af[c][j][s2], d[g][j][s2], (decltype(first)::value && s2 == 0) ? zero4 : acc[c][4 * g + j]);
                        }
        }
        PG_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        PG_STAMP(6);
        };
    chunk(kbeg, std::true_type{});
    for (int k0 = kbeg + KC; k0 < kend; k0 += KC) chunk(k0, std::false_type{});
    __builtin_amdgcn_sched_barrier(0);                          // (the epilogue's address arithmetic stays out of the last chunk's MFMA stream)
    if constexpr (KSP) {
        // partial sums of this K slice: 2x2 outputs (the output transform is linear) -> slice (blk, ks) of the scratch, one float4
        // per lane and output pixel; the workgroup that arrives last adds the slices in split order (so the result does not depend
        // on who was last) and runs the fused epilogue
        f32x4 yq[4];
        wino_output_transform(acc[0], yq);
        // Slices and ticket are exchanged with agent-scope accesses (sc1: coherent across the XCDs' L2s by themselves) instead of
        // plain stores + __threadfence(): the release fence writes back the WHOLE L2 of the XCD (buffer_wbl2), ~1 us each and
        // serialised per XCD -- 768 workgroups spent 100+ us in fences (n3 @64 128->128: 141 us against 25 us unsplit).
        constexpr int SC1 = 16;                               // cache-policy bit 4 of the buffer builtins = sc1 on gfx94x / gfx950
        const __amdgpu_buffer_rsrc_t rp = pg_make_rsrc(p.ks_part + (size_t)blk * p.ksplit * 4096, (unsigned)p.ksplit * 16384u);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(pg_u32x4{__float_as_uint(yq[q][0]), __float_as_uint(yq[q][1]), __float_as_uint(yq[q][2]),
                                                            __float_as_uint(yq[q][3])}, rp, ((ks * 4 + q) * 256 + tid) * 16, 0, SC1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the slice has reached the coherence point ...
        __syncthreads();                                      // ... for every wave of the workgroup, before the ticket is taken
        __attribute__((address_space(3))) unsigned* const ticket = (__attribute__((address_space(3))) unsigned*)lds;
        if (tid == 0) *ticket = __hip_atomic_fetch_add(p.ks_count + blk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (*ticket != (unsigned)(p.ksplit - 1)) return;
        if (tid == 0) __hip_atomic_store(p.ks_count + blk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
#pragma unroll
        for (int q = 0; q < 4; ++q) yq[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int s2 = 0; s2 < p.ksplit; ++s2)                 // slice order, whoever arrived last: a deterministic sum
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const pg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, ((s2 * 4 + q) * 256 + tid) * 16, 0, SC1);
                yq[q] += f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
            }
        wino_epilogue_sel<EPI>(p, yq, co0 + 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
    } else if constexpr (EPI != EPI_GENERIC) {
        f32x4 yq[4];
        wino_output_transform(acc[0], yq);
        wino_epilogue_fast<EPI>(p, yq, co0 + 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
    } else if (p.pn_r) {                                             // (workgroup-uniform; the host launches ncob == 1 then)
        wino_epilogue_pixelnorm<NCB>(p, acc, 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
    } else if (p.pnb_y) {
        wino_epilogue_pnbwd<NCB>(p, acc, 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
    } else {
#pragma unroll
        for (int c = 0; c < NCB; ++c)
            wino_epilogue(p, acc[c], co0 + 16 * c + 4 * kk, n0 + ttn, 2 * (ty0 + tty), 2 * (tx0 + ttx));
    }
#ifdef PG_WINO_TRACE
    __builtin_amdgcn_sched_barrier(0);
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 8 + 1) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
}

// all layers of a network in one launch (layer l: w at wbase + woff[l], u at ubase + uoff[l])
constexpr int WB_MAX = 32;
struct WinoBatch {
    int n;
    int first_block[WB_MAX + 1];
    long long woff[WB_MAX], uoff[WB_MAX];
    int cout[WB_MAX], cin[WB_MAX];
    unsigned transposed;                        // bit l: layer l is a backward-data form (see pg_wino_transform_weights_batched)
};

__global__ __launch_bounds__(256) void wino_weights_batched_kernel(const float* __restrict__ wbase, float* __restrict__ ubase, WinoBatch d)
{
    int l = 0;
    while (l + 1 < d.n && (int)blockIdx.x >= d.first_block[l + 1]) ++l;
    const float* w = wbase + d.woff[l];
    float* u = ubase + d.uoff[l];
    const size_t n = (size_t)d.cout[l] * d.cin[l];
    const size_t i = (size_t)(blockIdx.x - d.first_block[l]) * 256 + threadIdx.x;
    const size_t xs = (size_t)d.cout[l] * 8;               // stride between Winograd positions inside a pack
    if ((d.transposed >> l) & 1u) {
        // Backward-data form straight from the forward parameter w[3][3][Cin'][Cout'] (flipped taps, channels swapped): the element
        // (co', ci') reads w[8 - tap][ci'][co'], i.e. the contiguous direction of the SOURCE is co' while the packs want ci'
        // contiguous.  A block covers 32 co' x 8 ci' (one pack): read with co' fastest (128-byte rows), transpose through LDS, write
        // with the pack order.  (Replaces pg_pack_dgrad_weights + a forward-form transform of its output: one read of w less and no
        // intermediate copy for the layers whose backward-data conv always runs Winograd.)
        __shared__ float tile[16][32 * 9];
        const int cout = d.cout[l];
        const size_t blk = blockIdx.x - d.first_block[l];
        const size_t ncob = (size_t)(cout + 31) / 32, cob = blk % ncob, pk = blk / ncob;
        const int rc = threadIdx.x & 31, rk = threadIdx.x >> 5;                 // read mapping: co' = 32 cob + rc, ci' = 8 pk + rk
        const size_t co = 32 * cob + rc, ci = 8 * pk + rk;
        float g[3][3], t[4][3];
        const bool live = co < (size_t)cout && ci < (size_t)d.cin[l];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = live ? w[(size_t)(8 - (a * 3 + b)) * n + ci * cout + co] : 0.f;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            t[0][b] = g[0][b];
            t[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
            t[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
            t[3][b] = g[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            tile[4 * a + 0][rc * 9 + rk] = t[a][0];
            tile[4 * a + 1][rc * 9 + rk] = 0.5f * ((t[a][0] + t[a][1]) + t[a][2]);
            tile[4 * a + 2][rc * 9 + rk] = 0.5f * ((t[a][0] - t[a][1]) + t[a][2]);
            tile[4 * a + 3][rc * 9 + rk] = t[a][2];
        }
        __syncthreads();
        const int wc = threadIdx.x >> 3, wk = threadIdx.x & 7;                  // write mapping: pack order
        const size_t wco = 32 * cob + wc, wci = 8 * pk + wk;
        if (wco < (size_t)cout && wci < (size_t)d.cin[l]) {
            float* ub = u + wino_u_index(0, wco, wci, cout);
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) ub[(size_t)xi * xs] = tile[xi][wc * 9 + wk];
        }
        return;
    }
    if (i >= n) return;
    const size_t pk = i / ((size_t)8 * d.cout[l]), rr = i - pk * 8 * d.cout[l];      // see wino_weights_kernel
    const size_t co = rr >> 3, ci = 8 * pk + (rr & 7), src = co * d.cin[l] + ci;
    float g[3][3], t[4][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) g[a][b] = w[(size_t)(a * 3 + b) * n + src];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        t[0][b] = g[0][b];
        t[1][b] = 0.5f * ((g[0][b] + g[1][b]) + g[2][b]);
        t[2][b] = 0.5f * ((g[0][b] - g[1][b]) + g[2][b]);
        t[3][b] = g[2][b];
    }
    float* ub = u + wino_u_index(0, co, ci, d.cout[l]);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        ub[(size_t)(4 * a + 0) * xs] = t[a][0];
        ub[(size_t)(4 * a + 1) * xs] = 0.5f * ((t[a][0] + t[a][1]) + t[a][2]);
        ub[(size_t)(4 * a + 2) * xs] = 0.5f * ((t[a][0] - t[a][1]) + t[a][2]);
        ub[(size_t)(4 * a + 3) * xs] = t[a][2];
    }
}

inline int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
inline bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

thread_local char g_wino_last[64] = "";
#ifdef PG_WINO_TRACE
thread_local unsigned long long* g_wino_trace = nullptr;
#endif
// Scratch for launches that split K across workgroups, registered per (device, stream) by the host layer (pg_set_workspace):
// layout in convp.h.  The library never allocates device memory itself.
using pgk::Workspace; using pgk::WS_TICKETS; using pgk::WS_HEAD; using pgk::find_workspace;
std::mutex g_ws_mutex;
std::vector<Workspace> g_ws;

thread_local int g_wino_epi = -1;              // 0: general epilogue everywhere (pg_debug_set_wino_epi)
thread_local int g_wino_ksplit = -1;           // -1: built-in choice; 0 / 1: never split K across workgroups; n: n slices where legal
thread_local int g_wino_vec = 0;               // 2 / 4: first-generation kernel with K chunks of 4*vec channels; 0: second-generation kernel,
                                               // built-in choice of couts per workgroup; 11 / 12: second generation, 16 / 32 couts (pg_debug_set_wino)

}  // namespace

extern "C" const char* pg_debug_last_wino_kernel(void) { return g_wino_last; }
#ifdef PG_WINO_TRACE
extern "C" int pg_debug_wino_trace(void* buf) { g_wino_trace = (unsigned long long*)buf; return 0; }
#endif
bool pgk::find_workspace(hipStream_t s, pgk::Workspace& out)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (const Workspace& w : g_ws)
        if (w.device == dev && w.stream == s) { out = w; return true; }
    return false;
}

extern "C" int pg_set_workspace(pg_stream_t stream, void* ptr, size_t bytes)
{
    if ((ptr == nullptr) != (bytes == 0)) return PG_E_ARG;
    if (ptr && (bytes <= WS_HEAD || ((size_t)ptr & 15))) return PG_E_ARG;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    for (size_t i = 0; i < g_ws.size(); ++i)
        if (g_ws[i].device == dev && g_ws[i].stream == (hipStream_t)stream) {
            if (ptr) { g_ws[i].ptr = (char*)ptr; g_ws[i].bytes = bytes; }
            else g_ws.erase(g_ws.begin() + i);
            return 0;
        }
    if (ptr) g_ws.push_back(Workspace{dev, (hipStream_t)stream, (char*)ptr, bytes});
    return 0;
}

extern "C" int pg_debug_set_wino_epi(int mode)
{
    if (mode != -1 && mode != 0) return PG_E_ARG;
    g_wino_epi = mode;
    return 0;
}

extern "C" int pg_debug_set_wino_ksplit(int n)
{
    if (n < -1 || n > 64) return PG_E_ARG;
    g_wino_ksplit = n;
    return 0;
}

extern "C" int pg_debug_set_wino(int vec)
{
    if (vec != 0 && vec != 2 && vec != 4 && vec != 11 && vec != 12 && vec != 20 && vec != 21) return PG_E_ARG;   // 20: second generation, tile kernels only; 21: the row-streaming form wherever it exists
    g_wino_vec = vec;
    return 0;
}

extern "C" int pg_wino_transform_weights(const float* w, float* u, int Cout, int Cin, pg_stream_t stream)
{
    if (!w || !u || Cout <= 0 || Cin <= 0) return PG_E_ARG;
    if (Cin & 7) return PG_E_ALIGN;                         // 8-channel packs
    size_t g = ((size_t)Cout * Cin + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wino_weights_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, u, Cout, Cin);
    return (int)hipGetLastError();
}

extern "C" int pg_wino_transform_weights_batched(const float* wbase, float* ubase, int nlayers, const int64_t* woff,
                                                 const int64_t* uoff, const int* cout, const int* cin, const int* transposed,
                                                 pg_stream_t stream)
{
    if (!wbase || !ubase || nlayers <= 0 || !woff || !uoff || !cout || !cin) return PG_E_ARG;
    for (int l0 = 0; l0 < nlayers; l0 += WB_MAX) {
        WinoBatch d;
        d.n = nlayers - l0 < WB_MAX ? nlayers - l0 : WB_MAX;
        d.transposed = 0;
        int total = 0;
        for (int l = 0; l < d.n; ++l) {
            const int i = l0 + l;
            if (cout[i] <= 0 || cin[i] <= 0 || woff[i] < 0 || uoff[i] < 0) return PG_E_ARG;
            if (cin[i] & 7) return PG_E_ALIGN;
            d.first_block[l] = total;
            d.woff[l] = woff[i]; d.uoff[l] = uoff[i]; d.cout[l] = cout[i]; d.cin[l] = cin[i];
            if (transposed && transposed[i]) {
                d.transposed |= 1u << l;
                total += ((cout[i] + 31) / 32) * (cin[i] / 8);               // one block per (32 couts, 8-channel pack)
            } else total += (int)(((size_t)cout[i] * cin[i] + 255) / 256);
        }
        d.first_block[d.n] = total;
        hipLaunchKernelGGL(wino_weights_batched_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, wbase, ubase, d);
    }
    return (int)hipGetLastError();
}

namespace {
// K slices of a launch with nblk (64-tile, 16-cout) blocks and nch 8-channel chunks when a scratch is registered (measured:
// tools/sweeps/bench_ksplit.py in isolation and in-step sweeps of the constants)
int wino_default_slices(int nblk, int nch)
{
    static const int ks_pairs = getenv("PG_WINO_KS_PAIRS") ? atoi(getenv("PG_WINO_KS_PAIRS")) : 432;
    static const int ks_target = getenv("PG_WINO_KS_TARGET") ? atoi(getenv("PG_WINO_KS_TARGET")) : 864;
    static const int ks_max = getenv("PG_WINO_KS_MAX") ? atoi(getenv("PG_WINO_KS_MAX")) : 8;
    static const int ks_minch = getenv("PG_WINO_KS_MINCH") ? atoi(getenv("PG_WINO_KS_MINCH")) : 4;
    int ks = 1;
    if (nblk <= ks_pairs) { ks = ks_target / nblk; if (ks > ks_max) ks = ks_max; if (ks > nch / ks_minch) ks = nch / ks_minch; }
    return ks < 1 ? 1 : ks;
}

// tile-block geometry of the tile kernels: 64 tiles = TN images x TTH x TTW tiles
void wino_block_geometry(int N, int H, int W, int& TTW, int& TTH, int& TN, int& ntb)
{
    static const int ttw_env = getenv("PG_WINO_TTW") ? atoi(getenv("PG_WINO_TTW")) : 8;          // tile-block width in tiles (experiment: 16 / 32 = flatter blocks)
    const int tilesW = W >> 1, tilesH = H >> 1;
    TTW = tilesW < ttw_env ? tilesW : ttw_env;
    TTH = 64 / TTW; if (TTH > tilesH) TTH = tilesH;
    TN = 64 / (TTW * TTH);
    ntb = ((N + TN - 1) / TN) * (tilesW / TTW) * (tilesH / TTH);
}
}  // namespace

extern "C" int pg_workspace_bytes(int kind, int N, int H, int W, int Cin, int Cout, size_t* bytes)
{
    if (!bytes || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    *bytes = 0;
    if (kind == 0) {                                         // pg_conv2d_wino_nhwc: K slices of the small maps
        if ((Cin & 7) || (Cout & 3) || !pow2(H) || !pow2(W) || H < 8 || W < 8) return PG_E_UNSUP;
        int TTW, TTH, TN, ntb;
        wino_block_geometry(N, H, W, TTW, TTH, TN, ntb);
        const int nblk = ntb * ((Cout + 15) / 16), nch = Cin >> 3;
        if (nblk > (int)WS_TICKETS) return 0;
        int ks = wino_default_slices(nblk, nch);
        if (ks > nch) ks = nch;
        if (ks > 1) { const int kcper = (nch + ks - 1) / ks; ks = (nch + kcper - 1) / kcper; }
        if (ks > 1) *bytes = WS_HEAD + (size_t)nblk * ks * 16384;
        return 0;
    }
    if (kind == 1) {                                         // pg_conv2d_nhwc, 4x4 valid conv on a 4x4 map (H = W = 4): one workgroup per (cout block, input pixel)
        if (H != 4 || W != 4 || (Cout & 15)) return PG_E_UNSUP;
        const int nt = N <= 16 ? 1 : 2;
        const size_t nblk = (size_t)(Cout >> 4) * ((N + 16 * nt - 1) / (16 * nt));
        if (nblk <= 256) *bytes = WS_HEAD + nblk * 16 * nt * 1024;
        return 0;
    }
    return PG_E_ARG;
}

namespace {
int wino_conv(const float* x, const float* u, const float* bias, const float* mask, float* y,
              float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
              float* yup, const float* upmask, float up_mul,
              int N, int H, int W, int Cin, int Cout, int ups,
              float scale, float slope, float mask_slope, float* pn_r, float pn_eps, pg_stream_t stream,
              const float* pnb_y = nullptr, const float* pnb_r = nullptr)
{
    if (!x || !u || !y || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return PG_E_ARG;
    if ((Cin & 7) || (Cout & 3)) return PG_E_ALIGN;         // 8-channel packs of U, four couts per lane
    if ((Cin & 15) && (g_wino_vec == 2 || g_wino_vec == 4)) return PG_E_UNSUP;   // first generation: 16-channel chunks
    if (!pow2(H) || !pow2(W) || H < 8 || W < 8) return PG_E_UNSUP;
    if ((long long)N * H * W * Cin >= (1ll << 31) || (long long)N * H * W * Cout >= (1ll << 29) || (long long)16 * Cout * Cin >= (1ll << 31))
        return PG_E_UNSUP;
    WinoP p;
    p.x = x; p.u = u; p.bias = bias; p.mask = mask; p.y = y;
#ifdef PG_WINO_TRACE
    p.trace = g_wino_trace;
#endif
    const int flags = ups;                                  // PG_FLAG_* (pggan_hip.h)
    ups = flags & PG_FLAG_UPSAMPLE;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.ups = ups;
    p.mask_bytes = (flags & PG_FLAG_MASK_BYTES) ? 1 : 0; p.y_bytes = (flags & PG_FLAG_Y_BYTES) ? 1 : 0;
    if (p.y_bytes && !ypool) return PG_E_UNSUP;
    p.ysigns = nullptr;
    if (flags & PG_FLAG_SIGNS_OUT) {
        if (!mask || p.mask_bytes || yup) return PG_E_ARG;
        p.ysigns = reinterpret_cast<unsigned char*>(const_cast<float*>(mask));
        p.mask = nullptr;
    }
    p.scale = scale; p.slope = slope; p.mask_slope = mask_slope;
    p.ypool = ypool; p.pool_other = pool_other; p.pool_a = pool_a; p.pool_b = pool_b; p.pool_only = pool_only;
    p.yup = yup; p.upmask = upmask; p.up_mul = up_mul;
    p.pn_r = pn_r; p.pn_eps = pn_eps;
    if (pn_r && (Cout > 32 || mask || ypool || yup || flags != ups || (g_wino_vec != 0 && g_wino_vec < 10))) return PG_E_UNSUP;
    p.pnb_y = pnb_y; p.pnb_r = pnb_r;
    if (pnb_y && (!pnb_r || pn_r || Cout > 32 || mask || yup || flags != 0 || (g_wino_vec != 0 && g_wino_vec < 10))) return PG_E_UNSUP;
    // Thin layers of the high-resolution stages: the row-streaming kernel (conv_wino_strip.hip) where it measured faster than the
    // tile kernel below (PG_WINO_STRIP=0 / pg_debug_set_wino(20): tile kernels only; a forced tile variant keeps the tile kernel)
    static const int strip_env = getenv("PG_WINO_STRIP") ? atoi(getenv("PG_WINO_STRIP")) : 1;
    static const int strip_minw = getenv("PG_WINO_STRIP_MINW") ? atoi(getenv("PG_WINO_STRIP_MINW")) : 128;
    static const int strip_maxcin = getenv("PG_WINO_STRIP_MAXCIN") ? atoi(getenv("PG_WINO_STRIP_MAXCIN")) : 16;   // (32-channel inputs: the 87 KB ring leaves one workgroup per CU, 0.6-1.0x the tile kernel)
    // (exception: the pool-adjoint epilogue of a 32-channel layer writes 4x its pixels -- the tile kernel's 64-byte segments reach 2 TB/s,
    //  the strip kernel's full rows 3.1: 127.6 vs 160.8 us at n9 256^2 32->32, but 42.8 vs 39.2 us at n3)
    const bool strip_unpool32 = Cin == 32 && yup && (long long)N * H * W >= 400000;
    if ((strip_env && g_wino_vec == 0 && W >= strip_minw && (Cin <= strip_maxcin || strip_unpool32)) || g_wino_vec == 21) {
        const int sepi = (g_wino_epi != 0 ? 1 : 0) | (g_wino_vec == 21 ? 2 : 0);
        p.ksplit = 1; p.kcper = Cin >> 3; p.mKs = 0; p.ks_count = nullptr; p.ks_part = nullptr;
        const int rc = launch_wino_strip(p, sepi, (hipStream_t)stream, g_wino_last, sizeof(g_wino_last));
        if (rc != PG_E_UNSUP) return rc;
    }
    const int tilesW = W >> 1, tilesH = H >> 1;
    int TTW, TTH, TN, ntb_geo;
    wino_block_geometry(N, H, W, TTW, TTH, TN, ntb_geo);
    p.lgTW = ilog2i(TTW); p.lgTH = ilog2i(TTH); p.TN = TN;
    p.blocksW = tilesW / TTW; p.blocksH = tilesH / TTH;
    const int WT = 2 * TTW + 2, HT = 2 * TTH + 2;
    if (TN * HT * WT > XMAX) return PG_E_UNSUP;
    {   // spans addressed through 32-bit buffer offsets (offset 0x80000000 is the out-of-range marker)
        const long long xspan = (long long)TN * (ups ? (H >> 1) : H) * (ups ? (W >> 1) : W) * Cin * 4, uspan = (long long)16 * Cout * Cin * 4;
        if (xspan >= (1ll << 31) || uspan >= (1ll << 31)) return PG_E_UNSUP;
    }
    p.mWT = (unsigned)((1ull << 32) / (unsigned)WT) + 1u; p.mHT = (unsigned)((1ull << 32) / (unsigned)HT) + 1u;
    const int vec = g_wino_vec >= 20 ? 0 : g_wino_vec;
    const int ntb = ntb_geo;                                                      // tile blocks of 64 tiles
    if (vec >= 10 || vec == 0) {
        // second-generation kernel (LDS-DMA, 8-channel chunks, 16*NCB couts per workgroup); two cout blocks per workgroup
        // when that still leaves at least two workgroups per CU
        int ncb = (vec >= 10 && vec < 20) ? vec - 10 : ((Cin >= 512 && (long long)ntb * ((Cout + 31) / 32) >= 768) ? 2 : 1);   // measured: tools/sweeps/sweep_wino.py
        if (ncb != 1 && ncb != 2) return PG_E_ARG;
        if (pn_r || pnb_y) ncb = Cout > 16 ? 2 : 1;           // PixelNorm epilogue / adjoint: all couts of a pixel in one workgroup
        // Staging the input region 16 channels at a time (XK = 2: every activation line comes from L2 twice instead of four times)
        // costs a third workgroup per CU (74 KB of LDS) and measured SLOWER on every layer of the 1024^2 step but 512->512 @16
        // (n9 @256 32->64: 137 -> 170 us, step 13.8 -> 14.6 ms): PG_WINO_XK=2 keeps it reachable for sweeps.  Issuing the copies
        // after the transform or spread over the MFMA groups instead of right after the patch reads: equal / 1 % slower.
        static const int xk_env = getenv("PG_WINO_XK") ? atoi(getenv("PG_WINO_XK")) : 1;
        static const int xs_env = getenv("PG_WINO_XS") ? atoi(getenv("PG_WINO_XS")) : 3;     // 4: always 16 KB X buffers (sweeps)
        const int WTP = WT + 1;
        const int xk = (xk_env == 2 && ncb == 1 && 4 * TN * HT * WTP <= 1792) ? 2 : 1;
        const int xslots = 2 * xk * TN * HT * WTP;
        if (xslots > (xk == 1 ? 1024 : 1792)) return PG_E_UNSUP;
        const int xs = xk == 2 ? 7 : (xslots <= 768 && xs_env == 3) ? 3 : 4;
        p.mWT = (unsigned)((1ull << 32) / (unsigned)WTP) + 1u;
        p.ncob = (Cout + 16 * ncb - 1) / (16 * ncb);
        p.ntb = ntb; p.lgBW = ilog2i(p.blocksW); p.lgBH = ilog2i(p.blocksH);
        // cout blocks of one tile block adjacent (small weights) or all tile blocks of one cout block adjacent; the magic
        // division by ntb is exact only while ncob * ntb^2 < 2^32
        p.cout_minor = (long long)16 * Cout * Cin * 4 <= (2ll << 20) || (long long)p.ncob * ntb * ntb >= (1ll << 32) || ntb == 1;
        if ((long long)p.ncob * p.ncob * ntb >= (1ll << 32)) return PG_E_UNSUP;
        p.mDiv = (unsigned)((1ull << 32) / (unsigned)(p.cout_minor ? p.ncob : ntb)) + 1u;
        const size_t smem2 = (size_t)2 * (xs * 256 + 512 * ncb) * 16;
        // Fewer workgroups than the chip holds (four per CU) on a deep K loop -- 16x16 / 32x32 maps at minibatch 3: slice K over
        // up to 8 workgroups per (tile block, cout block), at least 4 chunks each, through the stream's registered scratch
        const int nblk = ntb * p.ncob, nch = Cin >> 3;
        int ks = 1;
        Workspace ws{};
        if (ncb == 1 && xk == 1 && !pn_r && !pnb_y && g_wino_ksplit != 0 && g_wino_ksplit != 1 && nblk <= (int)WS_TICKETS &&
            find_workspace((hipStream_t)stream, ws)) {
            ks = g_wino_ksplit > 1 ? g_wino_ksplit : wino_default_slices(nblk, nch);
            if (ks > nch) ks = nch;
            if (ks > 1) {
                p.kcper = (nch + ks - 1) / ks;
                ks = (nch + p.kcper - 1) / p.kcper;
                if (WS_HEAD + (size_t)nblk * ks * 16384 > ws.bytes || (long long)nblk * ks * ks >= (1ll << 32)) ks = 1;
            }
            if (ks < 1) ks = 1;
        }
        p.ksplit = ks;
        if (ks > 1) {
            p.mKs = (unsigned)((1ull << 32) / (unsigned)ks) + 1u;
            p.ks_count = reinterpret_cast<unsigned*>(ws.ptr); p.ks_part = reinterpret_cast<float*>(ws.ptr + WS_HEAD);
        } else { p.kcper = nch; p.mKs = 0; p.ks_count = nullptr; p.ks_part = nullptr; }
        dim3 grid2((unsigned)(nblk * ks));

        // the two specialised epilogues (see wino_epilogue_fast) when the launch asks for nothing else
        static const int epi_env = getenv("PG_WINO_EPI") ? atoi(getenv("PG_WINO_EPI")) : 1;              // 0: general epilogue everywhere (A/B)
        int epi = EPI_GENERIC;
        if (ncb == 1 && xk == 1 && g_wino_epi != 0 && epi_env != 0 && !ypool && !yup && !p.y_bytes && !p.ysigns && !pn_r && !pnb_y &&
            (long long)H * W * Cout * 4 < (1ll << 31)) {
            if (!p.mask) epi = EPI_PLAIN;
            else if (p.mask_bytes) epi = EPI_MASKB;
        }
        void (*fn)(WinoP);
#define PG_W2(XS_, KSP_) (epi == EPI_PLAIN ? conv_wino2_kernel<1, 1, XS_, KSP_, EPI_PLAIN> : epi == EPI_MASKB ? conv_wino2_kernel<1, 1, XS_, KSP_, EPI_MASKB> \
                                                                                                             : conv_wino2_kernel<1, 1, XS_, KSP_, EPI_GENERIC>)
        if (ncb == 2) fn = xs == 3 ? conv_wino2_kernel<2, 1, 3> : conv_wino2_kernel<2, 1, 4>;
        else if (xk == 2) fn = conv_wino2_kernel<1, 2, 7>;
        else if (ks > 1) fn = xs == 3 ? PG_W2(3, true) : PG_W2(4, true);
        else fn = xs == 3 ? PG_W2(3, false) : PG_W2(4, false);
#undef PG_W2
        snprintf(g_wino_last, sizeof(g_wino_last), "conv_wino2_kernel<%d, %d, %d, %s, %d>", ncb, xk, xs, ks > 1 ? "true" : "false", epi);   // (as the profiler demangles it)
        if (smem2 > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(fn, grid2, dim3(256), smem2, (hipStream_t)stream, p);
        return (int)hipGetLastError();
    }
    const size_t smem = ((size_t)16 * 16 * (vec == 4 ? 24 : 12) + (size_t)TN * HT * WT * (vec == 4 ? 20 : 12)) * sizeof(float);
    p.ncob = (Cout + 15) / 16;
    p.cout_minor = (long long)16 * Cout * Cin * 4 <= (2ll << 20);      // measured: tools/sweeps/sweep_wino.py
    dim3 grid((unsigned)(ntb * p.ncob));
    snprintf(g_wino_last, sizeof(g_wino_last), "conv_wino_kernel<%d>", vec);
    if (vec == 4) {
        if (smem > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wino_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(conv_wino_kernel<4>, grid, dim3(256), smem, (hipStream_t)stream, p);
    } else {
        hipLaunchKernelGGL(conv_wino_kernel<2>, grid, dim3(256), smem, (hipStream_t)stream, p);
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" int pg_conv2d_wino_nhwc(const float* x, const float* u, const float* bias, const float* mask, float* y,
                                   float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
                                   float* yup, const float* upmask, float up_mul,
                                   int N, int H, int W, int Cin, int Cout, int ups,
                                   float scale, float slope, float mask_slope, pg_stream_t stream)
{
    return wino_conv(x, u, bias, mask, y, ypool, pool_other, pool_a, pool_b, pool_only, yup, upmask, up_mul,
                     N, H, W, Cin, Cout, ups, scale, slope, mask_slope, nullptr, 0.f, stream);
}

extern "C" int pg_conv2d_wino_pixelnorm_nhwc(const float* x, const float* u, const float* bias, float* y, float* r,
                                             int N, int H, int W, int Cin, int Cout, int ups,
                                             float scale, float slope, float eps, pg_stream_t stream)
{
    if (!r) return PG_E_ARG;
    return wino_conv(x, u, bias, nullptr, y, nullptr, nullptr, 1.f, 0.f, 0, nullptr, nullptr, 1.f,
                     N, H, W, Cin, Cout, ups ? PG_FLAG_UPSAMPLE : 0, scale, slope, 0.2f, r, eps, stream);
}

extern "C" int pg_conv2d_wino_pnbwd_nhwc(const float* x, const float* u, const float* ysaved, const float* r, float* y,
                                         int pool, const float* pool_other, float pool_a, float pool_b,
                                         int N, int H, int W, int Cin, int Cout, float scale, float slope, pg_stream_t stream)
{
    if (!ysaved || !r) return PG_E_ARG;
    if (pool && ((H | W) & 1)) return PG_E_ARG;
    return wino_conv(x, u, nullptr, nullptr, y, pool ? y : nullptr, pool ? pool_other : nullptr, pool_a, pool_b, pool ? 1 : 0, nullptr, nullptr, 1.f,
                     N, H, W, Cin, Cout, 0, scale, 1.f, slope, nullptr, 0.f, stream, ysaved, r);
}
