// Row-streaming 3x3 convolution for the 8-cout layers of the 1024^2 stage (gfx950 / CDNA4).
//
// Replaces F.conv2d + bias + LeakyReLU of reference network.py:33-36 (and its backward-data / gradient-penalty tangent forms)
// for the layers 8->8 and 16->8 — the shapes conv_thin_kernel (conv_igemm.hip) serves with one 8 x 32 pixel tile per workgroup.
// Those layers sit at the HBM ridge (18-24 FLOP/B); a tile kernel loses there to (a) the halo re-read of a small tile (1.33x on
// the input), (b) 1 KB row pieces scattered over ten DRAM pages per workgroup, (c) a load -> barrier -> compute -> store life
// cycle whose overlap depends on other workgroups being in a different phase, and (d) ~700 scalar / vector instructions of
// address arithmetic and run-time epilogue selection per 144 MFMAs (measured: the scalar unit, the VALU, the LDS and the matrix
// pipe all sit at 40-75 % — nothing saturates, everything waits).
//
// This kernel: a workgroup owns a STRIP of 64 output columns and walks DOWN `seg` rows of one image, four rows per step.
//   * input rows enter LDS exactly once per strip (66 of 64 columns: 3 % horizontal halo, no vertical re-read) through LDS-DMA
//     (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass; rows above / below the image fall outside the
//     records of the per-image buffer and are zero-filled by the hardware, the left / right border columns carry an
//     out-of-range offset from the start; a ring of three blocks of four rows, the block two steps ahead is in flight while the
//     current one is consumed, ONE barrier per step;
//   * LDS layout "quad-planar" (plane q = channels 4q..4q+3 of every pixel as consecutive 16-byte slots): the only layout LDS-DMA
//     can write, and conflict-free for the ds_read_b128 of v_mfma_f32_4x4x1_16B operands (16 consecutive pixels per lane group);
//     every LDS address is a per-lane base plus a compile-time constant (the ring phase is a template parameter);
//   * MFMA mapping as in conv_thin_kernel: v_mfma_f32_4x4x1_16B_f32, block = (cout quad, pixel quad), A = weights, B = pixels,
//     so a lane ends up with 4 consecutive couts of one pixel (16-byte NHWC stores); wave w owns rows {2(w>>1), 2(w>>1)+1} x
//     columns [32(w&1), +32) of the step, i.e. complete 2x2 pooling windows; same accumulation order as the tile kernel
//     (bit-identical results);
//   * the epilogue is a TEMPLATE parameter (forward / masked / PixelNorm / PixelNorm adjoint / generic), every global access goes
//     through a per-image raw buffer with a 32-bit offset that advances by a constant per step, the masks of the step are
//     fetched before its MFMAs, and the next step waits for its DMA only (counted vmcnt: the stores of a step drain under the
//     MFMAs of the next one).
// Fused epilogues (same set as the tile kernel): x scale, + bias, LeakyReLU | x LeakyReLU'(saved activation, fp32 or sign bytes),
// sign bytes out, 2x2 average pool + fade-in blend, PixelNorm, adjoint of the previous (LeakyReLU -> PixelNorm); nearest x2
// upsample fused into the row gather (the DMA source address).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <cstdio>
#include <type_traits>
#include "pggan_hip.h"
#include "bufload.h"
#include "convp.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef PG_STRIP_ABL            // ablation build of tools/sweeps/bench_strip.py: 1 no DMA after the prologue, 2 no stores, 4 one tap only
#define PG_STRIP_ABL 0
#endif

namespace {

using pgk::ConvP;
using pgk::pg_sign_byte;
using pgk::pg_sign_factors;

constexpr int SW = 64;            // output columns of a strip
constexpr int RP = SW + 2;        // slots of one plane row (left / right halo column included)
constexpr int RB = 4;             // rows of a block = output rows per step
constexpr int NBLK = 3;           // ring: blocks it, it + 1 are read while block it + 2 lands

enum { EPI_GENERIC = 0, EPI_FWD = 1, EPI_MASK = 2, EPI_PN = 3, EPI_PNB = 4 };

struct SArgs {
    const float* x; const float* w; const float* bias; const void* mask; float* y;
    unsigned char* ysigns; float* pn_r; const float* pnb_y; const float* pnb_r; float* ypool; const float* pool_other;
    float scale, slope, mask_slope, pn_eps, pool_a, pool_b;
    int H, W, ups, mask_bytes, y_bytes, pool_only;
    int strips, segs, seg_rows;
    // pool adjoint evaluated in the row gather (pg_conv2d_unpooled_nhwc): input[h][w][c] = gmul * x[h/2][w/2][c] * lrelu'(gbytes[h][w][c])
    const unsigned char* gbytes; float gmul, gslope;
    // fromRGB evaluated in the row gather (pg_conv2d_fromrgb_nhwc): x is the IMAGE [N][rgbC][H][W]; the conv's input
    //   x0[h][w][co] = lrelu(rgb_scale * sum_c rgb_w[co][c] * img[c][h][w] + rgb_b[co])   (never written), sign bytes of x0 -> xsigns
    const float* rgb_w; const float* rgb_b; float rgb_scale, rgb_slope; int rgbC; unsigned char* xsigns;
    // toRGB on top of the PixelNorm epilogue (pg_conv2d_pixelnorm_torgb_nhwc): t_out[n][c][h][w] = t_scale * sum_co t_w[c][co] * y[h][w][co] + t_b[c]
    // ... or, on top of the masked form (pg_conv2d_masked_fromrgb_bwd_nhwc: the entry block's backward-data conv), fromRGB's backward-data:
    //   t_out[n][c][h][w] = t_scale * sum_co t_w[co][c] * y[h][w][co];  element (c, co) of t_w at t_w[c * t_sc + co * t_sco]; t_only: y itself is not written
    float* t_out; const float* t_w; const float* t_b; float t_scale; int tC, t_sc, t_sco, t_only;
    // ... and fromRGB's WEIGHT gradient accumulated over the workgroup's pixels (one commit of 8 x (C + 1) atomics per workgroup):
    //   fw_dw[co][c] += fw_scale * sum_pixels y[co] * fw_img[c],  fw_db[co] += sum_pixels y[co]      (y = the masked result, as above)
    const float* fw_img; float* fw_dw; float* fw_db; float fw_scale;
};

template <int CIN> struct Blk {
    static constexpr int C4 = CIN / 4;
    static constexpr int USED = C4 * RB * RP;                 // slots carrying data: [plane q][row m][pixel]
    static constexpr int NWI = (USED + 63) / 64;              // DMA wave-instructions per block
    static constexpr int NI = (NWI + 3) / 4;                  // ... per thread (instruction ii = i * 4 + wave)
    static constexpr int SLOTS = NWI * 64;                    // block pitch (the tail of the last instruction is zero-filled padding)
};

typedef __attribute__((address_space(3))) const char* lds_cptr;
typedef __attribute__((address_space(3))) const f32x4* lds_v4ptr;

__device__ __forceinline__ pg_u32x4 rsrc_words(const void* base, unsigned bytes)
{
    const unsigned long long a = (unsigned long long)base;
    return pg_u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                    (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)), bytes, 0x00020000u};
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(pg_u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)},
                                           r, (int)voff, 0, 0);
}

// GM 1 (GATH): the input rows are not copied but COMPUTED (pool adjoint of a coarser gradient x sign bytes of the finer activation): they
// are fetched into registers one block ahead (under the MFMAs of the step), multiplied and written to the ring with ds_write_b128.
// GM 2 (round 6): the same ring filled with the output of the block's fromRGB layer, computed from the image (reference
// network.py:145 in front of :33-36): the 1x1 conv + LeakyReLU of a pixel's four channels per slot, in fromrgb_fwd_pix_kernel's
// order of operations (bit-identical x0); 12 B of image per pixel instead of a 32 B activation written by one launch and read by this one.
constexpr int RGB_MAXC = 3;
// XM: the RGB-side extras of the epilogue, compiled in only where asked for (their registers -- 12 weights, 16 + 6 for the weight gradient --
// made the plain masked kernel spill 62 VGPRs when they were run-time options of one instantiation): bit 0 the 1x1 RGB layer on the finished
// value (toRGB / fromRGB's backward-data), bit 1 fromRGB's weight gradient.
template <int COUT, int CIN, int EPI, bool WREG, int GM, int XM = 0>
__device__ __forceinline__ void conv_strip_body(const SArgs& p)
{
    constexpr bool GATH = GM != 0;
    using B = Blk<CIN>;
    constexpr int C4 = B::C4;
    constexpr int QO = COUT / 4, QP = 16 / QO, PXG = 4 * QP;             // pixels per MFMA group: 32 (8 couts)
    constexpr int GPR = 32 / PXG, G = 2 * GPR;                           // groups per row of the wave's 32 columns; groups per wave
    constexpr int BLKB = B::SLOTS * 16;                                   // bytes of a ring block
    constexpr bool GEN = EPI == EPI_GENERIC;
    extern __shared__ __align__(16) float lds[];
    // weights in LDS (when not in registers): [9][COUT][WS]; the lanes of a ds_read_b128 group read 8 different cout rows, which a
    // 64-byte row stride (Cin = 16) would put on the same banks two by two (SQ_LDS_BANK_CONFLICT 0.28-0.30 of the LDS cycles, round 3)
    constexpr int WS = CIN == 16 ? CIN + 4 : CIN;
    float* wl = lds + NBLK * B::SLOTS * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, j = lane & 3, qo = blk % QO, qp = blk / QO;

    int t = (int)pg_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = __builtin_amdgcn_readfirstlane(t % p.strips); t /= p.strips;      // (scalar: the buffer descriptors live in SGPRs)
    const int seg = __builtin_amdgcn_readfirstlane(t % p.segs), n = __builtin_amdgcn_readfirstlane(t / p.segs);
    const int r0 = seg * p.seg_rows, ow0 = strip * SW;
    const int niter = p.seg_rows / RB;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;
    const unsigned npix = (unsigned)(p.H * p.W);

    // ---- DMA offsets (per thread): instruction i of wave w fills slots [(4 i + w) 64, +64) of a block.  voff[i] = byte offset of
    // the lane's source inside the image for block 0 of the segment; it advances by a constant per block.  A row above the image
    // makes the offset negative (= huge), a row below it exceeds the records: the hardware returns zeros.  Border columns and the
    // padding slots carry PG_OOB and never advance.
    const unsigned rowbytes = 4u * (unsigned)(xW * CIN);
    unsigned voff[B::NI], vstep[B::NI];
    const unsigned blkstep = (p.ups ? 2u : 4u) * rowbytes;
#pragma unroll
    for (int i = 0; i < B::NI; ++i) {
        const int sl = (i * 4 + wave) * 64 + lane;
        const int q = sl / (RB * RP), rem = sl - q * (RB * RP);
        const int m = rem / RP, px = rem - m * RP;
        const int col = ow0 - 1 + px, row = r0 - 1 + m;
        const bool ok = sl < B::USED && (unsigned)col < (unsigned)p.W;
        // (mod 2^32: a row above the image is "negative" = beyond the records, and walks into the image as blocks are added)
        voff[i] = ok ? (unsigned)(p.ups ? (row >> 1) : row) * rowbytes + 4u * (unsigned)((p.ups ? (col >> 1) : col) * CIN + 4 * q) : PG_OOB;
        vstep[i] = ok ? blkstep : 0u;                               // (an out-of-range lane stays out of range)
    }
    const size_t ximg = (size_t)xH * xW * CIN;
    const pg_u32x4 rxs = rsrc_words(p.x + (size_t)n * ximg, (unsigned)(ximg * 4));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    // LDS-DMA through inline asm (hipcc would order every later ds_read behind a DMA it can see); M0 = the wave's 1 KiB destination
    auto dma16 = [&](unsigned vo, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(rxs), "s"(dst) : "memory");
    };
    auto issue_block = [&](int pos) {                         // the next block of four input rows -> ring position pos
#pragma unroll
        for (int i = 0; i < B::NI; ++i) {
            if (i * 4 + wave < B::NWI) dma16(voff[i], wdst + (unsigned)pos * BLKB + (unsigned)i * 4096u);
            voff[i] += vstep[i];
        }
    };

    // ---- gathered input (GATH): slot sl = i * 256 + tid of a block; running byte offsets into the coarse image and the sign bytes
    constexpr int NG = GATH ? (B::USED + 255) / 256 : 1;
    unsigned cvo[NG], cvs[NG], bvo[NG], bvs[NG];
    float4 gxr[NG];
    unsigned gbr[NG];
    __amdgpu_buffer_rsrc_t rgx = pg_make_rsrc(p.x + (size_t)n * ximg, (unsigned)(ximg * 4)), rgb = rgx;
    // GM 2: the lane's pixel of every slot (plane-0 byte offset inside the image, running row), the fromRGB weights of the slot's channel quad
    constexpr int NR = GM == 2 ? NG : 1;
    int rrow[NR];
    float rw[NR][4][RGB_MAXC], rb[NR][4], rpx[NR][RGB_MAXC];
    if constexpr (GM == 2) {
        rgx = pg_make_rsrc(p.x + (size_t)n * npix * p.rgbC, npix * (unsigned)p.rgbC * 4u);
        rgb = pg_make_rsrc(p.xsigns + (size_t)n * npix * C4, npix * C4);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int sl = i * 256 + tid;
            const int q = sl / (RB * RP), rem = sl - q * (RB * RP);
            const int m = rem / RP, px = rem - m * RP;
            const int col = ow0 - 1 + px;
            const bool ok = sl < B::USED && (unsigned)col < (unsigned)p.W;
            rrow[i] = ok ? r0 - 1 + m : -(1 << 28);                    // (a row that never enters the image)
            cvo[i] = ok ? 4u * (unsigned)((r0 - 1 + m) * p.W + col) : PG_OOB;       // (row -1: wraps, never fetched -- the row test below)
            cvs[i] = ok ? 4u * (unsigned)(RB * p.W) : 0u;
            // sign byte of the slot: interior columns of the strip only (the halo columns belong to the neighbours)
            bvo[i] = (ok && px >= 1 && px <= SW) ? (unsigned)((r0 - 1 + m) * p.W + col) * C4 + q : PG_OOB;
            bvs[i] = (ok && px >= 1 && px <= SW) ? (unsigned)(RB * p.W * C4) : 0u;
            const int qq = sl < B::USED ? q : 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c) rw[i][k][c] = c < p.rgbC ? p.rgb_w[(4 * qq + k) * p.rgbC + c] : 0.f;
                rb[i][k] = p.rgb_b ? p.rgb_b[4 * qq + k] : 0.f;
            }
        }
    }
    if constexpr (GM == 1) {
        rgb = pg_make_rsrc(p.gbytes + (size_t)n * npix * C4, npix * C4);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int sl = i * 256 + tid;
            const int q = sl / (RB * RP), rem = sl - q * (RB * RP);
            const int m = rem / RP, px = rem - m * RP;
            const int col = ow0 - 1 + px, row = r0 - 1 + m;
            const bool ok = sl < B::USED && (unsigned)col < (unsigned)p.W;
            cvo[i] = ok ? (unsigned)(row >> 1) * rowbytes + 4u * (unsigned)((col >> 1) * CIN + 4 * q) : PG_OOB;
            cvs[i] = ok ? 2u * rowbytes : 0u;
            bvo[i] = ok ? (unsigned)(row * p.W + col) * C4 + q : PG_OOB;
            bvs[i] = ok ? (unsigned)(RB * p.W * C4) : 0u;
        }
    }
    auto gather_load = [&]() {                                // the next block of four fine rows: coarse values + sign bytes -> registers
        if constexpr (GM == 2) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const bool in = (unsigned)rrow[i] < (unsigned)p.H;
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c)
                    rpx[i][c] = c < p.rgbC ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rgx, in ? (int)(cvo[i] + (unsigned)c * npix * 4u) : (int)PG_OOB, 0, 0)) : 0.f;
                cvo[i] += cvs[i];
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            gxr[i] = pg_buf_load4(rgx, cvo[i], 0);
            gbr[i] = __builtin_amdgcn_raw_buffer_load_b8(rgb, (int)bvo[i], 0, 0);
            cvo[i] += cvs[i]; bvo[i] += bvs[i];
        }
    };
    auto gather_store = [&](int pos) {                        // x (gmul x LeakyReLU' factor), into ring position pos
        if constexpr (GM == 2) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int sl = i * 256 + tid;
                const bool in = (unsigned)rrow[i] < (unsigned)p.H;              // rows above / below the image: the conv's zero padding
                float o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float a = 0.f;
#pragma unroll
                    for (int c = 0; c < RGB_MAXC; ++c) if (c < p.rgbC) a = fmaf(rpx[i][c], rw[i][k][c], a);
                    const float v = __fadd_rn(__fmul_rn(a, p.rgb_scale), rb[i][k]);      // (two roundings, as fromrgb_fwd_pix_kernel: no contraction)
                    o[k] = in ? (v > 0.f ? v : v * p.rgb_slope) : 0.f;
                }
                const float4 v4 = make_float4(o[0], o[1], o[2], o[3]);
                if (sl < B::SLOTS) *reinterpret_cast<float4*>(lds + (pos * B::SLOTS + sl) * 4) = (cvs[i] ? v4 : make_float4(0.f, 0.f, 0.f, 0.f));
                // the sign bytes of x0 (what fromRGB's backward and this conv's masked backward-data form read), rows of this segment only
                if (p.xsigns && in && rrow[i] >= r0 && rrow[i] < r0 + p.seg_rows && bvs[i])
                    __builtin_amdgcn_raw_buffer_store_b8(pg_sign_byte(v4), rgb, (int)bvo[i], 0, 0);
                bvo[i] += bvs[i];
                rrow[i] += RB;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int sl = i * 256 + tid;
            if (sl < B::SLOTS) {
                const float4 f = pg_sign_factors((unsigned char)gbr[i], p.gslope);
                const float4 v = gxr[i];
                *reinterpret_cast<float4*>(lds + (pos * B::SLOTS + sl) * 4) =
                    make_float4(v.x * (f.x * p.gmul), v.y * (f.y * p.gmul), v.z * (f.z * p.gmul), v.w * (f.w * p.gmul));
            }
        }
    };

    // ---- weights of this lane's cout row (4 qo + j): registers (Cin = 8, optional) or LDS
    float4 wreg[WREG ? 9 : 1][WREG ? C4 : 1];
    if constexpr (WREG) {
#pragma unroll
        for (int tp = 0; tp < 9; ++tp)
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4)
                wreg[tp][c4] = *reinterpret_cast<const float4*>(p.w + ((size_t)(tp * COUT + 4 * qo + j) * CIN + 4 * c4));
    } else {
        for (int e = tid; e < 9 * COUT * C4; e += 256)
            *reinterpret_cast<float4*>(wl + (e / C4) * WS + 4 * (e % C4)) = *reinterpret_cast<const float4*>(p.w + 4 * e);
    }
    const float* wrow = wl + (4 * qo + j) * WS;

    if constexpr (GATH) {
        gather_load(); gather_store(0);
        gather_load(); gather_store(1);
    } else {
        issue_block(0);
        issue_block(1);
    }

    const int rp = wave >> 1;                                     // row pair of the step owned by this wave
    const int col0 = (wave & 1) * 32;
    const lds_cptr lbase = (lds_cptr)lds + (col0 + 4 * qp + j) * 16;

    // ---- epilogue operands: per-image raw buffers, 32-bit offsets.  pix = pixel index inside the image of group 0 at step `it`
    const bool has_mask = (GEN && p.mask) || EPI == EPI_MASK;
    const bool mask_bytes = has_mask && p.mask_bytes;
    const bool has_pnb = (GEN && p.pnb_y) || EPI == EPI_PNB;
    const bool has_pn = (GEN && p.pn_r) || EPI == EPI_PN;
    const bool has_signs = (GEN || EPI == EPI_FWD) && p.ysigns;
    const bool has_pool = GEN && p.ypool;
    const bool y_bytes = GEN && p.y_bytes;
    const bool has_trgb = (XM & 1) != 0 && (EPI == EPI_PN || EPI == EPI_MASK) && p.t_out != nullptr;
    const bool has_fw = (XM & 2) != 0 && EPI == EPI_MASK && COUT == 8 && p.fw_dw != nullptr;
    const bool y_store = !(has_pool && p.pool_only) && !y_bytes && !((has_trgb || has_fw) && p.t_only);
    const __amdgpu_buffer_rsrc_t ry = y_bytes ? pg_make_rsrc((const unsigned char*)p.y + (size_t)n * npix * (COUT / 4), npix * (COUT / 4))
                                              : pg_make_rsrc(p.y + (size_t)n * npix * COUT, npix * COUT * 4u);
    __amdgpu_buffer_rsrc_t rmask = ry, rsig = ry, rpnr = ry, rpnby = ry, rpnbr = ry;
    if (has_mask) rmask = mask_bytes ? pg_make_rsrc((const unsigned char*)p.mask + (size_t)n * npix * (COUT / 4), npix * (COUT / 4))
                                     : pg_make_rsrc((const float*)p.mask + (size_t)n * npix * COUT, npix * COUT * 4u);
    if (has_signs) rsig = pg_make_rsrc(p.ysigns + (size_t)n * npix * (COUT / 4), npix * (COUT / 4));
    if (has_pn) rpnr = pg_make_rsrc(p.pn_r + (size_t)n * npix, npix * 4u);
    if (has_pnb) {
        rpnby = pg_make_rsrc(p.pnb_y + (size_t)n * npix * COUT, npix * COUT * 4u);
        rpnbr = pg_make_rsrc(p.pnb_r + (size_t)n * npix, npix * 4u);
    }
    const unsigned pix0 = (unsigned)((r0 + 2 * rp) * p.W + ow0 + col0 + 4 * qp + j);
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!has_mask && p.bias) bv = *reinterpret_cast<const float4*>(p.bias + 4 * qo);
    // toRGB behind the PixelNorm (round 6): the lane's four couts x the image channels; the QO lanes of a pixel add their partial sums
    float tw[RGB_MAXC][4], tb[RGB_MAXC];
    float aw[4][RGB_MAXC], ab[4], fi[G][RGB_MAXC];               // fromRGB weight gradient: this lane's four couts x image channels, bias sums; the step's image values
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        ab[k] = 0.f;
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c) aw[k][c] = 0.f;
    }
    __amdgpu_buffer_rsrc_t rfimg = ry;
    if (has_fw) rfimg = pg_make_rsrc(p.fw_img + (size_t)n * npix * p.tC, npix * (unsigned)p.tC * 4u);
    __amdgpu_buffer_rsrc_t rtout = ry;
    if (has_trgb) {
        rtout = pg_make_rsrc(p.t_out + (size_t)n * npix * p.tC, npix * (unsigned)p.tC * 4u);
#pragma unroll
        for (int c = 0; c < RGB_MAXC; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) tw[c][k] = c < p.tC ? p.t_w[c * p.t_sc + (4 * qo + k) * p.t_sco] : 0.f;
            tb[c] = (c < p.tC && p.t_b) ? p.t_b[c] : 0.f;
        }
    }

    f32x4 acc[G], acc2[G];
    // MFMAs of one step: ring phase PH (block of the step at ring position PH) and row pair RPAIR are compile-time, so every LDS
    // address below is lbase + constant
    auto compute = [&](auto ph_, auto rp_) {
        constexpr int PH = decltype(ph_)::value, RPAIR = decltype(rp_)::value;
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int tp = 0; tp < ((PG_STRIP_ABL & 4) ? 1 : 9); ++tp) {
            const int dy = tp / 3, dx = tp % 3;
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4) {
                float4 a;
                if constexpr (WREG) a = wreg[tp][c4];
                else a = *reinterpret_cast<const float4*>(wrow + tp * COUT * WS + 4 * c4);
                f32x4 bq[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int m = 2 * RPAIR + g / GPR + dy;                          // row of the 6-row window of the step
                    const int pos = m < RB ? PH : (PH + 1) % NBLK;
                    const int off = ((pos * B::SLOTS) + c4 * (RB * RP) + (m & (RB - 1)) * RP + (g % GPR) * PXG + dx) * 16;
                    bq[g] = *(lds_v4ptr)(lbase + off);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, bq[g][0], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc2[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, bq[g][1], acc2[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, bq[g][2], acc[g], 0, 0, 0);
#pragma unroll
                for (int g = 0; g < G; ++g) acc2[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, bq[g][3], acc2[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] += acc2[g];
    };

    // Operands of the epilogue that do not depend on the MFMAs (LeakyReLU' mask, saved PixelNorm output): fetched BEFORE the MFMAs of
    // the step, so that their latency is not a serial stall of this wave between its MFMAs and its stores
    float4 pm[G];
    unsigned pmb[G];
    float prr[G];
    auto prefetch_epilogue = [&](unsigned pix) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned pg_ = pix + (unsigned)((g / GPR) * p.W + (g % GPR) * PXG);
            if (has_fw) {
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c)
                    fi[g][c] = c < p.tC ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rfimg, (int)(((unsigned)c * npix + pg_) * 4u), 0, 0)) : 0.f;
            }
            if (has_mask) {
                if (mask_bytes) pmb[g] = __builtin_amdgcn_raw_buffer_load_b8(rmask, (int)(pg_ * (COUT / 4) + qo), 0, 0);
                else pm[g] = pg_buf_load4(rmask, (pg_ * COUT + 4 * qo) * 4u, 0);
            } else if (has_pnb) {
                pm[g] = pg_buf_load4(rpnby, (pg_ * COUT + 4 * qo) * 4u, 0);
                prr[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rpnbr, (int)(pg_ * 4u), 0, 0));
            }
        }
    };
    // D register r of this lane = out[pixel][cout 4 qo + r]
    auto epilogue = [&](unsigned pix) {
        float4 ov[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const unsigned pg_ = pix + (unsigned)((g / GPR) * p.W + (g % GPR) * PXG);
            const unsigned off = pg_ * COUT + 4 * qo;                                     // element offset inside the image
            float4 o = make_float4(acc[g][0] * p.scale, acc[g][1] * p.scale, acc[g][2] * p.scale, acc[g][3] * p.scale);
            if (has_mask) {
                float4 f;
                if (mask_bytes) f = pg_sign_factors((unsigned char)pmb[g], p.mask_slope);
                else {
                    const float4 mk = pm[g];
                    f = make_float4(mk.x > 0.f ? 1.f : p.mask_slope, mk.y > 0.f ? 1.f : p.mask_slope,
                                    mk.z > 0.f ? 1.f : p.mask_slope, mk.w > 0.f ? 1.f : p.mask_slope);
                }
                o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
            } else {
                o.x += bv.x; o.y += bv.y; o.z += bv.z; o.w += bv.w;
                o.x = o.x > 0.f ? o.x : o.x * p.slope; o.y = o.y > 0.f ? o.y : o.y * p.slope;
                o.z = o.z > 0.f ? o.z : o.z * p.slope; o.w = o.w > 0.f ? o.w : o.w * p.slope;
                if (has_signs && !(PG_STRIP_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b8(pg_sign_byte(o), rsig, (int)(off >> 2), 0, 0);
            }
            if (has_pnb) {                               // adjoint of the previous layer's (LeakyReLU -> PixelNorm), see ConvP
                const float4 yv = pm[g];
                const float4 gv = make_float4(acc[g][0] * p.scale, acc[g][1] * p.scale, acc[g][2] * p.scale, acc[g][3] * p.scale);
                float dt = (gv.x * yv.x + gv.y * yv.y) + (gv.z * yv.z + gv.w * yv.w);
                dt += __shfl_xor(dt, 4, 64);
                if (QO >= 4) dt += __shfl_xor(dt, 8, 64);
                const float rr = prr[g], mean = dt / (float)COUT;
                o.x = rr * (gv.x - yv.x * mean) * (yv.x > 0.f ? 1.f : p.mask_slope);
                o.y = rr * (gv.y - yv.y * mean) * (yv.y > 0.f ? 1.f : p.mask_slope);
                o.z = rr * (gv.z - yv.z * mean) * (yv.z > 0.f ? 1.f : p.mask_slope);
                o.w = rr * (gv.w - yv.w * mean) * (yv.w > 0.f ? 1.f : p.mask_slope);
            }
            if (has_pn) {                                // PixelNorm over the COUT channels of the pixel: QO lanes (4 apart) share it
                float ssq = (o.x * o.x + o.y * o.y) + (o.z * o.z + o.w * o.w);
                ssq += __shfl_xor(ssq, 4, 64);
                if (QO >= 4) ssq += __shfl_xor(ssq, 8, 64);
                const float rr = rsqrtf(ssq / (float)COUT + p.pn_eps);
                o.x *= rr; o.y *= rr; o.z *= rr; o.w *= rr;
                if (qo == 0 && !(PG_STRIP_ABL & 2)) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rr), rpnr, (int)(pg_ * 4u), 0, 0);
            }
            if (has_fw) {                                // (workgroup-uniform) fromRGB's weight / bias gradient: partial sums over this lane's pixels
                const float ov4[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ab[k] += ov4[k];
#pragma unroll
                    for (int c = 0; c < RGB_MAXC; ++c) aw[k][c] = fmaf(ov4[k], fi[g][c], aw[k][c]);
                }
            }
            if (has_trgb) {                              // (workgroup-uniform) the 1x1 RGB layer on the finished value: toRGB / fromRGB's backward-data
#pragma unroll
                for (int c = 0; c < RGB_MAXC; ++c) {
                    if (c >= p.tC) break;
                    float a = fmaf(o.w, tw[c][3], fmaf(o.z, tw[c][2], fmaf(o.y, tw[c][1], o.x * tw[c][0])));
                    a += __shfl_xor(a, 4, 64);
                    if (QO >= 4) a += __shfl_xor(a, 8, 64);
                    if (qo == 0) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaf(a, p.t_scale, tb[c])), rtout, (int)(((unsigned)c * npix + pg_) * 4u), 0, 0);
                }
            }
            if (!(PG_STRIP_ABL & 2)) {
                if (y_bytes) __builtin_amdgcn_raw_buffer_store_b8(pg_sign_byte(o), ry, (int)(off >> 2), 0, 0);
                else if (y_store) buf_store4(ry, off * 4u, o);
            }
            ov[g] = o;
        }
        if (has_pool) {                              // 2x2 mean: column partner = lane ^ 1, row partner = group g + GPR (same wave)
#pragma unroll
            for (int g = 0; g < G; ++g) {                 // (same order of additions as the tile kernel: columns first, then rows)
                ov[g].x += __shfl_xor(ov[g].x, 1, 64); ov[g].y += __shfl_xor(ov[g].y, 1, 64);
                ov[g].z += __shfl_xor(ov[g].z, 1, 64); ov[g].w += __shfl_xor(ov[g].w, 1, 64);
            }
#pragma unroll
            for (int g = 0; g < GPR; ++g) {
                float4 v = make_float4((ov[g].x + ov[g + GPR].x) * 0.25f, (ov[g].y + ov[g + GPR].y) * 0.25f,
                                       (ov[g].z + ov[g + GPR].z) * 0.25f, (ov[g].w + ov[g + GPR].w) * 0.25f);
                if (j & 1) continue;
                const unsigned pg_ = pix + (unsigned)(g * PXG);
                const unsigned oy = pg_ / (unsigned)p.W, ox = pg_ - oy * (unsigned)p.W;
                const size_t poff = (((size_t)n * (p.H >> 1) + (oy >> 1)) * (p.W >> 1) + (ox >> 1)) * COUT + 4 * qo;
                if (p.pool_other) {
                    const float4 q = *reinterpret_cast<const float4*>(p.pool_other + poff);
                    v.x = fmaf(v.x, p.pool_a, p.pool_b * q.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * q.y);
                    v.z = fmaf(v.z, p.pool_a, p.pool_b * q.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * q.w);
                } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
                *reinterpret_cast<float4*>(p.ypool + poff) = v;
            }
        }
    };

    // Vector-memory instructions a step issues AFTER its DMA and whose completion the next step need not wait for: the stores of
    // the epilogue (the gfx9-family vmcnt retires loads and stores in issue order, so "at most S outstanding" = "everything issued
    // before them is done"; the loads of the epilogue have been consumed by then).
    int nstores = (y_bytes || y_store) ? G : 0;
    if (has_signs) nstores += G;
    if (has_pn) nstores += G;
    if (has_pool) nstores += GPR;
    if (has_trgb) nstores += G * p.tC;
    if (PG_STRIP_ABL & 2) nstores = 0;
    nstores = __builtin_amdgcn_readfirstlane(nstores);
    auto wait_dma = [&]() {
        if constexpr (EPI == EPI_MASK || EPI == EPI_PNB || EPI == EPI_PN) {
            static_assert(G == 2, "");
            if (!has_trgb && !has_fw) {
                if constexpr (EPI == EPI_PN) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            } else switch (nstores) {                      // (+ G stores per image channel of the RGB output, - G when y itself stays unwritten)
                case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
        }
        else switch (nstores) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        }
    };
    unsigned pix = pix0;
    const unsigned pixstep = (unsigned)(RB * p.W);
    auto step = [&](int it, auto ph_) {
        constexpr int PH = decltype(ph_)::value;
        if constexpr (!GATH) {
            if (it == 0 || (PG_STRIP_ABL & 2)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wait_dma();                                      // this wave's share of blocks it, it + 1 has landed ...
        }
        __syncthreads();                                          // ... everyone's has, and nobody still reads block it - 1
        const bool more = it + 2 <= niter;                        // (block niter: the two rows below the segment)
        if constexpr (GATH) { if (more) gather_load(); }
        else if (more && !((PG_STRIP_ABL & 1) && it > 0)) issue_block((PH + 2) % NBLK);
        prefetch_epilogue(pix);
        __builtin_amdgcn_sched_barrier(0);
        if (rp == 0) compute(ph_, std::integral_constant<int, 0>{});
        else compute(ph_, std::integral_constant<int, 1>{});
        if constexpr (GATH) { if (more) gather_store((PH + 2) % NBLK); }
        epilogue(pix);
        pix += pixstep;
    };
    for (int it = 0; it < niter; it += 3) {
        step(it, std::integral_constant<int, 0>{});
        if (it + 1 < niter) step(it + 1, std::integral_constant<int, 1>{});
        if (it + 2 < niter) step(it + 2, std::integral_constant<int, 2>{});
    }
    if (has_fw) {
        // lane = [qp : 3][qo : 1][j : 2] (QO = 2): fold the lanes of one cout quad (bits 0-1, 3-5), then the four waves through LDS (the ring is
        // free after the barrier), one atomic per output and workgroup (as fromrgb_wgrad_small_kernel)
        static_assert(COUT != 8 || QO == 2, "");
        float v[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int c = 0; c < RGB_MAXC; ++c) v[4 * k + c] = aw[k][c];
            v[4 * k + 3] = ab[k];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float a = v[i];
            a += __shfl_xor(a, 1, 64); a += __shfl_xor(a, 2, 64); a += __shfl_xor(a, 8, 64); a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            v[i] = a;
        }
        __syncthreads();
        if ((lane & ~4) == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) lds[(wave * 2 + (lane >> 2)) * 16 + i] = v[i];
        }
        __syncthreads();
        const int q2 = (tid >> 4) & 1, i = tid & 15, k = i >> 2, c = i & 3, co = 4 * q2 + k;
        auto commit = [&](float a) {
            if (c < 3) { if (c < p.tC) atomicAdd(p.fw_dw + co * p.tC + c, a * p.fw_scale); }
            else if (p.fw_db) atomicAdd(p.fw_db + co, a);
        };
        float a = 0.f;
        if (tid < 32) a = (lds[(0 * 2 + q2) * 16 + i] + lds[(1 * 2 + q2) * 16 + i]) + (lds[(2 * 2 + q2) * 16 + i] + lds[(3 * 2 + q2) * 16 + i]);
        // (32 replicas of the sums folded by the last workgroup to arrive -- 2304 workgroups commit to ONE 128-byte line -- were built and
        //  measured: 10.027 vs 10.009 ms per step with the plain commit, docs/experiments_r6.md §7)
        if (tid < 32) commit(a);
    }
}

template <int COUT, int CIN, int EPI, bool WREG, bool GATH>
__global__ __launch_bounds__(256, GATH ? 2 : (WREG ? 3 : 4)) void conv_strip_kernel(SArgs p)
{
    conv_strip_body<COUT, CIN, EPI, WREG, GATH ? 1 : 0>(p);
}

// 8 -> 8 with the RGB-side extras in the epilogue (XM above): XM 1 keeps the weights in registers, XM 3 (weight gradient) in LDS
template <int EPI, bool WREG, int XM>
__global__ __launch_bounds__(256, 3) void conv_strip_x_kernel(SArgs p)
{
    conv_strip_body<8, 8, EPI, WREG, 0, XM>(p);
}

// 8 -> 8 with the block's fromRGB layer in the gather (GM 2)
template <int EPI>
__global__ __launch_bounds__(256, 2) void conv_strip_rgb_kernel(SArgs p)
{
    conv_strip_body<8, 8, EPI, false, 2>(p);
}

// Dynamic LDS above 48 KB needs the function attribute; it is a per-device property of the loaded code object, so it is set on
// every launch that needs it (an idempotent host-side call: no per-process flag that a second GPU or a second thread could miss).
inline int strip_set_smem(const void* fn, size_t smem)
{
    if (smem <= 48 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    return e == hipSuccess ? 0 : (int)e;
}

template <int COUT, int CIN, int EPI, bool WREG, bool GATH = false>
int launch_strip(const SArgs& a, int N, hipStream_t s, char* name, size_t name_len)
{
    using B = Blk<CIN>;
    const size_t smem = (size_t)NBLK * B::SLOTS * 16 + (WREG ? 0 : (size_t)9 * COUT * (CIN == 16 ? CIN + 4 : CIN) * 4);
    auto kern = conv_strip_kernel<COUT, CIN, EPI, WREG, GATH>;
    if (int rc = strip_set_smem(reinterpret_cast<const void*>(kern), smem); rc) return rc;
    snprintf(name, name_len, "conv_strip_kernel<%d, %d, %d, %s, %s>", COUT, CIN, EPI, WREG ? "true" : "false", GATH ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * a.strips * a.segs)), dim3(256), smem, s, a);
    return (int)hipGetLastError();
}

template <int EPI, bool WREG, int XM>
int launch_strip_x(const SArgs& a, int N, hipStream_t s, char* name, size_t name_len)
{
    const size_t smem = (size_t)NBLK * Blk<8>::SLOTS * 16 + (WREG ? 0 : (size_t)9 * 8 * 8 * 4);
    auto kern = conv_strip_x_kernel<EPI, WREG, XM>;
    if (int rc = strip_set_smem(reinterpret_cast<const void*>(kern), smem); rc) return rc;
    snprintf(name, name_len, "conv_strip_x_kernel<%d, %s, %d>", EPI, WREG ? "true" : "false", XM);
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * a.strips * a.segs)), dim3(256), smem, s, a);
    return (int)hipGetLastError();
}

template <int COUT, int CIN, bool WREG>
int launch_strip_epi(const SArgs& a, int epi, int N, hipStream_t s, char* name, size_t name_len)
{
    switch (epi) {
        case EPI_FWD: return launch_strip<COUT, CIN, EPI_FWD, WREG>(a, N, s, name, name_len);
        case EPI_MASK: return launch_strip<COUT, CIN, EPI_MASK, WREG>(a, N, s, name, name_len);
        case EPI_PN: return launch_strip<COUT, CIN, EPI_PN, WREG>(a, N, s, name, name_len);
        case EPI_PNB: return launch_strip<COUT, CIN, EPI_PNB, WREG>(a, N, s, name, name_len);
        default: return launch_strip<COUT, CIN, EPI_GENERIC, WREG>(a, N, s, name, name_len);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight (+ bias) gradient of the same layers, row-streaming form:  dW[tap][co][ci] += scale * sum_pixels gz[p][co] x[p + tap][ci]
// (the autograd of F.conv2d network.py:34 wrt its weight; replaces conv_wgrad_thin_kernel where the shape allows).
// The tile kernel stages 16 x 4-pixel tiles (2.25x halo over-fetch of x from L2, ~8000 cycles per tile for ~1200 of MFMA work, the
// k-step -> LDS address map re-derived per tile).  Here a workgroup walks down a strip of 64 columns: x rows (with the one-column
// halo) and gz rows enter LDS once through LDS-DMA rings, wave w contracts row w of the step with v_mfma_f32_4x4x1_16B_f32
// (block = (cout quad, cin quad, pixel slot), 16 / NB pixels per instruction), and the nine tap accumulators stay in registers
// for the WHOLE strip: one cross-wave reduction and one atomic commit per workgroup.
//   * pixel slots walk CONSECUTIVE pixels of their 16- (32-) pixel range of the row, so of the nine x operands of a k-step six
//     are the previous k-steps' registers: 4 ds_read_b32 per 9 MFMAs instead of 10;
//   * LDS: plane q = channels 4q..4q+3; inside a plane row the ranges are interleaved (slot = PPM * u + s, each range with its own
//     two halo pixels: the DMA source address is per lane, duplicates cost nothing) and the plane stride is 32 bytes off a
//     multiple of 128: the 32-lane groups of every ds_read_b32 hit 32 different banks.
struct WArgs { const float* x; const float* gz; float* dw; float* db; float scale; int H, W, ups, strips, segs, seg_rows;
               const unsigned char* gbytes; float gmul, gslope; };   // pool adjoint in the gz gather (pg_conv2d_wgrad_unpooled_nhwc)

template <int CO, int CI, bool GATH>
__global__ __launch_bounds__(256, 2) void wgrad_strip_kernel(WArgs p)
{
    constexpr int TAPS = 9;
    constexpr int QO = CO / 4, QI = CI / 4, NB = QO * QI, PPM = 16 / NB, RNG = SW / PPM;   // NB blocks per pixel, PPM pixel slots per MFMA
    static_assert(NB == 4 || NB == 8, "8x8, 16x8 or 8x16 channels");
    constexpr int XU = RNG + 2, XRS = PPM * XU;              // x: pixels of a range incl. halo; slots of a plane row (72 / 68)
    constexpr int XBS = 320, GBS = RB * SW;                  // slots of one (plane, block): x 4 XRS <= 5 DMA instructions; gz 4 x 64
    static_assert(RB * XRS <= XBS, "");
    constexpr int XRING = 3, GRING = 2;
    constexpr int XPS = XRING * XBS + 2, GPS = GRING * GBS + 2;   // plane strides (slots): + 32 bytes -> planes 8 banks apart
    constexpr int XNWI = QI * 5, GNWI = QO * RB, XNI = (XNWI + 3) / 4, GNI = (GNWI + 3) / 4;
    extern __shared__ __align__(16) float lds[];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    constexpr unsigned GLDS = (unsigned)QI * XPS * 16;       // byte offset of the gz ring
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int blk = lane >> 2, i4 = lane & 3;
    const int hi = blk % QI, ho = (blk / QI) % QO, slot = blk / NB;

    int t = (int)pg_xcd_remap(blockIdx.x, gridDim.x);
    const int strip = __builtin_amdgcn_readfirstlane(t % p.strips); t /= p.strips;
    const int seg = __builtin_amdgcn_readfirstlane(t % p.segs), n = __builtin_amdgcn_readfirstlane(t / p.segs);
    const int r0 = seg * p.seg_rows, ow0 = strip * SW;
    const int niter = p.seg_rows / RB;
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;

    // ---- DMA offsets.  x: instruction ii = (plane q, part 0..4) fills slots [part * 64, +64) of the (plane, block) area
    const unsigned xrowbytes = 4u * (unsigned)(xW * CI), grow = 4u * (unsigned)(p.W * CO);
    unsigned xvoff[XNI], xvstep[XNI], xdst[XNI], gvoff[GNI], gdst[GNI];
    const unsigned xblkstep = (p.ups ? 2u : 4u) * xrowbytes;
#pragma unroll
    for (int i = 0; i < XNI; ++i) {
        const int ii = i * 4 + wave, q = ii / 5, part = ii - q * 5;
        const int idx = part * 64 + lane;
        const int m = idx / XRS, e = idx - m * XRS, u = e / PPM, s = e - u * PPM;
        const int col = ow0 + RNG * s + u - 1, row = r0 - 1 + m;
        const bool ok = ii < XNWI && idx < RB * XRS && (unsigned)col < (unsigned)p.W;
        xvoff[i] = ok ? (unsigned)(p.ups ? (row >> 1) : row) * xrowbytes + 4u * (unsigned)((p.ups ? (col >> 1) : col) * CI + 4 * q) : PG_OOB;
        xvstep[i] = ok ? xblkstep : 0u;
        xdst[i] = lds0 + (unsigned)(q * XPS + part * 64) * 16u;                  // + ring position * XBS * 16 (wave-uniform)
    }
#pragma unroll
    for (int i = 0; i < GNI; ++i) {                              // gz: instruction jj = (plane q, row m): one row of 64 slots
        const int jj = i * 4 + wave, q = jj / RB, m = jj - q * RB;
        const int u = lane / PPM, s = lane - u * PPM;
        const int col = ow0 + RNG * s + u, row = r0 + m;
        gvoff[i] = jj < GNWI ? (unsigned)row * grow + 4u * (unsigned)(col * CO + 4 * q) : PG_OOB;
        gdst[i] = lds0 + GLDS + (unsigned)(q * GPS + m * SW) * 16u;
    }
    const size_t ximg = (size_t)xH * xW * CI, gimg = GATH ? (size_t)(p.H >> 1) * (p.W >> 1) * CO : (size_t)p.H * p.W * CO;
    const pg_u32x4 rxs = rsrc_words(p.x + (size_t)n * ximg, (unsigned)(ximg * 4));
    const pg_u32x4 rgs = rsrc_words(p.gz + (size_t)n * gimg, (unsigned)(gimg * 4));
    // GATH: gz[h][w][c] = gmul * g[h/2][w/2][c] * lrelu'(sign byte of a2[h][w][c]) is computed on the way into the ring: thread tid
    // owns slot tid of every (plane, block) area; values are fetched one block ahead and written with ds_write_b128 after the MFMAs
    constexpr int NGZ = GATH ? QO : 1;
    unsigned cvo[NGZ], bvo[NGZ];
    float4 gxr[NGZ];
    unsigned gbr[NGZ];
    __amdgpu_buffer_rsrc_t rgc = pg_make_rsrc(p.gz + (size_t)n * gimg, (unsigned)(gimg * 4)), rgb = rgc;
    if constexpr (GATH) {
        rgb = pg_make_rsrc(p.gbytes + (size_t)n * p.H * p.W * QO, (unsigned)(p.H * p.W * QO));
        const int m = tid >> 6, e = tid & 63, u = e / PPM, s = e - u * PPM;
        const int col = ow0 + RNG * s + u, row = r0 + m;
#pragma unroll
        for (int q = 0; q < NGZ; ++q) {
            cvo[q] = 4u * (unsigned)(((row >> 1) * (p.W >> 1) + (col >> 1)) * CO + 4 * q);
            bvo[q] = (unsigned)(row * p.W + col) * QO + q;
        }
    }
    auto gather_load = [&]() {
#pragma unroll
        for (int q = 0; q < NGZ; ++q) {
            gxr[q] = pg_buf_load4(rgc, cvo[q], 0);
            gbr[q] = __builtin_amdgcn_raw_buffer_load_b8(rgb, (int)bvo[q], 0, 0);
            cvo[q] += 2u * 4u * (unsigned)((p.W >> 1) * CO); bvo[q] += (unsigned)(RB * p.W * QO);
        }
    };
    auto gather_store = [&](int pos) {
#pragma unroll
        for (int q = 0; q < NGZ; ++q) {
            const float4 f = pg_sign_factors((unsigned char)gbr[q], p.gslope);
            const float4 v = gxr[q];
            *reinterpret_cast<float4*>(lds + GLDS / 4 + ((q * GPS + pos * GBS) + tid) * 4) =
                make_float4(v.x * (f.x * p.gmul), v.y * (f.y * p.gmul), v.z * (f.z * p.gmul), v.w * (f.w * p.gmul));
        }
    };
    auto dma16 = [&](const pg_u32x4& rs, unsigned vo, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(rs), "s"(dst) : "memory");
    };
    auto issue_x = [&](int pos) {
#pragma unroll
        for (int i = 0; i < XNI; ++i) {
            if (i * 4 + wave < XNWI) dma16(rxs, xvoff[i], __builtin_amdgcn_readfirstlane(xdst[i]) + (unsigned)pos * (XBS * 16u));
            xvoff[i] += xvstep[i];
        }
    };
    auto issue_g = [&](int pos) {
#pragma unroll
        for (int i = 0; i < GNI; ++i) {
            if (i * 4 + wave < GNWI) dma16(rgs, gvoff[i], __builtin_amdgcn_readfirstlane(gdst[i]) + (unsigned)pos * (GBS * 16u));
            gvoff[i] += 4u * grow;
        }
    };
    issue_x(0);
    if constexpr (GATH) { gather_load(); gather_store(0); } else issue_g(0);
    issue_x(1);

    f32x4 acc[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp) acc[tp] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    typedef __attribute__((address_space(3))) const float* lds_fptr;
    const lds_cptr xl = (lds_cptr)lds + (hi * XPS + slot) * 16 + i4 * 4;
    const lds_cptr gl = (lds_cptr)lds + GLDS + (ho * GPS + slot) * 16 + i4 * 4;

    for (int it = 0; it < niter; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // x blocks it, it + 1 and gz block it have landed (this wave's share) ...
        __syncthreads();                                          // ... everyone's; nobody still reads what the next DMAs overwrite
        if (it + 2 <= niter) issue_x((it + 2) % XRING);
        if constexpr (GATH) { if (it + 1 < niter) gather_load(); }
        else if (it + 1 < niter) issue_g((it + 1) % GRING);
        __builtin_amdgcn_sched_barrier(0);
        // row w of the step: ring rows of its three x rows (wave-uniform), then lane base + immediate offsets
        const int rr0 = (RB * (it % XRING) + wave);
        lds_cptr xb[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            int rr = rr0 + dy; if (rr >= RB * XRING) rr -= RB * XRING;
            xb[dy] = xl + ((rr / RB) * XBS + (rr % RB) * XRS) * 16;
        }
        const lds_cptr gb = gl + ((it % GRING) * GBS + wave * SW) * 16;
        float bcol[3][3];                                         // bcol[dy][c % 3] = x[row + dy][column c] of this lane's range
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            bcol[dy][0] = *(lds_fptr)(xb[dy] + 0 * PPM * 16);
            bcol[dy][1] = *(lds_fptr)(xb[dy] + 1 * PPM * 16);
        }
#pragma unroll
        for (int ks = 0; ks < RNG; ++ks) {
            const float a = *(lds_fptr)(gb + ks * PPM * 16);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) bcol[dy][(ks + 2) % 3] = *(lds_fptr)(xb[dy] + (ks + 2) * PPM * 16);
            bsum += a;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, bcol[dy][(ks + dx) % 3], acc[dy * 3 + dx], 0, 0, 0);
        }
        if constexpr (GATH) { if (it + 1 < niter) gather_store((it + 1) % GRING); }
    }

    // ---- sum the pixel slots (lanes 4 NB apart), then the 4 waves through LDS, then ONE commit per workgroup
#pragma unroll
    for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[tp][r];
            v += __shfl_xor(v, 32, 64);
            if (PPM >= 4) v += __shfl_xor(v, 16, 64);
            acc[tp][r] = v;
        }
    bsum += __shfl_xor(bsum, 32, 64);
    if (PPM >= 4) bsum += __shfl_xor(bsum, 16, 64);
    constexpr int NL = 4 * NB, NE = TAPS * 4 + 1;               // lanes holding distinct results; values per lane: [tap * 4 + r | bias]
    __syncthreads();                                             // (the rings are free: the reduction scratch aliases them)
    float* red = lds;                                            // [wave][NE][NL]
    if (lane < NL) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * NE + tp * 4 + r) * NL + lane] = acc[tp][r];
        red[(wave * NE + TAPS * 4) * NL + lane] = bsum;
    }
    __syncthreads();
    for (int e = tid; e < NE * NL; e += 256) {
        const int l = e % NL, ve = e / NL;
        const float v = (red[e] + red[e + NE * NL]) + (red[e + 2 * NE * NL] + red[e + 3 * NE * NL]);
        const int b_ = l >> 2, jj = l & 3;
        const int hi_ = b_ % QI, ho_ = (b_ / QI) % QO;
        if (ve < TAPS * 4) {                                     // D register r of lane 4 b + j = dW[tap][4 ho + r][4 hi + j]
            const int tp = ve >> 2, r = ve & 3;
            atomicAdd(p.dw + ((size_t)(tp * CO + 4 * ho_ + r) * CI + 4 * hi_ + jj), v * p.scale);
        } else if (p.db && hi_ == 0) {                           // lane (ho, hi = 0, i) carries sum_p gz[p][4 ho + i]
            atomicAdd(p.db + 4 * ho_ + jj, v);
        }
    }
}

template <int CO, int CI, bool GATH = false>
int launch_wgrad_strip_t(const WArgs& a, int N, hipStream_t s, char* name, size_t name_len)
{
    constexpr int QO = CO / 4, QI = CI / 4;
    const size_t smem = ((size_t)QI * (3 * 320 + 2) + (size_t)QO * (2 * RB * SW + 2)) * 16;
    auto kern = wgrad_strip_kernel<CO, CI, GATH>;
    if (int rc = strip_set_smem(reinterpret_cast<const void*>(kern), smem); rc) return rc;
    snprintf(name, name_len, "wgrad_strip_kernel<%d, %d, %s>", CO, CI, GATH ? "true" : "false");
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * a.strips * a.segs)), dim3(256), smem, s, a);
    return (int)hipGetLastError();
}

}  // namespace

int pgk::launch_conv_strip(ConvP& p, hipStream_t s, char* name, size_t name_len)
{
    static const int seg_env = getenv("PG_STRIP_SEG") ? atoi(getenv("PG_STRIP_SEG")) : 0;
    // (8 -> 8 weights in registers: 0.5-3 % slower alone on the device in round 3, but faster INSIDE the two-stream step, where the LDS pipe is
    //  shared with the weight-gradient kernels -- round 4, same box, three pairs: 10.71 / 10.68 / 10.68 vs 10.81 / 10.82 / 10.72 ms per step)
    static const int wreg_env = getenv("PG_STRIP_WREG") ? atoi(getenv("PG_STRIP_WREG")) : 1;
    static const int epi_env = getenv("PG_STRIP_EPI") ? atoi(getenv("PG_STRIP_EPI")) : -1;      // 0: always the generic epilogue (A/B)
    if (p.KS != 3 || p.pad != 1 || p.yup || p.ksplit != 1) return PG_E_UNSUP;
    if ((p.Wout % SW) || (p.Hout % 16) || p.Hout != p.Hin || p.Wout != p.Win) return PG_E_UNSUP;
    if (!(p.Cout == 8 && (p.Cin == 8 || p.Cin == 16))) return PG_E_UNSUP;       // (16 couts: the tile / Winograd kernels keep those layers)
    if ((long long)p.Hin * p.Win * 16 * 4 >= (1ll << 31)) return PG_E_UNSUP;     // 32-bit byte offsets inside an image
    // rows per workgroup: long enough to amortise the two-block prologue, short enough for >= 3 workgroups per CU
    int seg = seg_env > 0 ? seg_env : 64;
    while (seg > 16 && ((long long)p.N * (p.Wout / SW) * (p.Hout / seg) < 768 || (p.Hout % seg))) seg >>= 1;
    if (seg < 16 || (seg % RB) || (p.Hout % seg)) return PG_E_UNSUP;
    SArgs a{};                                   // (the fields this entry does not use -- fromRGB gather, toRGB epilogue -- must read as "off")
    a.x = p.x; a.w = p.w; a.bias = p.bias; a.mask = p.mask; a.y = p.y;
    a.ysigns = p.ysigns; a.pn_r = p.pn_r; a.pnb_y = p.pnb_y; a.pnb_r = p.pnb_r; a.ypool = p.ypool; a.pool_other = p.pool_other;
    a.scale = p.scale; a.slope = p.slope; a.mask_slope = p.mask_slope; a.pn_eps = p.pn_eps; a.pool_a = p.pool_a; a.pool_b = p.pool_b;
    a.H = p.Hout; a.W = p.Wout; a.ups = p.ups; a.mask_bytes = p.mask_bytes; a.y_bytes = p.y_bytes; a.pool_only = p.pool_only;
    a.strips = p.Wout / SW; a.segs = p.Hout / seg; a.seg_rows = seg;
    a.gbytes = p.gbytes; a.gmul = p.gmul; a.gslope = p.gslope;
    if (p.gbytes) {                              // pool adjoint in the gather (backward-data conv of a DBlock's c2): masked or plain
        if (!p.ups || p.bias || p.ypool || p.y_bytes || p.pnb_y || p.pn_r || p.ysigns) return PG_E_UNSUP;
        if (p.Cin == 8) return p.mask ? launch_strip<8, 8, EPI_MASK, false, true>(a, p.N, s, name, name_len)
                                      : launch_strip<8, 8, EPI_FWD, false, true>(a, p.N, s, name, name_len);
        return p.mask ? launch_strip<8, 16, EPI_MASK, false, true>(a, p.N, s, name, name_len)
                      : launch_strip<8, 16, EPI_FWD, false, true>(a, p.N, s, name, name_len);
    }
    int epi = EPI_GENERIC;
    if (!p.ypool && !p.y_bytes) {
        if (p.pnb_y && !p.mask && !p.pn_r && !p.ysigns) epi = EPI_PNB;
        else if (p.pn_r && !p.mask && !p.pnb_y && !p.ysigns) epi = EPI_PN;
        else if (p.mask && !p.pnb_y && !p.pn_r) epi = EPI_MASK;
        else if (!p.mask && !p.pnb_y && !p.pn_r) epi = EPI_FWD;
    }
    if (epi_env == 0) epi = EPI_GENERIC;
    // 8 -> 8: the weights of a lane's cout row fit in registers (72 VGPRs, three waves per SIMD) or stay in LDS (four+ waves per SIMD)
    if (p.Cin == 8) return wreg_env ? launch_strip_epi<8, 8, true>(a, epi, p.N, s, name, name_len)
                                    : launch_strip_epi<8, 8, false>(a, epi, p.N, s, name, name_len);
    return launch_strip_epi<8, 16, false>(a, epi, p.N, s, name, name_len);
}

int pgk::launch_conv_strip_fromrgb(const float* img, const float* rgb_w, const float* rgb_b, float rgb_scale, float rgb_slope,
                                   unsigned char* x_signs, const float* w, const float* bias, float* y, unsigned char* y_signs,
                                   int N, int C, int H, int W, int Cmid, int Cout, float scale, float slope,
                                   hipStream_t s, char* name, size_t name_len)
{
    if (C > RGB_MAXC || Cmid != 8 || Cout != 8 || (W % SW) || (H % 16)) return PG_E_UNSUP;
    if ((long long)H * W * 16 * 4 >= (1ll << 31)) return PG_E_UNSUP;            // 32-bit byte offsets inside an image
    int seg = 64;                                                                // (as launch_conv_strip)
    while (seg > 16 && ((long long)N * (W / SW) * (H / seg) < 768 || (H % seg))) seg >>= 1;
    if (seg < 16 || (seg % RB) || (H % seg)) return PG_E_UNSUP;
    SArgs a{};
    a.x = img; a.w = w; a.bias = bias; a.y = y; a.ysigns = y_signs;
    a.scale = scale; a.slope = slope; a.mask_slope = 1.f; a.pool_a = 1.f;
    a.H = H; a.W = W; a.strips = W / SW; a.segs = H / seg; a.seg_rows = seg;
    a.rgb_w = rgb_w; a.rgb_b = rgb_b; a.rgb_scale = rgb_scale; a.rgb_slope = rgb_slope; a.rgbC = C; a.xsigns = x_signs;
    const size_t smem = (size_t)NBLK * Blk<8>::SLOTS * 16 + (size_t)9 * 8 * 8 * 4;
    auto kern = conv_strip_rgb_kernel<EPI_FWD>;
    if (int rc = strip_set_smem(reinterpret_cast<const void*>(kern), smem); rc) return rc;
    snprintf(name, name_len, "conv_strip_rgb_kernel<%d>", (int)EPI_FWD);
    hipLaunchKernelGGL(kern, dim3((unsigned)(N * a.strips * a.segs)), dim3(256), smem, s, a);
    return (int)hipGetLastError();
}

int pgk::launch_conv_strip_pn_torgb(const float* x, const float* w, const float* bias, float* y, float* r,
                                    const float* t_w, const float* t_b, float t_scale, float* img,
                                    int N, int C, int H, int W, int Cin, int Cout, float scale, float slope, float eps,
                                    hipStream_t s, char* name, size_t name_len)
{
    static const int wreg_env = getenv("PG_STRIP_WREG") ? atoi(getenv("PG_STRIP_WREG")) : 1;
    if (C < 1 || C > RGB_MAXC || Cin != 8 || Cout != 8 || (W % SW) || (H % 16)) return PG_E_UNSUP;
    if ((long long)H * W * 16 * 4 >= (1ll << 31)) return PG_E_UNSUP;
    int seg = 64;                                                                // (as launch_conv_strip)
    while (seg > 16 && ((long long)N * (W / SW) * (H / seg) < 768 || (H % seg))) seg >>= 1;
    if (seg < 16 || (seg % RB) || (H % seg)) return PG_E_UNSUP;
    SArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.y = y; a.pn_r = r; a.pn_eps = eps;
    a.scale = scale; a.slope = slope; a.mask_slope = 1.f; a.pool_a = 1.f;
    a.H = H; a.W = W; a.strips = W / SW; a.segs = H / seg; a.seg_rows = seg;
    a.t_out = img; a.t_w = t_w; a.t_b = t_b; a.t_scale = t_scale; a.tC = C; a.t_sc = Cout; a.t_sco = 1;
    return wreg_env ? launch_strip_x<EPI_PN, true, 1>(a, N, s, name, name_len) : launch_strip_x<EPI_PN, false, 1>(a, N, s, name, name_len);
}

int pgk::launch_conv_strip_masked_rgb_bwd(const float* gz, const float* wt, const unsigned char* mask_bytes, float mask_slope, float* y,
                                          const float* rgb_w, float rgb_scale, float* gimg,
                                          const float* img, float* rgb_dw, float* rgb_db,
                                          int N, int C, int H, int W, int Cin, int Cout, float scale,
                                          hipStream_t s, char* name, size_t name_len)
{
    // (with the weight gradient's 16 accumulators the register-weights form would drop to two waves per SIMD: weights in LDS then)
    static const int wreg_env0 = getenv("PG_STRIP_WREG") ? atoi(getenv("PG_STRIP_WREG")) : 1;
    const int wreg_env = rgb_dw ? 0 : wreg_env0;
    if (C < 1 || C > RGB_MAXC || Cin != 8 || Cout != 8 || (W % SW) || (H % 16)) return PG_E_UNSUP;
    if ((long long)H * W * 16 * 4 >= (1ll << 31)) return PG_E_UNSUP;
    int seg = 64;                                                                // (as launch_conv_strip)
    while (seg > 16 && ((long long)N * (W / SW) * (H / seg) < 768 || (H % seg))) seg >>= 1;
    if (seg < 16 || (seg % RB) || (H % seg)) return PG_E_UNSUP;
    SArgs a{};
    a.x = gz; a.w = wt; a.mask = mask_bytes; a.mask_bytes = 1; a.y = y;
    a.scale = scale; a.slope = 1.f; a.mask_slope = mask_slope; a.pool_a = 1.f;
    a.H = H; a.W = W; a.strips = W / SW; a.segs = H / seg; a.seg_rows = seg;
    a.t_out = gimg; a.t_w = rgb_w; a.t_b = nullptr; a.t_scale = rgb_scale; a.tC = C; a.t_sc = 1; a.t_sco = C; a.t_only = y ? 0 : 1;
    a.fw_img = img; a.fw_dw = rgb_dw; a.fw_db = rgb_db; a.fw_scale = rgb_scale;
    if (rgb_dw) return launch_strip_x<EPI_MASK, false, 3>(a, N, s, name, name_len);
    (void)wreg_env;                              // (the register-weights form of this one sits at the 168-VGPR cap of three waves per SIMD and spills)
    return launch_strip_x<EPI_MASK, false, 1>(a, N, s, name, name_len);
}

int pgk::launch_wgrad_strip(WgP& p, hipStream_t s, char* name, size_t name_len)
{
    static const int seg_env = getenv("PG_WSTRIP_SEG") ? atoi(getenv("PG_WSTRIP_SEG")) : 0;
    if (p.pad != 1 || p.Hout != p.Hin || p.Wout != p.Win || (p.gbytes && p.ups)) return PG_E_UNSUP;
    if ((p.Wout % SW) || (p.Hout % 16)) return PG_E_UNSUP;
    if (!((p.Cout == 8 && p.Cin == 8) || (p.Cout == 16 && p.Cin == 8) || (p.Cout == 8 && p.Cin == 16))) return PG_E_UNSUP;
    if ((long long)p.Hin * p.Win * 16 * 4 >= (1ll << 31)) return PG_E_UNSUP;     // 32-bit byte offsets inside an image
    // one commit of |dW| atomics per workgroup: few, long-lived workgroups -- ~1000 of them while the segments stay >= 64 rows
    // (measured at 1024^2: 9 images 128 rows = 1152 workgroups 143 us, 256 rows 160 us, 64 rows 147 us), >= 512 below that
    int seg = seg_env > 0 ? seg_env : 256;
    auto tasks = [&](int sg) { return (long long)p.N * (p.Wout / SW) * (p.Hout / sg); };
    while (seg > 64 && (seg > p.Hout || (p.Hout % seg) || tasks(seg) < 1024)) seg >>= 1;
    while (seg > 16 && (seg > p.Hout || (p.Hout % seg) || tasks(seg) < 512)) seg >>= 1;
    if (seg < 16 || (seg % RB) || (p.Hout % seg)) return PG_E_UNSUP;
    WArgs a;
    a.x = p.x; a.gz = p.gz; a.dw = p.dw; a.db = p.db; a.scale = p.scale;
    a.H = p.Hout; a.W = p.Wout; a.ups = p.ups; a.strips = p.Wout / SW; a.segs = p.Hout / seg; a.seg_rows = seg;
    a.gbytes = p.gbytes; a.gmul = p.gmul; a.gslope = p.gslope;
    if (p.gbytes) {
        if (p.Cout == 8 && p.Cin == 8) return launch_wgrad_strip_t<8, 8, true>(a, p.N, s, name, name_len);
        if (p.Cout == 16 && p.Cin == 8) return launch_wgrad_strip_t<16, 8, true>(a, p.N, s, name, name_len);
        return launch_wgrad_strip_t<8, 16, true>(a, p.N, s, name, name_len);
    }
    if (p.Cout == 8 && p.Cin == 8) return launch_wgrad_strip_t<8, 8>(a, p.N, s, name, name_len);
    if (p.Cout == 16 && p.Cin == 8) return launch_wgrad_strip_t<16, 8>(a, p.N, s, name, name_len);
    return launch_wgrad_strip_t<8, 16>(a, p.N, s, name, name_len);
}
