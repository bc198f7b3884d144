// Row-streaming Winograd F(2x2,3x3) convolution for the thin 3x3 layers of the high-resolution stages (gfx950 / CDNA4):
// 8->16 at 1024^2, 16->16 / 16->32 / 32->16 at 512^2, 32->32 / 32->64 at 256^2 -- forward (network.py:33-36), backward-data and the
// gradient-penalty tangent map, with every fused epilogue of the tile kernel (conv_wino.hip: the epilogues are shared, wino_epi.h).
//
// Why: the tile kernel (conv_wino2_kernel) gives such a layer ONE to FOUR 8-channel K chunks per workgroup; round 3 measured the
// 8->16 launch at 1024^2 as 36 864 workgroups of 12.3 k cycles each, a third of it prologue (argument load, block decode, DMA
// descriptors, first DMA round trip), i.e. bound by workgroup lifetime x rounds at 2.1 TB/s and 42 % MFMA-busy, far from both roofs.
// Here a workgroup owns a STRIP of 64 output columns (32 tile columns) and walks down `seg` rows of one image, two tile rows (four
// output rows) per step:
//   * one prologue per strip segment; the Winograd-domain weights of the workgroup's couts (64 B x Cin x couts: 8 KB for 8->16,
//     64 KB for 32->32) are staged ONCE and stay in LDS;
//   * input rows enter LDS exactly once per strip (66 of 64 columns, no vertical re-read) by LDS-DMA into a ring of FIVE ROW PAIRS
//     (the step reads three pairs = its 6-row window while the two pairs of the next step land), ONE barrier per step, no staging
//     registers; rows above / below the image are out of the per-image buffer's records (hardware zero fill);
//   * the ring image is quad-planar like the tile kernel's ([row pair][plane q = channels 4q..4q+3][2 rows][67 slots of 16 bytes])
//     with ODD row pairs shifted by one slot (applied through the DMA source addresses): the 16 tiles of a wave -- 2 tile rows x 8
//     tile columns, 2 pixels apart -- then cover the 256-byte bank row exactly once per ds_read_b64;
//   * MFMA mapping, accumulation order and lane layout of conv_wino2_kernel: v_mfma_f32_16x16x4_f32, A = U (16 couts), B = the
//     lane-transformed patch of its tile (lane = (tile li, channel pair kk)), 32 MFMAs per 8-channel chunk and cout block; results
//     agree with the tile kernel to the last bit (same sums in the same order).
// Bounds per 64-tile step (8->16: 1024 MFMA cycles per wave against ~1200 VALU cycles of output transform + pooled / sign-byte
// epilogue; 16->16 and wider: MFMA / HBM), see DESIGN.md.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pggan_hip.h"
#include "bufload.h"
#include "convp.h"
#include "wino_epi.h"

namespace {

using namespace pgw;

constexpr int SW = 64;            // output columns of a strip
constexpr int RPW = SW + 3;       // slots of a ring row: 66 pixels (one halo column on each side) + the one-slot shift of odd row pairs
constexpr int NPAIR = 5;          // ring of row pairs

template <int CIN> struct Ring {
    static constexpr int C4 = CIN / 4;
    static constexpr int USED = C4 * 2 * RPW;                 // slots of a row pair that carry data: [plane q][row 0/1][slot]
    static constexpr int NWI = (USED + 63) / 64;              // DMA wave-instructions per row pair
    static constexpr int NI = (NWI + 3) / 4;                  // ... per thread (instruction ii = i * 4 + wave)
    static constexpr int PITCH = NWI * 64;                    // slots between row pairs
    static constexpr int BYTES = NPAIR * PITCH * 16;
};

typedef float v2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const char* lds_cptr;
typedef __attribute__((address_space(3))) const volatile v2* lds_v2ptr;

// Epilogues.  SE_GENERIC: everything conv_wino2_kernel fuses, selected at run time (wino_epi.h); the others are the forms the train
// step launches on these layers, as template variants with one raw buffer per image, 32-bit offsets that advance by a constant per
// step, their inputs (mask bytes, fade-in partner) fetched BEFORE the MFMAs of the step, and a known number of stores, so that the
// next step waits for its DMA only (counted vmcnt: gfx9 retires vector-memory instructions in issue order, the stores of a step
// drain under the MFMAs of the next).  Same arithmetic, in the same order, as the general epilogue.
enum { SE_GENERIC = 0, SE_PLAIN = 1, SE_MASKB = 2,   // (= EPI_* of wino_epi.h) bias + LeakyReLU -> y | sign-byte LeakyReLU' factors -> y
       SE_PLAIN_SIGNS = 3,                            // bias + LeakyReLU -> y and its sign bytes (D forward, c1)
       SE_POOLB = 4,                                  // bias + LeakyReLU -> sign bytes; 2x2 mean (x a + b x other) -> ypool (D forward, c2)
       SE_MASKB_POOL = 5,                             // sign-byte factors, 2x2 mean (x a + b x other) -> ypool only (tangent pass, c2)
       SE_UNPOOL = 6 };                               // x 0.25 up_mul x sign-byte factors of the finer activation -> four copies (backward-data, c1)

#ifndef PG_WS_ABL       // ablation builds (tools/exp/build_ws_abl.sh): 1 no MFMAs, 2 no output transform / epilogue arithmetic / stores, 4 no DMA after the prologue, 8 no stores
#define PG_WS_ABL 0
#endif
#ifdef PG_WINO_TRACE   // tools/exp/wino_strip_trace.py: [workgroup < 1024][wave][step < 16][8] s_memtime stamps of lane 0 (step 15: entry, exit)
#define WS_STAMP(st, i) do { __builtin_amdgcn_sched_barrier(0); if (p.trace && lane == 0 && blockIdx.x < 1024 && (st) < 15) \
    p.trace[((size_t)(blockIdx.x * 4 + wave) * 16 + (st)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WS_STAMP(st, i) do { } while (0)
#endif

template <int EPI> struct NStores { static constexpr int value = EPI == SE_PLAIN || EPI == SE_MASKB ? 4 : EPI == SE_PLAIN_SIGNS ? 8 : EPI == SE_POOLB ? 5
                                                                 : EPI == SE_MASKB_POOL ? 1 : EPI == SE_UNPOOL ? 16 : 0; };

// Lane-local output transform Y = A^T M A as wino_output_transform (wino_epi.h: same sums, same order), written on 2-vectors so
// that hipcc emits v_pk_add_f32 (48 instead of 96 VALU instructions per cout block and step; the steady state of the 8-channel
// launches is VALU-issue bound: tools/exp/wino_strip_trace.py)
__device__ __forceinline__ void output_transform_pk(const f32x4 (&acc)[16], f32x4 (&yq)[4])
{
    v2 s[2][4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v2 a0 = {acc[0 + j][2 * h], acc[0 + j][2 * h + 1]}, a1 = {acc[4 + j][2 * h], acc[4 + j][2 * h + 1]};
            const v2 a2 = {acc[8 + j][2 * h], acc[8 + j][2 * h + 1]}, a3 = {acc[12 + j][2 * h], acc[12 + j][2 * h + 1]};
            s[0][j][h] = a0 + a1 + a2;
            s[1][j][h] = a1 - a2 - a3;
        }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const v2 e = s[a][0][h] + s[a][1][h] + s[a][2][h], o = s[a][1][h] - s[a][2][h] - s[a][3][h];
            yq[2 * a + 0][2 * h] = e[0]; yq[2 * a + 0][2 * h + 1] = e[1];
            yq[2 * a + 1][2 * h] = o[0]; yq[2 * a + 1][2 * h + 1] = o[1];
        }
}

// CIN: input channels (8 / 16 / 32), NCB: 16-cout blocks per workgroup, EPI: epilogue.
// Work split: the steps of all strips form one sequence [image][strip][row step]; workgroup b takes steps [b spw, (b + 1) spw) of it
// (for each of its cout groups), i.e. a run inside one strip or the tail of one strip and the head of the next ("segments": the ring
// is primed again at a strip boundary).  The host sizes spw so that the grid is ONE round of resident workgroups: with the 36 864 /
// 2304 short-lived workgroups of the first versions a third of the time went into prologues and the unbalanced last round.
template <int CIN, int NCB, int EPI>
__global__ __launch_bounds__(256, 2) void conv_wino_strip_kernel(WinoP p, WinoStripGeo g)
{
    using R = Ring<CIN>;
    constexpr int C4 = R::C4, NCH = CIN / 8;
    constexpr int CW = 16 * NCB;                              // couts of a workgroup
    constexpr int UBASE = R::BYTES;                           // LDS: [ring][U: plane q][xi][cout] slots
    constexpr int UPLANE = 16 * CW * 16;                      // bytes of one U plane
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kk = lane >> 4;

#ifdef PG_WINO_TRACE
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 16 + 15) * 8 + 0] = __builtin_amdgcn_s_memtime();
#endif
    int b = (int)pg_xcd_remap(blockIdx.x, gridDim.x);         // [run of steps][cout group], cout groups adjacent (shared input)
    const int cog = b & (g.ncog - 1); b >>= g.lgCog;
    const int co0 = cog * CW;
    int first = __builtin_amdgcn_readfirstlane(b * g.spw);    // first step of the run, in [image][strip][row step] order
    const int last = min(first + g.spw, g.total);
    const int xH = p.ups ? (p.H >> 1) : p.H, xW = p.ups ? (p.W >> 1) : p.W;
    const unsigned rowbytes = 4u * (unsigned)(xW * CIN);
    const unsigned advance = (p.ups ? 2u : 4u) * rowbytes;
    const size_t ximg = (size_t)xH * xW * CIN;
    auto rsrc_words = [](const void* base, unsigned bytes) {
        const unsigned long long a = (unsigned long long)base;
        return pg_u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                        (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)), bytes, 0x00020000u};
    };
    const pg_u32x4 rus = rsrc_words(p.u, (unsigned)((size_t)16 * p.Cout * CIN * 4));
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned wdst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)wave * 1024u);
    // LDS-DMA through inline asm (hipcc would order every later ds_read behind a DMA it can see); M0 = the wave's 1 KiB destination
    auto dma16 = [&](const pg_u32x4& rs, unsigned vo, unsigned dst) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vo), "s"(rs), "s"(dst) : "memory");
    };

    // ---- the Winograd-domain weights of this workgroup's couts -> LDS, once: slot (q, xi, cr) <- pack q / 2, position xi, cout
    // co0 + cr, half q & 1 of U[Cin/8][16][Cout][8] (wino_u_index)
#pragma unroll
    for (int i = 0; i < C4 * NCB; ++i) {
        const int su = (i * 4 + wave) * 64 + lane;
        const int q = su / (16 * CW), r = su - q * (16 * CW);
        const int xi = r / CW, co = co0 + r - xi * CW;
        dma16(rus, 4u * (unsigned)((((q >> 1) * 16 + xi) * p.Cout + co) * 8 + 4 * (q & 1)), wdst + (unsigned)UBASE + (unsigned)i * 4096u);
    }

    // The waves that share a SIMD start in lock step (one round of workgroups, all launched at once): their MFMA phases would
    // collide and their VALU phases too -- the trace shows 9.0 k cycles per step while the phases coincide against 5.2 k once they
    // have drifted apart.  Start them a quarter of a step apart (wave slot of the SIMD from HW_ID).
    if (g.stagger > 0) {
        const int slot = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 4) & 3;          // hwreg(HW_REG_HW_ID, 0, 4) = WAVE_ID
        for (int i = 0; i < slot * g.stagger; ++i) __builtin_amdgcn_s_sleep(8);      // 8 x 64 cycles
    }

    // ---- this lane's tile of a step: tile column 8 wave + (li & 7), tile row li >> 3; patch row a = window row 2 tty + a
    const int ttx = 8 * wave + (li & 7), tty = li >> 3;
    int lb[4];                                                // byte offset of patch element (a, 0) inside its row pair (chunk 0)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int j = tty + (a >> 1);                         // row pair of the window (its parity = the shift)
        lb[a] = (((kk >> 1) * 2 + (a & 1)) * RPW + 2 * ttx + (j & 1)) * 16 + (kk & 1) * 8;
    }
    const lds_cptr ub = (lds_cptr)lds + UBASE + ((kk >> 1) * 16 * CW + li) * 16 + (kk & 1) * 8;
    const int cb0 = co0 + 4 * kk;
    const unsigned npix = (unsigned)(p.H * p.W), cout = (unsigned)p.Cout;
    const unsigned pstr = cout * 4u, rstr = (unsigned)p.W * cout * 4u;        // bytes between pixels / rows of y
    float4 bv[NCB];
#pragma unroll
    for (int c = 0; c < NCB; ++c) {
        bv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (EPI == SE_PLAIN || EPI == SE_PLAIN_SIGNS || EPI == SE_POOLB)
            if (p.bias) bv[c] = *reinterpret_cast<const float4*>(p.bias + cb0 + 16 * c);
    }
    const bool has_other = (EPI == SE_POOLB || EPI == SE_MASKB_POOL) && p.pool_other != nullptr;
    auto st4 = [](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so, float4 v) {
        // (128-bit stores keep their whole offset in the VGPR: with a REGISTER soffset hipcc's hazard recognizer drops the wait state
        //  between the store and the next VALU write of its data registers -- it assumes the hardware needs none then -- and on gfx950
        //  that corrupted dword 1 of lanes 12-15 of every 16-lane row whenever a v_pk_fma_f32 reused the registers at once: found by the
        //  bit-for-bit comparison with the tile kernel; with soffset = 0 the compiler inserts the s_nop)
        if constexpr (PG_WS_ABL & 8) asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(vo + so));
        else
        __builtin_amdgcn_raw_buffer_store_b128(pg_u32x4{__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)}, r, (int)(vo + so), 0, 0);
    };
    // sign byte of four outputs: bit j = (o_j > 0).  As integers the positive floats are exactly the positive ints (+0 = 0, -0 and the
    // negatives < 0), so the bit is med3(int, 0, 1): one instruction instead of compare + select
    auto sbyte = [](float4 o) {
        auto bit = [](float x) { unsigned r; asm("v_med3_i32 %0, %1, 0, 1" : "=v"(r) : "v"(x)); return r; };
        return (unsigned char)(bit(o.x) | (bit(o.y) << 1) | (bit(o.z) << 2) | (bit(o.w) << 3));
    };
    // LeakyReLU for a slope in [0, 1] (the host checks): o > 0 ? o : o * slope == max(o, o * slope), no VCC round trip
    auto lrelu = [&](float x) { return __builtin_fmaxf(x, x * p.slope); };

    f32x4 acc[NCB][16];
    bool again = false;
    while (first < last) {
        // ---- segment: steps [first, first + niter) of one strip
        const int rs = first & (g.stepsH - 1);
        int t = first >> g.lgStepsH;
        const int strip = t & (g.strips - 1);
        const int n = __builtin_amdgcn_readfirstlane(t >> g.lgStrips);
        const int niter = min(last - first, g.stepsH - rs);
        const int r0 = 4 * rs, ow0 = strip * SW, ox0 = ow0 + 2 * ttx;
        first += niter;

        // DMA descriptors of the input rows.  Instruction i of wave w fills slots [(4 i + w) 64, +64) of a row pair; slot -> (plane q,
        // row ri of the pair, position ps); odd pairs are shifted by one slot (pixel px = ps - 1).  voff = byte offset of the lane's
        // source inside the image for the FIRST pair of that parity (pair 0: rows r0 - 1, r0; pair 1: r0 + 1, r0 + 2); it advances by
        // four rows per use.  A row above the image makes the offset "negative" (= huge, beyond the records), a row below it exceeds
        // the records: the hardware returns zeros.  Border columns and padding slots carry PG_OOB and never advance.
        unsigned voff[2][R::NI], vstep[2][R::NI];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < R::NI; ++i) {
                const int sl = (i * 4 + wave) * 64 + lane;
                const int q = sl / (2 * RPW), rem = sl - q * (2 * RPW);
                const int ri = rem / RPW, px = rem - ri * RPW - e;
                const int col = ow0 - 1 + px, row = r0 - 1 + 2 * e + ri;
                const bool ok = sl < R::USED && (unsigned)px < (unsigned)(SW + 2) && (unsigned)col < (unsigned)p.W;
                voff[e][i] = ok ? (unsigned)(p.ups ? (row >> 1) : row) * rowbytes + 4u * (unsigned)((p.ups ? (col >> 1) : col) * CIN + 4 * q) : PG_OOB;
                vstep[e][i] = ok ? advance : 0u;
            }
        const pg_u32x4 rxs = rsrc_words(p.x + (size_t)n * ximg, (unsigned)(ximg * 4));
        auto issue_pair = [&](auto par_, int ring_pair) {         // the next row pair of parity PAR -> ring position ring_pair
            constexpr int PAR = decltype(par_)::value;
            const unsigned dst = wdst + (unsigned)ring_pair * (unsigned)(R::PITCH * 16);
#pragma unroll
            for (int i = 0; i < R::NI; ++i) {
                if (i * 4 + wave < R::NWI) dma16(rxs, voff[PAR][i], dst + (unsigned)i * 4096u);
                voff[PAR][i] += vstep[PAR][i];
            }
        };
        if (again) __builtin_amdgcn_s_barrier();              // (nobody still reads the ring of the previous segment)
        again = true;
        issue_pair(std::integral_constant<int, 0>{}, 0);
        issue_pair(std::integral_constant<int, 1>{}, 1);
        issue_pair(std::integral_constant<int, 0>{}, 2);

        // epilogue operands (specialised forms): per-image raw buffers; byte offsets of the lane's first pixel / cout, advancing per step
        __amdgpu_buffer_rsrc_t ry = pg_make_rsrc(p.y, 0), rb = ry, rpool = ry, roth = ry;
        unsigned yo = (unsigned)(((r0 + 2 * tty) * p.W + ox0) * p.Cout + cb0) * 4u;                           // y: fp32, [H][W][Cout]
        unsigned po = (unsigned)((((r0 >> 1) + tty) * (p.W >> 1) + (ox0 >> 1)) * p.Cout + cb0) * 4u;           // pooled: [H/2][W/2][Cout]
        unsigned uo = (unsigned)(((2 * (r0 + 2 * tty)) * 2 * p.W + 2 * ox0) * p.Cout + cb0) * 4u;              // pool adjoint: [2H][2W][Cout]
        if constexpr (EPI != SE_GENERIC) {
            if constexpr (EPI == SE_PLAIN || EPI == SE_PLAIN_SIGNS || EPI == SE_MASKB)
                ry = pg_make_rsrc(p.y + (size_t)n * npix * cout, npix * cout * 4u);
            if constexpr (EPI == SE_PLAIN_SIGNS) rb = pg_make_rsrc(p.ysigns + (size_t)n * npix * (cout >> 2), npix * (cout >> 2));
            if constexpr (EPI == SE_POOLB) rb = pg_make_rsrc(reinterpret_cast<unsigned char*>(p.y) + (size_t)n * npix * (cout >> 2), npix * (cout >> 2));
            if constexpr (EPI == SE_MASKB || EPI == SE_MASKB_POOL)
                rb = pg_make_rsrc(reinterpret_cast<const unsigned char*>(p.mask) + (size_t)n * npix * (cout >> 2), npix * (cout >> 2));
            if constexpr (EPI == SE_POOLB || EPI == SE_MASKB_POOL) {
                rpool = pg_make_rsrc(p.ypool + (size_t)n * (npix >> 2) * cout, (npix >> 2) * cout * 4u);
                if (p.pool_other) roth = pg_make_rsrc(p.pool_other + (size_t)n * (npix >> 2) * cout, (npix >> 2) * cout * 4u);
            }
            if constexpr (EPI == SE_UNPOOL) {
                ry = pg_make_rsrc(p.yup + (size_t)n * npix * 4 * cout, npix * cout * 16u);
                rb = pg_make_rsrc(reinterpret_cast<const unsigned char*>(p.upmask) + (size_t)n * npix * cout, npix * cout);
            }
        }
        constexpr int NMB = EPI == SE_MASKB || EPI == SE_MASKB_POOL ? 4 : EPI == SE_UNPOOL ? 16 : 1;
        unsigned mbytes[NCB][NMB];                            // sign bytes of the step (fetched before its MFMAs)
        float4 oth[NCB];
        auto prefetch_epilogue = [&]() {
#pragma unroll
            for (int c = 0; c < NCB; ++c) {
                if constexpr (EPI == SE_MASKB || EPI == SE_MASKB_POOL) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        mbytes[c][q] = __builtin_amdgcn_raw_buffer_load_b8(rb, (int)((yo >> 4) + 4 * c), (int)((((q >> 1) * rstr) + (q & 1) * pstr) >> 4), 0);
                }
                if constexpr (EPI == SE_UNPOOL) {
#pragma unroll
                    for (int e = 0; e < 16; ++e)              // e = 4 (row of the 4x4 fine patch) + column
                        mbytes[c][e] = __builtin_amdgcn_raw_buffer_load_b8(rb, (int)((uo >> 4) + 4 * c), (int)(((e >> 2) * 2 * rstr + (e & 3) * pstr) >> 4), 0);
                }
                if constexpr (EPI == SE_POOLB || EPI == SE_MASKB_POOL)
                    if (has_other) oth[c] = pg_buf_load4(roth, po + 64 * c, 0);
            }
        };
        auto epilogue_fast = [&](const f32x4 (&a16)[16], int c) {
            f32x4 yq[4];
            output_transform_pk(a16, yq);
            float4 ov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 v = yq[q];
                const unsigned so = (unsigned)(q >> 1) * rstr + (unsigned)(q & 1) * pstr;
                float4 o;
                if constexpr (EPI == SE_MASKB || EPI == SE_MASKB_POOL) {
                    const float4 f = sign_factors((unsigned char)mbytes[c][q], p.mask_slope);
                    o = make_float4(v[0] * p.scale, v[1] * p.scale, v[2] * p.scale, v[3] * p.scale);
                    o.x *= f.x; o.y *= f.y; o.z *= f.z; o.w *= f.w;
                } else {
                    o = make_float4(fmaf(v[0], p.scale, bv[c].x), fmaf(v[1], p.scale, bv[c].y), fmaf(v[2], p.scale, bv[c].z), fmaf(v[3], p.scale, bv[c].w));
                    if constexpr (EPI != SE_UNPOOL) o = make_float4(lrelu(o.x), lrelu(o.y), lrelu(o.z), lrelu(o.w));   // (pool adjoint: slope 1, no bias)
                }
                ov[q] = o;
                if constexpr (EPI == SE_PLAIN || EPI == SE_PLAIN_SIGNS || EPI == SE_MASKB) st4(ry, yo + 64 * c, so, o);
                if constexpr (EPI == SE_PLAIN_SIGNS || EPI == SE_POOLB) {
                    if constexpr (PG_WS_ABL & 8) asm volatile("" :: "v"((unsigned)sbyte(o)));
                    else __builtin_amdgcn_raw_buffer_store_b8(sbyte(o), rb, (int)((yo >> 4) + 4 * c), (int)(so >> 4), 0);
                }
                if constexpr (EPI == SE_UNPOOL) {
                    const float k = p.up_mul * 0.25f;
#pragma unroll
                    for (int dd = 0; dd < 4; ++dd) {
                        const int e = (2 * (q >> 1) + (dd >> 1)) * 4 + 2 * (q & 1) + (dd & 1);
                        const float4 f = sign_factors((unsigned char)mbytes[c][e], p.mask_slope);
                        float4 w4 = make_float4(o.x * k, o.y * k, o.z * k, o.w * k);
                        w4.x *= f.x; w4.y *= f.y; w4.z *= f.z; w4.w *= f.w;
                        st4(ry, uo + 64 * c, (unsigned)(e >> 2) * 2 * rstr + (unsigned)(e & 3) * pstr, w4);
                    }
                }
            }
            if constexpr (EPI == SE_POOLB || EPI == SE_MASKB_POOL) {
                float4 v;
                v.x = ((ov[0].x + ov[1].x) + (ov[2].x + ov[3].x)) * 0.25f; v.y = ((ov[0].y + ov[1].y) + (ov[2].y + ov[3].y)) * 0.25f;
                v.z = ((ov[0].z + ov[1].z) + (ov[2].z + ov[3].z)) * 0.25f; v.w = ((ov[0].w + ov[1].w) + (ov[2].w + ov[3].w)) * 0.25f;
                if (has_other) {
                    const float4 q = oth[c];
                    v.x = fmaf(v.x, p.pool_a, p.pool_b * q.x); v.y = fmaf(v.y, p.pool_a, p.pool_b * q.y);
                    v.z = fmaf(v.z, p.pool_a, p.pool_b * q.z); v.w = fmaf(v.w, p.pool_a, p.pool_b * q.w);
                } else if (p.pool_a != 1.f) { v.x *= p.pool_a; v.y *= p.pool_a; v.z *= p.pool_a; v.w *= p.pool_a; }
                st4(rpool, po + 64 * c, 0, v);
            }
        };

        int base = 0;                                         // ring position of the first row pair of the step's window
        for (int it = 0; it < niter; ++it) {
            WS_STAMP(it, 0);
            // this wave's share of the window (and of U) has landed: everything but the stores of the previous step's epilogue ...
            if (EPI == SE_GENERIC || it == 0 || (PG_WS_ABL & 10)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NStores<EPI>::value * NCB) : "memory");
            WS_STAMP(it, 1);
            __builtin_amdgcn_s_barrier();                     // ... everyone's has, and nobody still reads the pairs of step it - 1
            asm volatile("" ::: "memory");                    // (a raw barrier: __syncthreads() would also drain the stores in flight)
            WS_STAMP(it, 2);
            int p1 = base + 1, p2 = base + 2, p3 = base + 3, p4 = base + 4;
            if (p1 >= NPAIR) p1 -= NPAIR;
            if (p2 >= NPAIR) p2 -= NPAIR;
            if (p3 >= NPAIR) p3 -= NPAIR;
            if (p4 >= NPAIR) p4 -= NPAIR;
            if (it + 2 <= niter && !(PG_WS_ABL & 4)) {        // rows of the next step's window that are not in this one
                issue_pair(std::integral_constant<int, 1>{}, p3);
                issue_pair(std::integral_constant<int, 0>{}, p4);
            }
            if constexpr (EPI != SE_GENERIC) prefetch_epilogue();
            __builtin_amdgcn_sched_barrier(0);
            WS_STAMP(it, 3);
            const int s0 = base * (R::PITCH * 16), s1 = p1 * (R::PITCH * 16), s2 = p2 * (R::PITCH * 16);
            const int plo = tty ? s1 : s0, phi = tty ? s2 : s1;
            lds_cptr ro[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) ro[a] = (lds_cptr)lds + lb[a] + (a < 2 ? plo : phi);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                v2 d[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int c = 0; c < 4; ++c) d[a][c] = *(lds_v2ptr)(ro[a] + ch * (4 * RPW * 16) + c * 16);   // volatile: keep ds_read_b64
#ifdef PG_WINO_TRACE
                if (ch == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WS_STAMP(it, 4); }      // (traced build: the patch reads have returned)
#endif
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
                if (ch == 0) WS_STAMP(it, 5);
                __builtin_amdgcn_sched_barrier(0);              // the transform's additions stay out of the MFMA stream (see conv_wino2_kernel)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {              // one row of Winograd positions at a time: 4 x NCB accumulators interleaved
                    v2 af[NCB][4];
#pragma unroll
                    for (int c = 0; c < NCB; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) af[c][j] = *(lds_v2ptr)(ub + ch * 2 * UPLANE + (((gq * 4 + j) * NCB + c) * 16) * 16);
#pragma unroll
                    for (int s2_ = 0; s2_ < 2; ++s2_)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int c = 0; c < NCB; ++c) {
                                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
                                if constexpr (PG_WS_ABL & 1) acc[c][4 * gq + j] = f32x4{af[c][j][0], d[gq][j][0], af[c][j][1], d[gq][j][1]};
                                else
                                acc[c][4 * gq + j] = MFMA16(af[c][j][s2_], d[gq][j][s2_], (ch == 0 && s2_ == 0) ? zero4 : acc[c][4 * gq + j]);
                            }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            WS_STAMP(it, 6);
            const int oy0 = r0 + 4 * it + 2 * tty;
            if constexpr ((PG_WS_ABL & 2) != 0) {
#pragma unroll
                for (int c = 0; c < NCB; ++c)
#pragma unroll
                    for (int k = 0; k < 16; ++k) asm volatile("" :: "v"(acc[c][k]));
            } else if constexpr (EPI != SE_GENERIC) {
#pragma unroll
                for (int c = 0; c < NCB; ++c) epilogue_fast(acc[c], c);
                yo += 4u * rstr; po += rstr; uo += 16u * rstr;
            } else if (p.pn_r) {                              // (workgroup-uniform; the host launches ncog == 1 then)
                wino_epilogue_pixelnorm<NCB>(p, acc, 4 * kk, n, oy0, ox0);
            } else if (p.pnb_y) {
                wino_epilogue_pnbwd<NCB>(p, acc, 4 * kk, n, oy0, ox0);
            } else {
#pragma unroll
                for (int c = 0; c < NCB; ++c) wino_epilogue(p, acc[c], cb0 + 16 * c, n, oy0, ox0);
            }
            base = p2;
            WS_STAMP(it, 7);
        }
    }
#ifdef PG_WINO_TRACE
    if (p.trace && lane == 0 && blockIdx.x < 1024) p.trace[((size_t)(blockIdx.x * 4 + wave) * 16 + 15) * 8 + 1] = __builtin_amdgcn_s_memtime();
#endif
}

template <int CIN, int NCB, int EPI>
int launch_ws(const WinoP& p, WinoStripGeo g, hipStream_t s, char* name, size_t name_len)
{
    static const int rounds_env = getenv("PG_WSTRIP_ROUNDS") ? atoi(getenv("PG_WSTRIP_ROUNDS")) : 1;
    static const int minspw_env = getenv("PG_WSTRIP_MINSPW") ? atoi(getenv("PG_WSTRIP_MINSPW")) : 4;
    static const int stagger_env = getenv("PG_WSTRIP_STAGGER") ? atoi(getenv("PG_WSTRIP_STAGGER")) : -1;
    const size_t smem = (size_t)Ring<CIN>::BYTES + (size_t)(CIN / 4) * 16 * 16 * NCB * 16;
    auto kern = conv_wino_strip_kernel<CIN, NCB, EPI>;
    // resident workgroups of this kernel on this device (queried once per device): the grid is sized to ONE round of them
    static int slots[16] = {0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= 16) return PG_E_UNSUP;
    if (!slots[dev]) {
        if (smem > 48 * 1024) {
            if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); e != hipSuccess)
                return (int)e;
        }
        int nb = 0, cus = 0;
        if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), 256, smem); e != hipSuccess) return (int)e;
        if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev); e != hipSuccess) return (int)e;
        if (nb < 1 || cus < 1) return PG_E_UNSUP;
        slots[dev] = nb * cus;
    }
    const int rounds = rounds_env > 0 ? rounds_env : 1;
    long long target = (long long)slots[dev] * rounds / g.ncog;
    if (target < 1) target = 1;
    int spw = (int)((g.total + target - 1) / target);
    if (spw < minspw_env) spw = minspw_env;
    g.spw = spw;
    g.nrun = (g.total + spw - 1) / spw;
    // start offset between the waves that share a SIMD, in units of 512 cycles: one wave's MFMA time per step (1024 cycles per
    // 8-channel chunk and cout block)
    g.stagger = stagger_env >= 0 ? stagger_env : 2 * (CIN / 8) * NCB;
    snprintf(name, name_len, "conv_wino_strip_kernel<%d, %d, %d>", CIN, NCB, EPI);
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.nrun * g.ncog)), dim3(256), smem, s, p, g);
    return (int)hipGetLastError();
}

template <int CIN, int NCB>
int launch_ws_epi(const WinoP& p, const WinoStripGeo& g, int epi, hipStream_t s, char* name, size_t name_len)
{
    switch (epi) {
        case SE_PLAIN: return launch_ws<CIN, NCB, SE_PLAIN>(p, g, s, name, name_len);
        case SE_MASKB: return launch_ws<CIN, NCB, SE_MASKB>(p, g, s, name, name_len);
        case SE_POOLB: return launch_ws<CIN, NCB, SE_POOLB>(p, g, s, name, name_len);
        case SE_MASKB_POOL: return launch_ws<CIN, NCB, SE_MASKB_POOL>(p, g, s, name, name_len);
        case SE_PLAIN_SIGNS: if constexpr (NCB == 1) return launch_ws<CIN, 1, SE_PLAIN_SIGNS>(p, g, s, name, name_len); break;
        case SE_UNPOOL: if constexpr (NCB == 1) return launch_ws<CIN, 1, SE_UNPOOL>(p, g, s, name, name_len); break;
        default: break;
    }
    return launch_ws<CIN, NCB, SE_GENERIC>(p, g, s, name, name_len);
}

inline int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

// Called by wino_conv (conv_wino.hip) with the epilogue fields of ``p`` filled in.  PG_E_UNSUP = "not this shape": the caller keeps
// the tile kernel.  ``mode`` bit 0: specialised epilogues allowed (off: pg_debug_set_wino_epi(0)); bit 1: take the launch even when it
// needs the general epilogue (pg_debug_set_wino(21): tests) -- by default those stay on the tile kernel, which measured 1.1-1.2x
// faster there (tools/sweeps/bench_wino_strip.py: the general epilogue costs the strip kernel half of its resident waves).
int pgw::launch_wino_strip(WinoP& p, int mode, hipStream_t s, char* name, size_t name_len)
{
    static const int ncb_env = getenv("PG_WSTRIP_WINO_NCB") ? atoi(getenv("PG_WSTRIP_WINO_NCB")) : 0;
    if (p.Cin != 8 && p.Cin != 16 && p.Cin != 32) return PG_E_UNSUP;
    if ((p.Cout & 15) || (p.W % SW) || (p.H & 15) || (p.H & (p.H - 1)) || (p.W & (p.W - 1))) return PG_E_UNSUP;
    if ((long long)p.H * p.W * 32 * 4 >= (1ll << 31)) return PG_E_UNSUP;          // 32-bit byte offsets inside an image
    // the specialised epilogue, when the launch asks for exactly one of the forms the train step uses
    int se = SE_GENERIC;
    const bool small = (long long)p.H * p.W * p.Cout * 4 * (p.yup ? 4 : 1) < (1ll << 31);
    const bool slope01 = p.slope >= 0.f && p.slope <= 1.f;        // (LeakyReLU as max(o, o * slope))
    if ((mode & 1) && small && slope01 && !p.pn_r && !p.pnb_y) {
        if (p.yup) {
            if (p.upmask && p.mask_bytes && !p.bias && p.slope == 1.f && !p.mask && !p.ypool && !p.ysigns && !p.y_bytes) se = SE_UNPOOL;
        } else if (p.ypool) {
            if (p.y_bytes && !p.mask && !p.ysigns && !p.pool_only) se = SE_POOLB;
            else if (p.mask && p.mask_bytes && p.pool_only && !p.y_bytes && !p.ysigns) se = SE_MASKB_POOL;
        } else if (!p.y_bytes) {
            if (!p.mask) se = p.ysigns ? SE_PLAIN_SIGNS : SE_PLAIN;
            else if (p.mask_bytes && !p.ysigns) se = SE_MASKB;
        }
    }
    if (se == SE_GENERIC && !(mode & 2)) return PG_E_UNSUP;
    // couts per workgroup: every cout of a pixel for the PixelNorm epilogues (<= 32), else 32 when the layer has them and the
    // 64 KB of U still leave room (Cin <= 16), 16 otherwise
    int ncb = (p.pn_r || p.pnb_y) ? (p.Cout > 16 ? 2 : 1) : ((p.Cout % 32 == 0 && p.Cin <= 16) ? 2 : 1);
    if (ncb_env == 1 || ncb_env == 2) { if (!(p.pn_r || p.pnb_y)) ncb = ncb_env; }
    if (se == SE_PLAIN_SIGNS || se == SE_UNPOOL) ncb = 1;       // (these two exist for 16 couts per workgroup only: never fall through to the general epilogue)
    if ((p.pn_r || p.pnb_y) && p.Cout > 32) return PG_E_UNSUP;
    if (p.Cout % (16 * ncb)) return PG_E_UNSUP;
    WinoStripGeo g;
    g.ncog = p.Cout / (16 * ncb);
    if (g.ncog & (g.ncog - 1)) return PG_E_UNSUP;
    g.strips = p.W / SW; g.stepsH = p.H >> 2;
    g.total = p.N * g.strips * g.stepsH;
    g.lgCog = ilog2i(g.ncog); g.lgStrips = ilog2i(g.strips); g.lgStepsH = ilog2i(g.stepsH);
    g.spw = g.nrun = g.stagger = 0;                          // (launch_ws: they depend on the kernel variant's occupancy)
    switch (p.Cin) {
        case 8: return ncb == 2 ? launch_ws_epi<8, 2>(p, g, se, s, name, name_len) : launch_ws_epi<8, 1>(p, g, se, s, name, name_len);
        case 16: return ncb == 2 ? launch_ws_epi<16, 2>(p, g, se, s, name, name_len) : launch_ws_epi<16, 1>(p, g, se, s, name, name_len);
        default: return ncb == 2 ? PG_E_UNSUP : launch_ws_epi<32, 1>(p, g, se, s, name, name_len);
    }
}
