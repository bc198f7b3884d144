/* libpggan_hip.so — C-ABI of the MI355X (gfx950) PGGAN hot-path kernels.
 *
 * The reference (deepsound-project/pggan-pytorch) has no FFI of its own: its hot path is the
 * sequence of ATen op call sites issued by network.py / wgan_gp_loss.py.  Each
 * entry point below replaces one such call site (or a fused group of them); the reference
 * file:line it stands in for is cited on every declaration.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 unless stated otherwise; the caller owns all buffers (no allocation, no
 *     ownership transfer), including the scratch of the launches that slice a reduction across workgroups: its size is queried
 *     with pg_workspace_bytes and it is handed over per (device, stream) with pg_set_workspace;
 *   - process-wide state, all of it: (1) the immutable table of RCCL entry points resolved on first use of the pg_comm_* /
 *     pg_allreduce_* calls; (2) the mutex-guarded registry of caller-owned scratch buffers that pg_set_workspace fills -- the one
 *     piece of MUTABLE global state: a launch on a stream reads its entry, nothing else ever writes it.  Everything else is
 *     re-entrant and callable from any thread (the thread-local tuning / attribution aids used by bench.py and tools/ are
 *     declared separately in pggan_hip_debug.h);
 *   - "feature" tensors are NHWC  [N][H][W][C]  with C % 4 == 0 and 16-byte aligned bases;
 *   - "image"   tensors are NCHW  [N][C][H][W]  (the reference's layout at the G-output/D-input);
 *   - conv weights are packed  [KH][KW][Cout][Cin]  (the K dimension contiguous);
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*), never
 *     synchronise, and are safe to capture into a hipGraph;
 *   - return value: 0 = ok, <0 = argument error (PG_E_*), >0 = hipError_t of the launch.
 */
#ifndef PGGAN_HIP_H
#define PGGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PG_E_ARG     (-1)   /* bad dimension / null pointer            */
#define PG_E_ALIGN   (-2)   /* channel count not a multiple of 4, ...  */
#define PG_E_UNSUP   (-3)   /* unsupported kernel size / configuration */
#define PG_E_NOLIB   (-4)   /* gradient exchange: no RCCL library could be loaded in this process */
#define PG_E_RCCL_BASE (-16) /* gradient exchange: RCCL returned ncclResult_t r  =>  return value -16 - r */

typedef void* pg_stream_t;

/* Library / device info.  Returns the ABI version (bumped on any signature change). */
int pg_abi_version(void);

/* ---------------------------------------------------------------------------------------------
 * Equalized-lr convolution, implicit GEMM on v_mfma_f32_16x16x4_f32.
 * Replaces: `h = x * self.c; h = self.conv(h); h = self.act(h)`      network.py:33-36
 *           (and, through autograd, its backward-data pass and the "masked linear map" of the
 *            gradient-penalty tangent pass, wgan_gp_loss.py:25-28 + trainer.py:98).
 *
 *   z[n,oh,ow,co] = scale * sum_{kh,kw,ci} xin[n, oh-pad+kh, ow-pad+kw, ci] * w[kh][kw][co][ci]
 *   xin = x, or (ups != 0) the nearest-neighbour x2 upsampling of x   (network.py:127,129)
 *   y = mask ? z * (mask>0 ? 1 : mask_slope)                          (LeakyReLU' re-applied)
 *            : lrelu_slope(z + bias)      (slope 1.0 == no activation; bias may be NULL)
 *   Hin,Win: dims of xin (AFTER upsampling).  Hout = Hin + 2*pad - KS + 1.  KS in {1,3,4}.
 *   x: [N][Hin/(ups?2:1)][Win/(ups?2:1)][Cin]   y,mask: [N][Hout][Wout][Cout]
 */
int pg_conv2d_nhwc(const float* x, const float* w, const float* bias, const float* mask, float* y,
                   int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                   float scale, float slope, float mask_slope, pg_stream_t stream);

/* Weight gradient of the convolution above (autograd of network.py:34):
 *   dw[kh][kw][co][ci] += scale * sum_{n,oh,ow} gz[n,oh,ow,co] * xin[n,oh-pad+kh,ow-pad+kw,ci]
 *   db[co]             += sum_{n,oh,ow} gz[n,oh,ow,co]          (db may be NULL)
 * gz is the adjoint of the pre-activation z (LeakyReLU mask already applied).  Accumulates
 * (atomic fp32 adds): the caller zeroes dw/db once per step.  */
int pg_conv2d_wgrad_nhwc(const float* x, const float* gz, float* dw, float* db,
                         int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                         float scale, pg_stream_t stream);

/* The same convolution with the 2x2 average pool that follows every DBlock (network.py:229,238) and the fade-in
 * blend (network.py:233) fused into its epilogue:
 *   y     = act(conv)                                  [N][Hout][Wout][Cout]   (left UNWRITTEN when pool_only != 0
 *                                                       and the fused path is taken; always pass a valid buffer)
 *   ypool = pool_a * avgpool2(y) + pool_b * pool_other  [N][Hout/2][Wout/2][Cout]   (pool_other may be NULL)
 * Bit-identical to pg_conv2d_nhwc followed by pg_avgpool2_fwd (same summation order); split-K and non-3x3
 * launches fall back to exactly that pair internally.                                              */
int pg_conv2d_pool_nhwc(const float* x, const float* w, const float* bias, const float* mask, float* y,
                        float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
                        int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                        float scale, float slope, float mask_slope, pg_stream_t stream);

/* Generator layer in one launch: conv -> bias -> LeakyReLU -> PixelNorm (network.py:32-41):
 *   y = act(conv) * r,  r[pixel] = rsqrt(mean_c act(conv)^2 + eps)          r: [N*Hout*Wout]
 * Fused when one wave holds every cout of a pixel (Cout <= 32); otherwise conv followed by pg_pixelnorm_fwd in place. */
int pg_conv2d_pixelnorm_nhwc(const float* x, const float* w, const float* bias, float* y, float* r,
                             int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int ups,
                             float scale, float slope, float eps, pg_stream_t stream);

/* ... and its backward counterpart: backward-data conv of a layer followed by the adjoint of the PREVIOUS layer's
 * (LeakyReLU -> PixelNorm), given that layer's saved output `ysaved` and factors `r` (r may be NULL: mask only):
 *   g = scale*conv(x,w);   y = r * (g - ysaved * mean_c(g*ysaved)) * (ysaved > 0 ? 1 : slope)
 * == pg_conv2d_nhwc + pg_pixelnorm_lrelu_bwd (which is also the fallback when the tile cannot hold a pixel's couts). */
int pg_conv2d_pnbwd_nhwc(const float* x, const float* w, const float* ysaved, const float* r, float* y,
                         int N, int Hin, int Win, int Cin, int Cout, int KS, int pad,
                         float scale, float slope, pg_stream_t stream);

/* Backward-data convolution with the ADJOINT of that pool fused into its epilogue (the avg_pool2d backward +
 * LeakyReLU' mask between two DBlocks in the backward sweep):
 *   yup[n][2h+dy][2w+dx][c] = 0.25 * up_mul * scale*conv(x,w)[n][h][w][c] * (upmask[n][2h+dy][2w+dx][c] > 0 ? 1 : mask_slope)
 * `y` ([N][Hout][Wout][Cout]) is scratch: written only when the launch cannot fuse (split-K / small-M / thin
 * kernels), in which case pg_avgpool2_bwd runs as a second pass.  Bit-identical to that pair.  upmask may be NULL. */
int pg_conv2d_unpool_nhwc(const float* x, const float* w, const float* upmask, float* y, float* yup,
                          int N, int Hin, int Win, int Cin, int Cout, int KS, int pad, int flags,
                          float scale, float up_mul, float mask_slope, pg_stream_t stream);

/* Sign-byte activations.  An activation that is only ever used for its sign -- the c2 output of a DBlock before the
 * avg_pool2d of network.py:229/238: its LeakyReLU' mask in the backward / tangent sweeps -- need not exist in fp32.
 * Flag bits passed in the `ups` / `flags` argument of the pool / unpool / Winograd entry points:
 *   PG_FLAG_UPSAMPLE    bit 0: nearest x2 upsample of x fused into the gather (the historic meaning of `ups`)
 *   PG_FLAG_MASK_BYTES  `mask` / `upmask` point to sign bytes: uint8 [N][H][W][C/4], bit j of a byte = (channel 4q+j > 0)
 *   PG_FLAG_Y_BYTES     `y` points to such a byte array and receives the signs of the activated output (needs ypool)
 *   PG_FLAG_SIGNS_OUT   forward mode (no mask): the otherwise unused `mask` argument points to a byte array that receives the
 *                       signs of y IN ADDITION to the fp32 y (activations that stay in fp32 for the weight gradients but
 *                       are re-read as LeakyReLU' masks: DBlock c1 and fromRGB outputs).  Also `pool` of pg_fromrgb_fwd.
 * Only the fused epilogues of the tile kernels know the format: PG_E_UNSUP means "redo this launch with fp32 masks"
 * (pg_signbytes_to_mask expands a byte array to a +1 / -1 fp32 mask for that case).                              */
#define PG_FLAG_UPSAMPLE   1
#define PG_FLAG_MASK_BYTES 2
#define PG_FLAG_Y_BYTES    4
#define PG_FLAG_SIGNS_OUT  8
int pg_signbytes_to_mask(const unsigned char* bytes, float* mask, int64_t nbytes, pg_stream_t stream);

/* The pool adjoint evaluated in the input gathers of its two consumers (the avg_pool2d backward between two DBlocks,
 * network.py:229/238, fused into the backward-data conv and the weight gradient of the finer block's c2):
 *   gz2[n][h][w][c] = gmul * g[n][h/2][w/2][c] * (bit c of gbytes[n][h][w] ? 1 : gslope)        (never materialised)
 *   pg_conv2d_unpooled_nhwc:        y  = scale * conv3x3(gz2, w) * lrelu'(mask)     (mask: fp32 or, PG_FLAG_MASK_BYTES, sign bytes)
 *   pg_conv2d_wgrad_unpooled_nhwc:  dw += scale * sum gz2 (x) x,   db += sum gz2
 * g: [N][Hin/2][Win/2][C] (C = Cin of the conv / Cout of the weight gradient), gbytes: [N][Hin][Win][C/4].  Implemented
 * for the 8/16-channel layers of the 512^2/1024^2 stages (block-MFMA kernels); PG_E_UNSUP otherwise.               */
int pg_conv2d_unpooled_nhwc(const float* g, const float* w, const unsigned char* gbytes, float gmul, float gslope,
                            const float* mask, float* y, int N, int Hin, int Win, int Cin, int Cout, int flags,
                            float scale, float mask_slope, pg_stream_t stream);
int pg_conv2d_wgrad_unpooled_nhwc(const float* x, const float* g, const unsigned char* gbytes, float gmul, float gslope,
                                  float* dw, float* db, int N, int Hin, int Win, int Cin, int Cout,
                                  float scale, pg_stream_t stream);

/* A DBlock's first conv with the block's fromRGB layer evaluated in its input gather (fromRGB = PGConv2d 1x1 + LeakyReLU,
 * /root/reference/network.py:145, in front of the block's c1, network.py:33-36 / :151; the call order D.forward network.py:227-228):
 *   x0[n][h][w][co] = lrelu(rgb_scale * sum_c rgb_w[co][c] * img[n][c][h][w] + rgb_b[co], rgb_slope)      (never materialised)
 *   y = lrelu(scale * conv3x3(x0, w, pad 1) + bias, slope)
 * img [N][C][H][W] fp32, rgb_w [Cmid][C], w [3][3][Cout][Cmid], y [N][H][W][Cout]; x_signs / y_signs (optional): the sign bytes
 * [N][H][W][C/4] of x0 / y (what the masked backward-data forms and pg_fromrgb_bwd_* read).  For forward passes that are not
 * followed by this conv's weight gradient (which needs x0 in fp32): the G step's pass through D.  Bit-identical to pg_fromrgb_fwd
 * followed by pg_conv2d_nhwc.  Implemented for Cmid = Cout = 8, C <= 3, W % 64 == 0, H % 16 == 0 (the 1024^2 stage); PG_E_UNSUP otherwise. */
/* The generator's last conv with the block's toRGB layer in its epilogue (PGConv2d + PixelNorm /root/reference/network.py:33-41,
 * toRGB network.py:49, applied at network.py:138 with alpha = 1):
 *   y = pixelnorm(lrelu(scale * conv3x3(x, w, pad 1) + bias, slope)), r[pixel] = the normalisation factor        (as pg_conv2d_pixelnorm_nhwc)
 *   img[n][c][h][w] = t_scale * sum_co t_w[c][co] * y[n][h][w][co] + t_b[c]                                     (as pg_torgb_fwd)
 * t_w [C][Cout], img [N][C][H][W].  Implemented for Cin = Cout = 8, C <= 3, W % 64 == 0, H % 16 == 0; PG_E_UNSUP otherwise.   */
int pg_conv2d_pixelnorm_torgb_nhwc(const float* x, const float* w, const float* bias, float* y, float* r,
                                   const float* t_w, const float* t_b, float t_scale, float* img,
                                   int N, int C, int H, int W, int Cin, int Cout, float scale, float slope, float eps,
                                   pg_stream_t stream);

/* The entry block's backward-data conv with fromRGB's backward-data in its epilogue (the adjoint of /root/reference/network.py:228
 * `h = self.blocks[...](self.blocks[...].fromRGB(x))` down to the image, autograd in the reference: wgan_gp_loss.py:25-28, trainer.py:111):
 *   gf[n][h][w][ci] = scale * conv3x3(gz, wt, pad 1) * (bit ci of mask_bytes[n][h][w] ? 1 : mask_slope)     (wt: the flipped / transposed weights of
 *   pg_pack_dgrad_weights; written to y unless y == NULL)
 *   gimg[n][c][h][w] = rgb_scale * sum_co rgb_w[co][c] * gf[n][h][w][co]                 (gimg != NULL; as pg_fromrgb_bwd_data)
 *   rgb_dw[co][c] += rgb_scale * sum_{n,h,w} gf[n][h][w][co] * img[n][c][h][w],  rgb_db[co] += sum gf   (rgb_dw != NULL; as pg_fromrgb_wgrad:
 *   fromRGB's weight gradient of trainer.py:98, one commit of 8 * (C + 1) atomics per workgroup)
 * At least one of gimg / rgb_dw.  Implemented for Cin = Cout = 8, C <= 3, W % 64 == 0, H % 16 == 0 (the 1024^2 stage); PG_E_UNSUP otherwise. */
int pg_conv2d_masked_fromrgb_bwd_nhwc(const float* gz, const float* wt, const unsigned char* mask_bytes, float mask_slope, float* y,
                                      const float* rgb_w, float rgb_scale, float* gimg,
                                      const float* img, float* rgb_dw, float* rgb_db,
                                      int N, int C, int H, int W, int Cin, int Cout, float scale, pg_stream_t stream);

int pg_conv2d_fromrgb_nhwc(const float* img, const float* rgb_w, const float* rgb_b, float rgb_scale, float rgb_slope,
                           unsigned char* x_signs, const float* w, const float* bias, float* y, unsigned char* y_signs,
                           int N, int C, int H, int W, int Cmid, int Cout, float scale, float slope, pg_stream_t stream);

/* Winograd F(2x2,3x3) path for the 3x3 layers (pad 1; Cin % 8 == 0; H, W powers of two >= 8): 2.25x fewer MFMAs
 * than the direct implicit GEMM, same fp32 sums re-associated (transform coefficients +-1, 1/2; ~1e-6 relative).
 *   pg_wino_transform_weights: u = G g G^T of w[3][3][Cout][Cin] (once per weight version), 16*Cout*Cin floats stored in
 *   8-channel packs u[Cin/8][16][Cout][8] (the slice a workgroup stages per K chunk is then whole 128-byte lines); Cin % 8 == 0
 *   pg_conv2d_wino_nhwc: the conv of pg_conv2d_nhwc (KS 3, pad 1) on the transformed weights, with the optional fused
 *   epilogues of pg_conv2d_pool_nhwc (ypool/pool_other/pool_a/pool_b/pool_only) and pg_conv2d_unpool_nhwc
 *   (yup/upmask/up_mul); pass NULL for the ones not wanted.  Returns PG_E_UNSUP for shapes it does not take.      */
int pg_wino_transform_weights(const float* w, float* u, int Cout, int Cin, pg_stream_t stream);
/* ... for nlayers layers in one launch (layer i: weights at wbase + woff[i], result at ubase + uoff[i]).  transposed (may be NULL):
 * non-zero = layer i is the BACKWARD-DATA form of a forward layer taken straight from that layer's parameter: cout[i] / cin[i] are the
 * channel counts of the backward-data conv (= the forward layer's Cin / Cout), the source is the forward w[3][3][cin[i]][cout[i]], and
 * the result equals pg_wino_transform_weights of pg_pack_dgrad_weights of it -- without the intermediate copy.                     */
int pg_wino_transform_weights_batched(const float* wbase, float* ubase, int nlayers, const int64_t* woff,
                                      const int64_t* uoff, const int* cout, const int* cin, const int* transposed,
                                      pg_stream_t stream);
int pg_conv2d_wino_nhwc(const float* x, const float* u, const float* bias, const float* mask, float* y,
                        float* ypool, const float* pool_other, float pool_a, float pool_b, int pool_only,
                        float* yup, const float* upmask, float up_mul,
                        int N, int H, int W, int Cin, int Cout, int ups,
                        float scale, float slope, float mask_slope, pg_stream_t stream);
/* pg_conv2d_pixelnorm_nhwc on the transformed weights (PGConv2d with pixelnorm=True, network.py:32-52, KS 3, pad 1): conv ->
 * bias -> LeakyReLU -> PixelNorm in the Winograd epilogue, r[N*H*W] = rsqrt(mean_c y^2 + eps) kept for the backward pass.
 * A workgroup must hold every cout of its pixels: Cout <= 32 (PG_E_UNSUP otherwise; wider layers normalise in a second pass). */
int pg_conv2d_wino_pixelnorm_nhwc(const float* x, const float* u, const float* bias, float* y, float* r,
                                  int N, int H, int W, int Cin, int Cout, int ups,
                                  float scale, float slope, float eps, pg_stream_t stream);

/* Backward-data conv of a generator layer on the transformed weights + the adjoint of the previous layer's (LeakyReLU -> PixelNorm)
 * (network.py:44-52) in the Winograd epilogue: y = r * (g - ysaved * mean_c(g * ysaved)) * lrelu'(ysaved) with g = scale * conv(x),
 * or, pool != 0, g = pool_a * avgpool2(scale * conv(x)) + pool_b * pool_other (the adjoint of the nearest x2 upsample is 4 * avgpool2:
 * network.py:62-66); y, ysaved [N][H(/2)][W(/2)][Cout], r [N*H*W(/4)].  Cout <= 32 (PG_E_UNSUP otherwise: second pass).          */
int pg_conv2d_wino_pnbwd_nhwc(const float* x, const float* u, const float* ysaved, const float* r, float* y,
                              int pool, const float* pool_other, float pool_a, float pool_b,
                              int N, int H, int W, int Cin, int Cout, float scale, float slope, pg_stream_t stream);

/* Scratch for the launches on `stream` (of the current device) that slice their K loop across workgroups: the 3x3 layers of
 * the 16x16 / 32x32 stages at minibatch 3 give pg_conv2d_wino_nhwc fewer workgroups than the chip has CUs, so up to 8 workgroups
 * share a (64-tile, 16-cout) block, each leaves its partial outputs in the scratch and the last one to arrive adds them in slice
 * order (deterministic) and runs the fused epilogue.  The library never allocates device memory: the caller owns `ptr` (16-byte
 * aligned, ZERO-FILLED once, > 16 KB; 32 MB covers every layer of the 1024x1024 schedule), keeps it alive until it registers
 * another one or clears the entry (ptr NULL, bytes 0), and uses it for nothing else.  Without a registered scratch, or when a layer
 * would need more than `bytes`, launches run unsplit -- same results to fp32 summation order.  Thread-safe.  The scratch carries
 * self-resetting tickets: it belongs to ONE stream (launches on a stream are ordered); two streams, or two concurrently running
 * branches of one captured hipGraph, must not share one.                                                                          */
int pg_set_workspace(pg_stream_t stream, void* ptr, size_t bytes);
/* Bytes of scratch the launch of that shape uses when at least as much is registered for its stream (0: it never slices).
 * kind 0: pg_conv2d_wino_nhwc(N, H, W, Cin, Cout); kind 1: pg_conv2d_nhwc with the 4x4 valid kernel on a 4x4 map (H = W = 4).
 * The maximum over the layers a caller launches is the size to register (1024x1024 schedule at minibatch 3: < 32 MB).            */
int pg_workspace_bytes(int kind, int N, int H, int W, int Cin, int Cout, size_t* bytes);

/* Winograd weight gradient of the same layers: dW[kh][kw][co][ci] += scale * sum gz*x (3x3, pad 1), db[co] += sum gz, computed as
 * G^T [ sum_tiles (A dY A^T) (.) (B^T d B) ] G  -- 16 MFMAs per 4 output tiles instead of 36.  H, W powers of two with
 * H >= 8, W >= 16; commits with fp32 atomics.  Same arguments as pg_conv2d_wgrad_nhwc (KS 3, pad 1 implied).          */
int pg_conv2d_wgrad_wino_nhwc(const float* x, const float* gz, float* dw, float* db,
                              int N, int H, int W, int Cin, int Cout, int ups, float scale, pg_stream_t stream);
/* Two batches of the same layer in one launch (the regions of both walk through the same workgroups, dW is committed once):
 * dW += scale * (wgrad(x, gz) + wgrad(x2, gz2)); db += sum gz of the batches named in db_batches (bit 0: first, bit 1: second).
 * N2 = 0: the single-batch form.  Used for the gradient-penalty tangent term + the batched adjoint sweep of D (wgan_gp_loss.py:36-55). */
int pg_conv2d_wgrad_wino2_nhwc(const float* x, const float* gz, int N, const float* x2, const float* gz2, int N2,
                               float* dw, float* db, int db_batches,
                               int H, int W, int Cin, int Cout, int ups, float scale, pg_stream_t stream);

/* Repack forward weights [KS][KS][Cout][Cin] into the weights of the backward-data convolution
 * [KS][KS][Cin][Cout] (spatially flipped, channels transposed).                               */
int pg_pack_dgrad_weights(const float* w, float* wt, int KS, int Cout, int Cin, pg_stream_t stream);

/* The same for `nlayers` layers of one network in a single launch: layer l lives at element offset off[l] of the flat
 * weight buffer `wbase` and of its mirror `wtbase` (host arrays off/ks/cout/cin are read before the call returns). */
int pg_pack_dgrad_weights_batched(const float* wbase, float* wtbase, int nlayers, const int64_t* off,
                                  const int* ks, const int* cout, const int* cin, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fromRGB: 1x1 conv from an NCHW image (C_img in {1,3,4}) to NHWC features, fused LeakyReLU.
 * Replaces DBlock/DLastBlock.fromRGB  network.py:145,160 (+ F.avg_pool2d(x,2) network.py:231 when
 * pool != 0: the image is [N][C][2H][2W] and is 2x2-averaged on the fly).
 *   y[n,h,w,co] = epilogue(scale * sum_c img[n,c,h,w] * w[co][c])   epilogue as pg_conv2d_nhwc. */
int pg_fromrgb_fwd(const float* img, const float* w, const float* bias, const float* mask, float* y,
                   int N, int C, int H, int W, int Cout, int pool,
                   float scale, float slope, float mask_slope, pg_stream_t stream);

/* d/d img of the above: gimg[n,c,h,w] (+)= mul * scale * sum_co gz[n,h,w,co] * w[co][c]
 * (pool != 0: each of the 2x2 source pixels receives a quarter).  accumulate != 0 adds.       */
int pg_fromrgb_bwd_data(const float* gz, const float* w, float* gimg,
                        int N, int C, int H, int W, int Cout, int pool, int accumulate,
                        float scale, pg_stream_t stream);

/* dw[co][c] += scale * sum_pix gz[pix,co] * img[pix,c];  db[co] += sum_pix gz[pix,co].        */
int pg_fromrgb_wgrad(const float* gz, const float* img, float* dw, float* db,
                     int N, int C, int H, int W, int Cout, int pool, float scale, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * toRGB: 1x1 conv from NHWC features to an NCHW image, no activation, fused fade-in blend.
 * Replaces GBlock.toRGB network.py:49,65,70 and `preult_rgb*(1-alpha) + ult*alpha` network.py:138
 * (toRGB of the previous block commutes with the nearest upsample, so `prev` is low-res):
 *   out[n,c,h,w] = out_mul * (scale * sum_ci x[n,h,w,ci]*w[c][ci] + bias[c])
 *                + (prev ? prev_mul * prev[n,c,h/2,w/2] : 0)                                  */
int pg_torgb_fwd(const float* x, const float* w, const float* bias, const float* prev, float* out,
                 int N, int C, int H, int W, int Cin, float scale, float out_mul, float prev_mul,
                 pg_stream_t stream);

/* gx[n,h,w,ci] = mul * scale * sum_c g[n,c,h(*),w(*)] * w[c][ci]
 * down != 0: g is [N][C][2H][2W] and is 2x2-SUMMED on the fly (adjoint of the upsample of `prev`). */
int pg_torgb_bwd_data(const float* g, const float* w, float* gx,
                      int N, int C, int H, int W, int Cin, int down, float mul_scale, pg_stream_t stream);
/* ... followed by the adjoint of the block's (LeakyReLU -> PixelNorm), network.py:44-52, in the same launch:
 * gx = r * (gh - ysaved * mean_c(gh * ysaved)) * lrelu'(ysaved).  8 features on large maps; PG_E_UNSUP otherwise. */
int pg_torgb_bwd_data_pnbwd(const float* g, const float* w, const float* ysaved, const float* r, float* gx,
                            int N, int C, int H, int W, int Cin, float mul_scale, float slope, pg_stream_t stream);

/* dw[c][ci] += mul_scale * sum_pix g[pix,c]*x[pix,ci];  db[c] += mul * sum_pix g[pix,c].       */
int pg_torgb_wgrad(const float* g, const float* x, float* dw, float* db,
                   int N, int C, int H, int W, int Cin, int down, float mul_scale, float mul,
                   pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * 2x2 average pool on NHWC features with the D-side fade-in blend fused.
 * Replaces F.avg_pool2d network.py:229,238 and `h*alpha + (1-alpha)*preult_rgb` network.py:233.
 *   y = a * avgpool2(x) + (other ? b * other : 0)        x:[N][2H][2W][C]  y,other:[N][H][W][C] */
int pg_avgpool2_fwd(const float* x, const float* other, float* y, int N, int H, int W, int C,
                    float a, float b, pg_stream_t stream);

/* Adjoint: gx[n,h,w,c] = mul * 0.25 * gy[n,h/2,w/2,c] * (mask ? (mask[n,h,w,c]>0 ? 1 : mask_slope) : 1)
 * gx,mask: [N][2H][2W][C]   gy: [N][H][W][C]                                                   */
int pg_avgpool2_bwd(const float* gy, const float* mask, float* gx, int N, int H, int W, int C,
                    float mul, float mask_slope, pg_stream_t stream);

/* Adjoint of the nearest x2 upsample (network.py:127,129): gx[n,h,w,c] = sum_{2x2} g[n,2h+i,2w+j,c]. */
int pg_upsample2_bwd(const float* g, float* gx, int N, int H, int W, int C, pg_stream_t stream);

/* y = a*x + b*other (other may be NULL), y = (mask>0?1:mask_slope) * that.  Flat length n % 4 == 0. */
int pg_axpby_mask(const float* x, const float* other, const float* mask, float* y, int64_t n,
                  float a, float b, float mask_slope, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PixelNorm over channels.  Replaces network.py:37-40 and :120-123.
 *   r[p] = rsqrt(mean_c(x[p,c]^2) + eps);  y[p,c] = x[p,c]*r[p]     (y may alias x)            */
int pg_pixelnorm_fwd(const float* x, float* y, float* r, int64_t P, int C, float eps, pg_stream_t stream);

/* Adjoint of (LeakyReLU -> PixelNorm) given the saved OUTPUT y and r:
 *   gh = r * (gy - y * mean_c(gy*y));   gz = gh * (y>0 ? 1 : slope)          (gz may alias gy)
 * r == NULL: PixelNorm absent, only the LeakyReLU mask is applied.                             */
int pg_pixelnorm_lrelu_bwd(const float* gy, const float* y, const float* r, float* gz,
                           int64_t P, int C, float slope, pg_stream_t stream);

/* Discriminator(pixelnorm=True) (network.py:191-198 flag) under the gradient penalty: PixelNorm is not
 * piecewise linear, so the double backward of wgan_gp_loss.py:25-31 gets a Hessian-vector term per layer.
 * With P(h) = r (I - y y^T / C) (Jacobian == adjoint of y = h*r(h)), t the tangent at the PixelNorm input and
 * a the FIRST-backward adjoint at the PixelNorm output:
 *   ty  = P t;    inj = grad_h <t, P(h) a> = -(r^2 S / C) y - (r / C) P[(a.y) t + (t.y) a],  S = t.a - (t.y)(a.y)/C
 * `inj` is added to the adjoint of the PixelNorm input in the ordinary backward of the mixed samples:
 *   gz = (r * (gy - y * mean_c(gy*y)) + inj) * (y>0 ? 1 : slope)                                   */
int pg_pixelnorm_tangent(const float* t, const float* y, const float* r, const float* a, float* ty, float* inj,
                         int64_t P, int C, pg_stream_t stream);
int pg_pixelnorm_lrelu_bwd_inj(const float* gy, const float* y, const float* r, const float* inj, float* gz,
                               int64_t P, int C, float slope, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Minibatch stddev (network.py:174-187): ONE scalar per group over the whole [n,H,W,C] tensor.
 * x: [NB][HW][C]   y: [NB][HW][CP] (CP >= C+1, CP % 4 == 0; channels > C are zero-filled)
 * NB = G groups of n consecutive images.  stats: G rows of PG_MBSTD_STATS_STRIDE floats, row g = {mu, sigma,
 * workspace of the multi-workgroup reduction...} (caller-owned, no allocation inside).
 *   y[...,:C] = x;  y[...,C] = sigma_g = sqrt(mean((x-mu)^2) + 1e-8)                           */
#define PG_MBSTD_STATS_STRIDE 136
int pg_mbstd_fwd(const float* x, float* y, float* stats, int G, int n, int HW, int C, int CP,
                 pg_stream_t stream);

/* Tangent (forward-mode) of the above, used by the gradient-penalty second-order pass:
 *   ty[...,:C] = tx;  ty[...,C] = <x-mu, tx> / (M*sigma);  tstats row g = {mean(tx), <x-mu,tx>, workspace} (same stride) */
int pg_mbstd_tangent(const float* x, const float* tx, const float* stats, float* ty, float* tstats,
                     int G, int n, int HW, int C, int CP, pg_stream_t stream);

/* Adjoint: gx = (gy[...,:C] + Gs*(x-mu)/(M*sigma) + hvp) * (x>0 ? 1 : mask_slope),  Gs = sum gy[...,C]
 *   hvp (only when tx != NULL; Gs then comes from `gy_first`, the adjoint of the FIRST backward):
 *   hvp = Gs1/(M*sigma) * ((tx - mean tx) - (x-mu)*<x-mu,tx>/(M*sigma^2)),  Gs1 = sum gy_first[...,C]
 * gy may be NULL (then only the hvp term is produced).  `apply_mask`==0 skips the LeakyReLU mask. */
int pg_mbstd_bwd(const float* gy, const float* x, const float* stats,
                 const float* tx, const float* tstats, const float* gy_first,
                 float* gx, int G, int n, int HW, int C, int CP, int apply_mask, float mask_slope,
                 pg_stream_t stream);

/* Exact-global minibatch stddev under data parallelism (the optional mode; reference network.py:174-187 evaluated on the
 * GLOBAL batch world x mb): every rank holds an equal shard of each group.  The two launches of pg_mbstd_fwd / pg_mbstd_tangent are
 * separate entry points, so that the host exchanges the partial rows between them (a sum all-reduce of a zero-filled
 * [nranks][G][PG_MBSTD_STATS_STRIDE] buffer in which every rank fills its own slice = an all-gather):
 *   pg_mbstd_stats / pg_mbstd_tangent_stats   this shard's partials -> stats / tstats rows
 *   pg_mbstd_write / pg_mbstd_tangent_write   merge the rows of all ranks (`gathered`, rank-major; NULL with nranks 1 = the local row), in rank
 *                                             order on every rank (bit-identical mu / sigma everywhere), M = all shards; y / ty as above
 *   pg_mbstd_gsum                             out[2g] = sum gy[..., C], out[2g+1] = sum gy_first[..., C] of this shard (either may be NULL)
 *   pg_mbstd_bwd_global                       pg_mbstd_bwd with Gs / Gs1 taken from `gsums` (pg_mbstd_gsum summed over all ranks) and M = all shards */
int pg_mbstd_stats(const float* x, float* stats, int G, int n, int HW, int C, pg_stream_t stream);
int pg_mbstd_write(const float* x, float* y, float* stats, const float* gathered, int nranks,
                   int G, int n, int HW, int C, int CP, pg_stream_t stream);
int pg_mbstd_tangent_stats(const float* x, const float* tx, const float* stats, float* tstats,
                           int G, int n, int HW, int C, pg_stream_t stream);
int pg_mbstd_tangent_write(const float* tx, float* ty, float* tstats, const float* stats, const float* gathered, int nranks,
                           int G, int n, int HW, int C, int CP, pg_stream_t stream);
int pg_mbstd_gsum(const float* gy, const float* gy_first, float* out, int G, int n, int HW, int C, int CP, pg_stream_t stream);
int pg_mbstd_bwd_global(const float* gy, const float* x, const float* stats,
                        const float* tx, const float* tstats, const float* gy_first,
                        float* gx, const float* gsums, int nranks,
                        int G, int n, int HW, int C, int CP, int apply_mask, float mask_slope, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Final nn.Linear(nf0, 1)  network.py:219,239.   s[n] = sum_c h[n,c]*w[c] + b                  */
int pg_linear1_fwd(const float* h, const float* w, const float* b, float* s, int N, int C, pg_stream_t stream);
/* gh[n,c] = gs[n]*w[c] * (mask ? (mask[n,c]>0?1:mask_slope) : 1)                               */
int pg_linear1_bwd_data(const float* gs, const float* w, const float* mask, float* gh, int N, int C,
                        float mask_slope, pg_stream_t stream);
/* dw[c] += sum_n gs[n]*h[n,c];  db[0] += sum_n gs[n]                                          */
int pg_linear1_wgrad(const float* gs, const float* h, float* dw, float* db, int N, int C, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * WGAN-GP pieces  (wgan_gp_loss.py).
 * mixed = real*(1-m[n]) + fake*m[n]            wgan_gp_loss.py:8-10,19   (E elements per image) */
int pg_gp_mix(const float* real, const float* fake, const float* m, float* mixed, int N, int64_t E,
              pg_stream_t stream);
/* ss[n] = sum_e g[n,e]^2   (ss must be zeroed by the caller; atomic accumulation)              */
int pg_row_sumsq(const float* g, float* ss, int N, int64_t E, pg_stream_t stream);
/* gp[n] = lambda*(sqrt(ss[n])-target)^2/target^2                        wgan_gp_loss.py:31
 * u[n,e] = inv_n * 2*lambda*(norm-target)/(target^2*norm) * g[n,e]   (seed of the tangent pass) */
int pg_gp_seed(const float* g, const float* ss, float* gp, float* u, int N, int64_t E,
               float lambda, float target, float inv_n, pg_stream_t stream);
/* Loss algebra wgan_gp_loss.py:48,55,62: scores s = [real(N) | fake(N) | mixed(N)].
 *   d_real_loss[n] = -s_r + eps*s_r^2;  d_fake_loss[n] = s_f;  d_cost = mean(d_fake+d_real+gp)
 *   gscore[0:N] = (-1+2*eps*s_r)/N;  gscore[N:2N] = 1/N;  gscore[2N:3N] = 0                   */
int pg_d_loss(const float* s, const float* gp, float* d_cost, float* d_real_loss, float* d_fake_loss,
              float* gscore, int N, float eps, pg_stream_t stream);
/* g_cost = mean(-s);  gscore[n] = -1/N                                  wgan_gp_loss.py:72-73   */
int pg_g_loss(const float* s, float* g_cost, float* gscore, int N, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Adam (train.py:148-149,195; torch-2.10 form) on one flat segment, gradient pre-scaled by
 * grad_scale (1/world_size after the RCCL sum).  p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)   */
int pg_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
            float eps, float bc1, float bc2_sqrt, float grad_scale, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Steps either side of the path.
 * Real-image input: replaces DepthDataset.__getitem__ dataset.py:54-67 (alpha_fade :109-113 when alpha < 1,
 * adjust_dynamic_range utils.py:24-30, astype float32) for a whole uint8 batch on the device:
 *   t = 2x2 box mean (upsampled back);  v = x + (t - x)*(1-alpha);  out = (v - min_in)*(max_out-min_out)/(max_in-min_in) + min_out
 * evaluated in fp64 like numpy does for uint8 input, rounded once to fp32 (bit-exact).  in/out: [planes][H][W].  */
int pg_real_prepare_u8(const uint8_t* in, float* out, int64_t planes, int H, int W, double alpha,
                       double min_in, double max_in, double min_out, double max_out, pg_stream_t stream);

/* One level of the multi-depth image pyramid: replaces DefaultImageFolderDataset.create_datapoint_from_depth
 * dataset.py:243-250 for a uint8 batch: out[y][x] = uint8(clip(rint(mean of in[y*s + {0,1}][x*s + {0,1}]), min_in, max_in)),
 * s = 2^depthdiff (the reference's sampling: a 2x2 box for depthdiff 1).  in [planes][H][W] -> out [planes][H/s][W/s]. */
int pg_pyramid_level_u8(const uint8_t* in, uint8_t* out, int64_t planes, int H, int W, int depthdiff,
                        float min_in, float max_in, pg_stream_t stream);

/* Sample output: replaces ImageSaver.__call__ output_postprocess.py:35-62 up to the PIL hand-off: nearest
 * upsample by `up` (utils.py:33-53), tiled grid of ceil(sqrt(n)) columns, CHW->HWC, range (min_in,max_in)->(0,255)
 * in fp32, round-half-even, clip, uint8.  grid: [grid_h*h*up][grid_w*w*up][C].                                  */
int pg_image_grid_u8(const float* img, uint8_t* grid, int n, int C, int h, int w, int up,
                     float min_in, float max_in, pg_stream_t stream);

/* Sound input step: replaces SoundImageDataset.load_file dataset.py:285-300 after the file read, for a waveform on the device.
 *   pg_stft_abslog: y [nsamp][channels] fp32 (channels > 1: mono mix-down sum/2, :287-288) ->
 *       out[k][t] = log(1 + |STFT(y)[k][t]|), k < bins, t < frames, with librosa's stft definition (periodic Hann window,
 *       center=True / reflect padding, hop_length; :293-296).  n_fft: power of two <= 2048.  librosa is not in this image:
 *       parity unpinned (oracle/sound_steps.py restates the published algorithm).
 *   pg_minmax_f32 + pg_stretch_to_u8: np.uint8(adjust_dynamic_range(s, (s.min(), s.max()), (0, max_out))) dataset.py:299. */
int pg_stft_abslog(const float* y, int64_t nsamp, int channels, float* out, int n_fft, int hop_length,
                   int bins, int frames, pg_stream_t stream);
/*   pg_stft_image: the same with the image mode as an argument: PG_SOUND_ABSLOG (dataset.py:296, = pg_stft_abslog) or
 *       PG_SOUND_REALLOG log(1 + |Re s|) * sign(s) (dataset.py:298; sign of a complex number as in the reference's numpy 1.13:
 *       the sign of the real part, of the imaginary part where the real part is zero — the image is real).
 *   pg_mono_f32: img_mode 'raw' (dataset.py:287-291): out[i] = mono mix-down of sample i, i < count ((2^size)^2 samples,
 *       reshaped by the caller); followed by the same min/max stretch. */
#define PG_SOUND_ABSLOG 0
#define PG_SOUND_REALLOG 1
int pg_stft_image(const float* y, int64_t nsamp, int channels, float* out, int n_fft, int hop_length,
                  int bins, int frames, int mode, pg_stream_t stream);
int pg_mono_f32(const float* y, int64_t nsamp, int channels, float* out, int64_t count, pg_stream_t stream);
int pg_minmax_f32(const float* x, int64_t n, float* lohi, pg_stream_t stream);
int pg_stretch_to_u8(const float* x, uint8_t* out, int64_t n, const float* lohi, float max_out, pg_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gradient exchange of the data-parallel step: RCCL over xGMI (SURVEY.md §8b "the all-reduce itself is a C-ABI call
 * taking ncclComm_t, buffer, count, stream", §8e).  The reference is single-GPU and has no collective; the exchange
 * points are after `D_loss.backward()` trainer.py:98 (before optimizer_d.step() :100) and after `G_loss.backward()`
 * trainer.py:111 (before optimizer_g.step() :112).  One process per GPU.  Bootstrap from the host language: rank 0
 * calls pg_comm_unique_id, the PG_COMM_ID_BYTES bytes reach the other ranks out of band (torch.distributed store),
 * every rank calls pg_comm_init_rank (collective: all ranks must call it).  `comm` is an ncclComm_t.
 *   pg_allreduce_sum_f32: IN-PLACE sum over ranks of buf[0..count) (fp32, device), asynchronous on `stream`; averaging
 *   is folded into pg_adam (grad_scale = 1/world).  Calls on one communicator must be issued in the same order on
 *   every rank (RCCL semantics).  Returns PG_E_NOLIB when no RCCL can be loaded, PG_E_RCCL_BASE - r for an RCCL error. */
#define PG_COMM_ID_BYTES 128
int pg_rccl_version(int* version);
int pg_comm_unique_id(void* id_out /* host, PG_COMM_ID_BYTES */);
int pg_comm_init_rank(void** comm_out, int nranks, const void* id /* host, PG_COMM_ID_BYTES */, int rank);
int pg_comm_info(void* comm, int* nranks, int* rank);
int pg_comm_destroy(void* comm);
int pg_allreduce_sum_f32(void* comm, float* buf, int64_t count, pg_stream_t stream);

/* Utility: async fill with zero bytes.                                                         */
int pg_zero(void* p, int64_t bytes, pg_stream_t stream);
/* n draws of U[0,1): replaces torch.cuda.FloatTensor(n, 1).uniform_() wgan_gp_loss.py:15-17 (the mixing factors of the gradient
 * penalty).  Counter-based (Philox4x32-10): out[i] is a pure function of (seed, offset, i); the caller advances `offset` per draw. */
int pg_uniform_f32(float* out, int64_t n, uint64_t seed, uint64_t offset, pg_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PGGAN_HIP_H */
