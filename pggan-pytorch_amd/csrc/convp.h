// Launch parameter blocks of the conv / weight-gradient kernels, shared by conv_igemm.hip (tile kernels) and conv_strip.hip
// (row-streaming kernels of the 8/16-channel 1024^2 layers).  Host + device, plain data.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pgk {

struct ConvP {
    const float* x; const float* w; const float* bias; const float* mask; float* y;
    int N, Hin, Win, Cin, Cout, Hout, Wout, KS, pad, ups;
    float scale, slope, mask_slope;
    int lgTW, lgTH, TN, tilesW, tilesH;
    unsigned mWT, mHT;          // floor(2^32/WT)+1, floor(2^32/HT)+1: exact n/d for n < 2^16 via __umulhi
    int ksplit;                 // >1: blockIdx.z owns a slice of the Cin chunks, partial sums are
                                // committed with fp32 atomics into a pre-zeroed y (epilogue deferred)
    // fused 2x2 average pool of the activated output (pg_conv2d_pool_nhwc): ypool = pool_a * avgpool2(y) + pool_b * pool_other
    float* ypool; const float* pool_other; float pool_a, pool_b; int pool_only;
    // fused adjoint of that pool (pg_conv2d_unpool_nhwc): yup[n][2h+dy][2w+dx][c] = 0.25*up_mul * y[n][h][w][c] * lrelu'(upmask[...])
    float* yup; const float* upmask; float up_mul;
    // fused PixelNorm of the activated output (pg_conv2d_pixelnorm_nhwc): y *= rsqrt(mean_c y^2 + pn_eps), pn_r[pixel] = that factor
    float* pn_r; float pn_eps;
    // fused adjoint of (LeakyReLU -> PixelNorm) applied to the conv result g (pg_conv2d_pnbwd_nhwc):
    //   y = r[pix] * (g - pnb_y * mean_c(g * pnb_y)) * lrelu'(pnb_y)
    const float* pnb_y; const float* pnb_r;
    // sign-byte activations (PG_FLAG_MASK_BYTES / PG_FLAG_Y_BYTES): one byte per float4, bit j = (channel 4q+j > 0)
    int mask_bytes, y_bytes;
    unsigned char* ysigns;      // PG_FLAG_SIGNS_OUT: the sign bytes of y are written here IN ADDITION to y (forward mode)
    // pool adjoint fused into the input gather (pg_conv2d_unpooled_nhwc): xin[n][h][w][c] = gmul * x[n][h/2][w/2][c] * lrelu'(gbytes[n][h][w][c])
    const unsigned char* gbytes; float gmul, gslope;
};

// LeakyReLU' factors of four channels from a sign byte / the sign byte of four activated outputs
__device__ __forceinline__ float4 pg_sign_factors(unsigned char b, float slope)
{
    return make_float4((b & 1) ? 1.f : slope, (b & 2) ? 1.f : slope, (b & 4) ? 1.f : slope, (b & 8) ? 1.f : slope);
}
__device__ __forceinline__ unsigned char pg_sign_byte(float4 o)
{
    return (unsigned char)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
}

struct WgP {
    const float* x; const float* gz; float* dw; float* db;
    int N, Hin, Win, Cin, Cout, Hout, Wout, pad, ups;
    float scale;
    int lgTW, lgTH, TN, tilesW, tilesH, ntiles, tiles_per_block;
    unsigned mWT, mHT;          // magic reciprocals of the halo tile width / height (see ConvP)
    int atomic;                 // 0: this workgroup is the only writer of its dW block -> plain +=
    // pool adjoint fused into the gz gather (pg_conv2d_wgrad_unpooled_nhwc): gz[n][h][w][c] = gmul * g[n][h/2][w/2][c] * lrelu'(gbytes[n][h][w][c])
    const unsigned char* gbytes; float gmul, gslope;
#ifdef PG_WINO_TRACE
    unsigned long long* trace;  // [workgroup][wave][tile < 8][8] s_memtime stamps (tools/exp/wgrad_trace.py)
#endif
};

// Scratch registered for a stream by pg_set_workspace (conv_wino.hip): [WS_TICKETS zero-initialised, self-resetting tickets][partial sums].
// Launches on one stream are ordered, so every kernel that slices a reduction across workgroups may use the whole of it.
struct Workspace { int device; hipStream_t stream; char* ptr; size_t bytes; };
constexpr size_t WS_TICKETS = 4096, WS_HEAD = WS_TICKETS * sizeof(unsigned);
bool find_workspace(hipStream_t s, Workspace& out);

// Row-streaming kernels (conv_strip.hip).  PG_E_UNSUP = "not this shape": the caller keeps its tile kernel.  ``name`` receives the
// kernel symbol for pg_debug_last_conv_kernel.
int launch_conv_strip(ConvP& p, hipStream_t s, char* name, size_t name_len);
int launch_wgrad_strip(WgP& p, hipStream_t s, char* name, size_t name_len);
int launch_conv_strip_pn_torgb(const float* x, const float* w, const float* bias, float* y, float* r,
                               const float* t_w, const float* t_b, float t_scale, float* img,
                               int N, int C, int H, int W, int Cin, int Cout, float scale, float slope, float eps,
                               hipStream_t s, char* name, size_t name_len);
int launch_conv_strip_masked_rgb_bwd(const float* gz, const float* wt, const unsigned char* mask_bytes, float mask_slope, float* y,
                                     const float* rgb_w, float rgb_scale, float* gimg,
                                     const float* img, float* rgb_dw, float* rgb_db,
                                     int N, int C, int H, int W, int Cin, int Cout, float scale,
                                     hipStream_t s, char* name, size_t name_len);
int launch_conv_strip_fromrgb(const float* img, const float* rgb_w, const float* rgb_b, float rgb_scale, float rgb_slope,
                              unsigned char* x_signs, const float* w, const float* bias, float* y, unsigned char* y_signs,
                              int N, int C, int H, int W, int Cmid, int Cout, float scale, float slope,
                              hipStream_t s, char* name, size_t name_len);

}  // namespace pgk
