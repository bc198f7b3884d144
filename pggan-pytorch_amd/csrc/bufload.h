// Raw buffer loads for the global->LDS fetch stages (gfx950).  An offset at or beyond num_records returns zeros in hardware,
// so "outside the image / beyond the channel block" is encoded as the offset OOB instead of an exec-mask branch around
// every load: the fetch stage becomes straight-line code the scheduler can place under the MFMAs.
#pragma once
#include <hip/hip_runtime.h>

typedef unsigned int pg_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned PG_OOB = 0x80000000u;                   // never a valid byte offset: every span is checked < 2 GiB on the host

__device__ __forceinline__ __amdgpu_buffer_rsrc_t pg_make_rsrc(const void* base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);   // raw, 32-bit data format
}

// one float4 at base + voff + soff (bytes); voff per lane, soff wave-uniform
__device__ __forceinline__ float4 pg_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    const pg_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// XCD-aware workgroup order (MI355X: 8 XCDs x 32 CUs, one private L2 per XCD; the dispatcher places block b on XCD b % 8).
// Neighbouring tiles share halo rows / columns, so each XCD gets a CONTIGUOUS range of the tile sequence instead of
// every 8th tile: the halo is then re-read from that XCD's L2 instead of from HBM.  Bijective for any grid size.
__device__ __forceinline__ unsigned pg_xcd_remap(unsigned b, unsigned n)
{
    const unsigned q = n >> 3, r = n & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
