// Split-precision activation planes: the producer of an fp32 activation tensor that feeds a skinny decoder-step GEMM
// (gemm_skinny_x3.hip) also writes its three bf16 pieces x = x0 + x1 + x2 (each the bf16 rounding of what the previous ones
// left: 24 mantissa bits), so that the GEMM stages them into LDS with plain 16-byte copies instead of re-splitting the same
// activations in every one of its ~256 workgroups (measured: 31 % of the kernel's time).
// Layout: plane p of element (row, k) of a [rows, ld] tensor at planes[p * stride + row * ld + sp_perm(k)] (16-bit units);
// sp_perm reorders k inside every 32-block into the order in which the MFMA fragments consume it (lane quarter q holds
// k = 4 q + e for e < 4 and 16 + 4 q + e - 4 above), so a fragment is 8 consecutive elements.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icz {

struct Planes {              // companion of one fp32 buffer (null base = none)
    unsigned short* base;
    long long stride;        // elements between planes
};

__host__ __device__ __forceinline__ int sp_perm(int k) {
    const int kk = k & 31;
    return (k & ~31) + 8 * ((kk & 15) >> 2) + 4 * (kk >> 4) + (kk & 3);
}

__device__ __forceinline__ unsigned short sp_bf16_rn(float x) {      // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(x);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// the three pieces of one value (same arithmetic as the GEMM's in-register split)
__device__ __forceinline__ void sp_split1(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    p0 = sp_bf16_rn(x);
    float r = x - __uint_as_float((uint32_t)p0 << 16);
    p1 = sp_bf16_rn(r);
    r -= __uint_as_float((uint32_t)p1 << 16);
    p2 = sp_bf16_rn(r);
}
__device__ __forceinline__ void sp_store1(const Planes& pl, size_t row_off, int k, float x) {
    if (!pl.base) return;
    unsigned short a, b, c;
    sp_split1(x, a, b, c);
    unsigned short* o = pl.base + row_off + sp_perm(k);
    o[0] = a; o[pl.stride] = b; o[2 * pl.stride] = c;
}
// four consecutive k (k % 4 == 0): they stay consecutive under sp_perm -> one 8-byte store per plane
__device__ __forceinline__ void sp_store4(const Planes& pl, size_t row_off, int k, float x0, float x1, float x2, float x3) {
    if (!pl.base) return;
    unsigned short a[4], b[4], c[4];
    sp_split1(x0, a[0], b[0], c[0]); sp_split1(x1, a[1], b[1], c[1]); sp_split1(x2, a[2], b[2], c[2]); sp_split1(x3, a[3], b[3], c[3]);
    typedef __attribute__((ext_vector_type(2))) uint32_t u2;
    unsigned short* o = pl.base + row_off + sp_perm(k);
    *reinterpret_cast<u2*>(o) = (u2){(uint32_t)a[0] | ((uint32_t)a[1] << 16), (uint32_t)a[2] | ((uint32_t)a[3] << 16)};
    *reinterpret_cast<u2*>(o + pl.stride) = (u2){(uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16)};
    *reinterpret_cast<u2*>(o + 2 * pl.stride) = (u2){(uint32_t)c[0] | ((uint32_t)c[1] << 16), (uint32_t)c[2] | ((uint32_t)c[3] << 16)};
}

}  // namespace icz
