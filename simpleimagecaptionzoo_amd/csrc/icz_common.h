// Shared helpers for the libicz HIP sources (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/icz.h"

namespace icz {

void set_error(const char* fmt, ...);

#define ICZ_CHECK_HIP(expr)                                                                    \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) {                                                                \
            icz::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return ICZ_ERR_HIP;                                                                \
        }                                                                                      \
    } while (0)

#define ICZ_REQUIRE(cond, ...)                    \
    do {                                          \
        if (!(cond)) {                            \
            icz::set_error(__VA_ARGS__);          \
            return ICZ_ERR_INVALID;               \
        }                                         \
    } while (0)

#define ICZ_TRY(expr)                   \
    do {                                \
        int s_ = (expr);                \
        if (s_ != ICZ_OK) return s_;    \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
// padded vocabulary size (row stride of logits, rows of the materialised predict weight): a multiple of 64 so that
// V can be the K dimension of a GEMM on the fast (tail-free) path; pad entries are kept at zero
static inline int pad_vocab(int V) { return (V + 63) & ~63; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

}  // namespace icz
