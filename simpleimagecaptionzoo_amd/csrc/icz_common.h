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
// event pairs around groups of small kernels (icz_kprof_begin / icz_kprof_end, include/icz.h); no-ops unless enabled, never inside a capture
enum { KP_ATTENTION = 0, KP_GREEDY_SELECT = 1, KP_SAMPLE_SELECT = 2, KP_LSTM_POINT = 3, KP_GROUPS = 4 };
void kprof_mark(int group, bool begin, hipStream_t st);
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

// Early-out of the rollout / BPTT steps behind the reference's break (`if unfinished.sum() == 0: break`, BUTD_Model.py:233,
// AoA_Model.py:400, NIC_Model.py:150): `live` points at the number of rows still unfinished after the PREVIOUS step of the
// sampled rollout (a device counter written by that step's sample_select_kernel); 0 = the reference never ran this step.  Every
// kernel of such a step returns at entry (producers of buffers that a later batched GEMM reads write zeros first); null = always
// live (greedy / beam / XE).  The address is a kernel argument: one scalar load, uniform branch.
__device__ __forceinline__ bool step_dead(const int* live) { return live != nullptr && *live == 0; }
// The same test in two parts for the hot kernels: live_flag() at the TOP of the kernel issues the scalar load, flag_dead() sits
// BEHIND the kernel's first batch of global loads and in front of its first write -- the flag's round trip passes while those
// loads are in flight, so a live step does not wait for it (tested at the top of every 5 - 9 us kernel it cost 0.3 us per launch:
// +1.2 % on the SCST step, same box, round 5), and a dead step leaves with loads outstanding.  The empty asm is a compiler
// barrier: the vector loads in front of it are not sunk below the branch.
__device__ __forceinline__ int live_flag(const int* live) { return live != nullptr ? *live : 1; }
__device__ __forceinline__ bool flag_dead(int flag) {
    asm volatile("" ::: "memory");
    return flag == 0;
}

}  // namespace icz
