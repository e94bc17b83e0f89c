// Counter-based RNG (Philox4x32-10, Salmon et al. SC'11) for dropout keep-masks and multinomial uniforms.
// Stateless: bits are a pure function of (seed, stream, step, element index), so the backward pass regenerates
// the forward masks instead of storing them (the attention mask alone would be T*B*R*A bytes per rollout).
#pragma once
#include "icz_common.h"

namespace icz {

enum RngStream : uint32_t { RNG_EMB = 1, RNG_ATT = 2, RNG_OUT = 3, RNG_UNIFORM = 4, RNG_SS_GATE = 5, RNG_SS_DRAW = 6 };

struct uint4_ { uint32_t x, y, z, w; };

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

__host__ __device__ inline uint4_ philox4x32_10(uint4_ c, uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = mulhi32(M0, c.x), lo0 = M0 * c.x;
        uint32_t hi1 = mulhi32(M1, c.z), lo1 = M1 * c.z;
        uint4_ n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += W0;
        k1 += W1;
    }
    return c;
}

// 32 random bits for the aligned group of 32 elements that contains element `idx`.
__host__ __device__ inline uint32_t rng_group_bits(uint64_t seed, uint32_t stream, uint32_t step, uint64_t idx) {
    uint64_t g = idx >> 7;                 // 128 elements per Philox call
    uint4_ c = {(uint32_t)g, (uint32_t)(g >> 32), step, stream};
    uint4_ r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    uint32_t w = (uint32_t)((idx >> 5) & 3);
    return w == 0 ? r.x : (w == 1 ? r.y : (w == 2 ? r.z : r.w));
}

// keep-bit of nn.Dropout(p=0.5) for element idx
__host__ __device__ inline bool rng_keep(uint64_t seed, uint32_t stream, uint32_t step, uint64_t idx) {
    return (rng_group_bits(seed, stream, step, idx) >> (idx & 31)) & 1u;
}

// uniform in [0,1) with 24 random bits (one per (step, row))
__host__ __device__ inline float rng_uniform(uint64_t seed, uint32_t step, uint64_t row, uint32_t stream = RNG_UNIFORM) {
    uint4_ c = {(uint32_t)row, (uint32_t)(row >> 32), step, stream};
    uint4_ r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(r.x >> 8) * (1.0f / 16777216.0f);
}

// How a kernel obtains a dropout keep-mask.
struct DropCfg {
    int mode;               // 0 = eval (no dropout), 1 = explicit mask array, 2 = Philox
    const uint8_t* mask;    // mode 1: keep flags for this step, element-indexed
    const uint64_t* seed_p; // mode 2: seed lives in device memory (a captured graph is replayed with new seeds)
    uint32_t stream, step;
    int row0;               // forward kernels of a merged greedy + sampled chain: rows < row0 are evaluation-mode rows (no dropout), row
                            // r >= row0 is row r - row0 of the sampled batch (mask / Philox index); 0 = every row is a training-mode row
    __device__ __forceinline__ bool keep(uint64_t idx) const {
        if (mode == 1) return mask[idx] != 0;
        return rng_keep(*seed_p, stream, step, idx);
    }
    // 4 consecutive elements starting at idx (idx % 4 == 0): bit j set = keep element idx + j
    __device__ __forceinline__ uint32_t keep4(uint64_t idx) const {
        if (mode == 1) {
            uint32_t m = *reinterpret_cast<const uint32_t*>(mask + idx);
            return ((m & 0xFFu) ? 1u : 0u) | ((m & 0xFF00u) ? 2u : 0u) | ((m & 0xFF0000u) ? 4u : 0u) | ((m & 0xFF000000u) ? 8u : 0u);
        }
        return (rng_group_bits(*seed_p, stream, step, idx) >> (idx & 31)) & 0xFu;
    }
};

}  // namespace icz
