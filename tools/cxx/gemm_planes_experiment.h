// gemm_planes.hip: split-precision GEMM on operands cut into bf16 planes ahead of time (see the .hip for layout and kernel).
#pragma once
#include "gemm_f32.h"

namespace icz {

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// bytes of the three planes of an operand of `rows` x K (rows padded to 256, K to 16)
size_t planes_bytes(int rows, int K);
// cut src into planes: kmajor = false: element(r, k) = src[r ld + k];  true: src[k ld + r].  The operand has `ks_total` k-steps of
// 16 in all (several K segments = several calls with their own `ks_off`); `rows_padded` = round_up(rows, 256).
int planes_pack(const float* src, int ld, int rows, int K, bool kmajor, void* dst, int rows_padded, int ks_total, int ks_off, bool relu,
                hipStream_t st);

struct PlanesGemm {
    const void* A;          // planes of the M-side operand
    const void* B;          // planes of the N-side operand
    int Mp, Np, KS;
    int M, N;
    float* out;             // nsplit == 1: C (row stride ldo);  else slabs [nsplit][M][N]
    int ldo;
    const float* bias;
    int accumulate;
    int nsplit;             // over the KS k-steps (no empty splits)
    int config;             // 0: 256 x 256 tile, 8 waves, 3-slot ring; 1: 128 x 256, 4 waves, 2 slots; 2: 128 x 128, 4 waves, 3 slots; 3: 256 x 256, 2 slots
};
int gemm_planes(const PlanesGemm& g, hipStream_t st);

}  // namespace icz
