// fp32 GEMMs on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered fma chain).
//
//   C[M x N] = sum over segments s of  A_s[M x K_s] * B_s[K_s x N]
//
// Three operand layouts, one tiling:
//   NT  (forward):  A_s(m,k) = X[m*lda + k],   B_s(k,n) = W[n*ldw + k]     y = x W^T
//   NN  (dgrad):    A_s(m,k) = dY[m*lda + k],  B_s(k,n) = W[k*ldw + n]     dx = dy W
//   TN  (wgrad):    A_s(m,k) = dY[k*lda + m],  B_s(k,n) = X[k*ldw + n]     dW = dy^T x
//
// Tiling: one 256-thread workgroup = 4 waves computes a 64 x 64 tile of C over a contiguous range of
// 64-deep K chunks (split-K over blockIdx.z).  Wave w owns columns [16w, 16w+16) and all 64 rows
// (MT = 4 row tiles of 16) -> 4 independent 16x16 accumulators per wave, which is what the 16x16x4 f32
// MFMA needs to issue back to back (40-cycle dependent latency vs 32-cycle issue).
//   * the A chunk (64 rows x 64 k, shared by the 4 waves) is staged through LDS, row stride 72 dwords:
//     with that stride the A-fragment ds_read_b128 (lane (i,q) reads row i, dwords 16s+4q..+3) is
//     conflict-free in every one of the instruction's four 16-lane groups;
//   * the B operand is streamed straight into registers (each wave reads only its own 16 columns; for NT
//     that is the weight matrix, read exactly once from HBM per workgroup row) -- LDS would be a pure
//     round trip for it;
//   * next chunk's global loads are issued before the current chunk's MFMAs (register double buffer for
//     B, LDS double buffer for A, one barrier per chunk).
// Split-K partials go to slabs [z][M][N]; the consumer kernels (LSTM pointwise, attention, argmax ...) sum
// the slabs in fixed z order, so results are bitwise reproducible run to run (no float atomics).
#pragma once
#include "icz_common.h"

namespace icz {

constexpr int GEMM_MAX_SEG = 4;
constexpr int GEMM_BM = 64, GEMM_BN = 64, GEMM_BK = 64;
constexpr int GEMM_LDS_STRIDE = 72;   // dwords per staged A row (64 + 8 pad)

enum GemmLayout { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };

struct GemmSeg {
    const float* A;
    const float* B;
    int lda, ldb;
    int K;
    // optional row gather for A (NT only): row m of A is A + gather[m]*lda, with ReLU applied on load
    const int64_t* gather;
};

struct GemmArgs {
    GemmSeg seg[GEMM_MAX_SEG];
    int nseg;
    int M, N;
    float* out;          // nsplit == 1: C (row stride ldo) ; else slabs [nsplit][M][N]
    int ldo;
    const float* bias;   // nsplit == 1 only, per column, may be null
    int nsplit;
    int chunks_per_split;
    int accumulate;      // nsplit == 1 only: C += result (used by wgrad accumulation)
    int spread;          // set by gemm_f32 (NN, skinny shapes): spread the next stage's loads over the stage
};

int gemm_f32(GemmLayout layout, const GemmArgs& a, hipStream_t stream);
// picks a split-K factor so that the launch has about `target_wgs` workgroups
int gemm_pick_split(const GemmArgs& a, int target_wgs, GemmLayout layout = GEMM_NT);
// the same for launches with many tiles: balances whole rounds of workgroups over the 256 CUs (see gemm_f32.hip)
int gemm_pick_split_balanced(const GemmArgs& a, GemmLayout layout, size_t slab_capacity_floats);
size_t gemm_slab_floats(int M, int N, int nsplit);
// clamp a caller-chosen split so that no split is empty (stage depth depends on layout and shape)
int gemm_normalize_split(GemmLayout layout, const GemmArgs& a, int nsplit);

}  // namespace icz
