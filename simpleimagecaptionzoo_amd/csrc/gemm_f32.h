// fp32 GEMMs on the f32-input MFMA (v_mfma_f32_16x16x4_f32: exact f32 products, k-ordered fma chain).
//
//   C[M x N] = sum over segments s of  A_s[M x K_s] * B_s[K_s x N]
//
// Three operand layouts, one tiling:
//   NT  (forward):  A_s(m,k) = X[m*lda + k],   B_s(k,n) = W[n*ldw + k]     y = x W^T
//   NN  (dgrad):    A_s(m,k) = dY[m*lda + k],  B_s(k,n) = W[k*ldw + n]     dx = dy W
//   TN  (wgrad):    A_s(m,k) = dY[k*lda + m],  B_s(k,n) = X[k*ldw + n]     dW = dy^T x
//
// Tiling: one 256-thread workgroup = 4 waves computes a 64 x 64 tile of C over a contiguous range of K stages
// (split-K over blockIdx.z; stage depth 128, or 64 for short / ragged K).  NT: wave w owns columns [16w, 16w+16) and
// all 64 rows (MT = 4 row tiles of 16) -> 4 independent 16x16 accumulators per wave, which is what the 16x16x4 f32
// MFMA needs to issue back to back (40-cycle dependent latency vs 32-cycle issue).
//   * the A stage (64 rows x BK, shared by the 4 waves) is staged through LDS, row stride BK + 8 dwords (the A-fragment
//     ds_read_b128 of lane (i,q) reads row i, dwords 16s+4q..+3);
//   * the B operand is streamed straight into registers (each wave reads only its own 16 columns; for NT that is the
//     weight matrix, read exactly once per workgroup row) -- LDS would be a pure round trip for it;
//   * the next stage's global loads are in flight behind the current stage's MFMAs (register double buffer for B, LDS
//     double buffer for A, one barrier per stage); fragment reads from LDS run one or two k-steps ahead of their MFMAs,
//     pinned with scheduling fences.  At the decoder-step shape (M <= 64: one workgroup per CU, all in lockstep) the
//     weight loads are issued one per k-group instead of in one burst at the top of the stage, which otherwise leaves
//     the fabric idle while the matrix pipe runs (NT and NN).
//   * NN mirrors NT (the weight tile goes through LDS, interleaved column tiles); TN streams both operands as float4
//     along their output index (2 loads -> 16 MFMAs), and for large outputs uses a 128 x 128 tile with both operand
//     tiles in LDS (half the operand bytes per MFMA).
// Split-K partials go to slabs [z][M][N]; the consumer kernels (LSTM pointwise, attention, argmax ...) sum
// the slabs in fixed z order, so results are bitwise reproducible run to run (no float atomics).
#pragma once
#include "icz_common.h"

namespace icz {

constexpr int GEMM_MAX_SEG = 4;
constexpr int GEMM_BM = 64, GEMM_BN = 64, GEMM_BK = 64;

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
    const int* live;     // step_dead(live): the kernel returns at entry without writing anything (icz_common.h); null = always run
    const int* rows_live; // 128 x 128 split-precision kernel only, optional: a device count of the leading (t, b) rows that matter (the steps a
                         // sampled rollout really ran) -- TN: K is cut there (rounded up to 32), NN: row tiles behind it return at entry.
                         // Rows behind the count must hold finite values whose products are zero or never read (BPTT: d gates = 0)
};

int gemm_f32(GemmLayout layout, const GemmArgs& a, hipStream_t stream);
// picks a split-K factor so that the launch has about `target_wgs` workgroups
int gemm_pick_split(const GemmArgs& a, int target_wgs, GemmLayout layout = GEMM_NT);
// the same for launches with many tiles: balances whole rounds of workgroups over the 256 CUs (see gemm_f32.hip)
int gemm_pick_split_balanced(const GemmArgs& a, GemmLayout layout, size_t slab_capacity_floats);
size_t gemm_slab_floats(int M, int N, int nsplit);
// clamp a caller-chosen split so that no split is empty (stage depth depends on layout and shape)
int gemm_normalize_split(GemmLayout layout, const GemmArgs& a, int nsplit);
// the largest split <= nsplit whose slabs fit the given capacity
int gemm_fit_split(GemmLayout layout, const GemmArgs& a, int nsplit, size_t capacity_floats);

// gemm_resident_x3.hip: NT at 33..64 rows (and, as one merged decode chain, 65..128 rows) with split-precision operands and the activations of a workgroup's k range resident in
// LDS (LSTM gates, vocabulary projection, per-step dgrad on transposed weights): fixed decomposition, nsplit = stages / stages per range
bool gemm_resident_x3_fits(const GemmArgs& a);
int gemm_resident_x3_stages(const GemmArgs& a);      // 64-deep stages per workgroup: 8 (a 512-deep k range on one column tile, <= 64 rows, round 6) or 4 (256 deep, two tiles), 0 = shape not taken
int gemm_resident_x3_nsplit(const GemmArgs& a);
int gemm_resident_x3(const GemmArgs& a, hipStream_t stream);
// two independent problems (each in the kernel's four-stage decomposition, N >= 512) as ONE launch; slabs [nsplit][M][N] go to each
// problem's `out`, nsplit = gemm_resident_x3_nsplit (the two recurrent dgrad products of a BPTT step)
bool gemm_resident_x3_pair_fits(const GemmArgs& a, const GemmArgs& b);
int gemm_resident_x3_pair(const GemmArgs& a, const GemmArgs& b, hipStream_t stream);
// Vocabulary projection of a decoder step: logits[rows, V] = x[rows, H] w_pred^T + bias, w_pred stored [Vp, H] with zero pad rows.
// At 33 - 64 rows the un-split GEMM is V / 64 workgroups that each re-read the whole activation matrix (as many bytes as their
// weights: 25 us for 41 MB at V = 10102); the resident-activation kernel over the padded vocabulary takes 256 columns and a
// quarter of K per workgroup (17.5 us) and leaves split-K slabs [ns][rows][Vp] in `ws`, which a slab-summing consumer
// (argmax_part_kernel, sample_select_kernel) adds up together with the bias.  pred_nsplit == nullptr: the caller's consumer
// needs finished logits -> always the un-split GEMM into `logits` (row stride ldl).  Otherwise *pred_nsplit reports what was
// done: 1 = finished logits in `logits`, > 1 = that many slabs in `ws` (no bias yet).  ICZ_PREDICT_SLABS=0 keeps the un-split GEMM.
int gemm_predict(const float* x, int H, const float* w_pred, const float* bias, int rows, int V, int Vp, float* logits, int ldl,
                 float* ws, size_t ws_cap_floats, int* pred_nsplit, hipStream_t st, const int* live = nullptr);

// gemm_big_x3.hip: the many-row split-precision products on large tiles, one barrier per 16-deep k-step (cfg 1: 256 x 256 / eight waves,
// 2: 128 x 256, 3: 256 x 128, 4: 128 x 128 three workgroups per CU, 5: 256 x 256 with the two wave halves half a step apart);
// `a` prepared as for the 128 x 128 kernel (nsplit, chunks_per_split in 128-deep chunks).  gemm_big_cfg picks per shape (0: the
// 128 x 128 two-barrier kernel); gemm_f32 routes through it.
int gemm_big_x3(GemmLayout layout, const GemmArgs& a, int cfg, hipStream_t stream);
int gemm_big_cfg(GemmLayout layout, const GemmArgs& a);
int gemm_big_switch();               // ICZ_GEMM_BIG, or what gemm_set_big_cfg put in its place (-1: per shape, 0: off, 1..5: forced)
void gemm_set_big_cfg(int cfg);      // -2: back to the environment's value
// Weight gradients that share d y:  out_j[M x cols_j] = dY^T X_j  over K rows, j < ngroups <= 4, as ONE launch (column groups of one
// output space; cols_j % 256 == 0).  rows_live as in GemmArgs.  _fits: the shape is taken (else the caller issues the products one by one).
constexpr int GEMM_MAX_COLGROUPS = 4;
struct GemmColGroup { const float* B; int ldb; int cols; float* out; int ldo; };
bool gemm_tn_grouped_fits(int M, int K, const GemmColGroup* groups, int ngroups);
int gemm_tn_grouped(const float* dY, int ldy, int M, int K, const GemmColGroup* groups, int ngroups, const int* rows_live, hipStream_t stream);

// Library switches, read from the environment once at first use (defaults = the product configuration): ICZ_GEMM_TN_X3,
// ICZ_GEMM_NN_X3, ICZ_GEMM_NT_X3BIG, ICZ_GEMM_RESIDENT_X3 (0: the fp32-input MFMA kernels -- bench.py's fp32_mfma_gemms leg),
// ICZ_GEMM_RESIDENT_M128 (0: 65..128 rows go to the 128 x 128-tile kernel instead of the 128-row resident kernel),
// ICZ_PREDICT_SLABS (0: un-split vocabulary projection -- the slab A/B test), ICZ_PROF_EVERY (event pairs on every n-th launch),
// ICZ_GEMM_BIG (unset / -1: gemm_big_cfg's choice per shape; 0: the 128 x 128 two-barrier kernel everywhere; 1..5: that large-tile configuration everywhere).
// (round 6) TN products too small to fill the chip on 128 x 128 tiles (fewer than 256 of them: the weight gradients of the two attention
// projections, 1024 x 1024 over 1280 rows and 1024 x 2048 over 2304) on the large-tile split-precision kernel with split-K slabs
// [nsplit][M][N] (the caller sums them: slab_reduce_kernel) instead of the fp32-MFMA 64 x 64 kernel.  gemm_tn_split_pick: the split (1 =
// shape not taken); gemm_tn_split: the launch (K a multiple of 16, splits on 128-deep chunk boundaries, rows_live as in gemm_tn_grouped).
int gemm_tn_split_pick(int M, int N, int K);
int gemm_tn_split(const float* dY, int ldy, int M, const float* X, int ldx, int N, int K, int nsplit, float* slabs, const int* rows_live, hipStream_t st);
struct GemmSwitches { bool tn_x3, nn_x3, nt_x3big, resident_x3, resident_m128, predict_slabs, resident_k512; unsigned prof_every; int big_cfg; };
const GemmSwitches& gemm_switches();

}  // namespace icz
