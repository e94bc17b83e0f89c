// fp32 MFMA GEMMs (see gemm_f32.h for the tiling rationale).  gfx950 only.
#include <stdlib.h>

#include <array>
#include <mutex>
#include <set>
#include <vector>

#include "gemm_f32.h"

namespace icz {

// ------------------------------------------------------------------------------------------------
// chunk bookkeeping: the K dimension is the concatenation of the segments, cut into 64-deep chunks
// (a segment's last chunk may be partial; loads beyond K are zero-filled).
static int total_chunks(const GemmArgs& a, int bk = GEMM_BK) {
    int t = 0;
    for (int s = 0; s < a.nseg; ++s) t += cdiv(a.seg[s].K, bk);
    return t;
}

// Walks the chunk sequence of one split: (segment, k offset) advance with scalar arithmetic only; the segment's
// operand descriptors are re-read from the kernel arguments only when the segment changes.
template <int BK>
struct ChunkCursorT {
    int seg, k0;
    const float* A; const float* B;
    int lda, ldb, K;
    __device__ __forceinline__ void load_seg(const GemmArgs& a) {
        const GemmSeg& g = a.seg[seg];
        A = g.A; B = g.B; lda = g.lda; ldb = g.ldb; K = g.K;
    }
    __device__ __forceinline__ void seek(const GemmArgs& a, int chunk) {
        seg = 0;
        int c = chunk;
#pragma unroll
        for (int s = 0; s < GEMM_MAX_SEG - 1; ++s) {
            if (seg == s && s < a.nseg - 1) {
                int n = (a.seg[s].K + BK - 1) / BK;
                if (c >= n) { c -= n; seg = s + 1; }
            }
        }
        k0 = c * BK;
        load_seg(a);
    }
    __device__ __forceinline__ void next(const GemmArgs& a) {
        k0 += BK;
        if (k0 >= K && seg < a.nseg - 1) { ++seg; k0 = 0; load_seg(a); }
    }
};
using ChunkCursor = ChunkCursorT<GEMM_BK>;

template <int BK = GEMM_BK>
__device__ __forceinline__ int split_range(const GemmArgs& a, int z, int* c_end_out) {
    const int c_begin = z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    int tot = 0;
#pragma unroll
    for (int s = 0; s < GEMM_MAX_SEG; ++s)
        if (s < a.nseg) tot += (a.seg[s].K + BK - 1) / BK;
    *c_end_out = c_end > tot ? tot : c_end;
    return c_begin;
}

// unconditional 16-byte load from a (clamped, always valid) address; zero-filled by a select when !ok.
// No branch -> the compiler issues a chunk's loads back to back and waits only at their first use.
// TAIL = false (every segment's K is a multiple of 64, the production shapes): a plain load, so nothing consumes
// the value before the MFMA block and the loads stay in flight behind it.
template <bool TAIL>
__device__ __forceinline__ f32x4 ld4z(const float* p, bool ok) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p);
    if (TAIL) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        return ok ? v : z;
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// NT: C = X W^T.  A chunk through LDS (shared by the 4 waves), W fragments straight to registers.
// Lane (i = lane&15, q = lane>>4).  MFMA operand maps (16x16x4 f32): A[i][k=q], B[k=q][j=i];
// one float4 along k per lane feeds 4 MFMAs (component c <-> k = 16s + 4q + c, same on both operands).
// Wave w owns NTW adjacent 16-column tiles and all MT row tiles.  BKC = K depth of one pipeline stage (64 or 128):
// per stage a wave issues BKC/4 * MT * NTW MFMAs between two barriers; the deeper stage halves the per-stage
// overhead (barrier, waits, address arithmetic) per MFMA.
// Rows beyond M / columns beyond N are read from a clamped (valid) row: they only feed accumulator rows /
// columns that are never stored, so no zero-fill is needed; only the K tail must be zero (TAIL variant).
template <int MT, int NTW, bool TAIL, int BKC, int NW = 4, bool SPREAD = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_kernel(GemmArgs a) {
    const int lflag = live_flag(a.live);
    constexpr int BN = 16 * NW * NTW;
    constexpr int NTHR = 64 * NW;
    constexpr int LDSS = BKC + 8;                 // dwords per staged A row; (LDSS/4) mod 16 == 2 -> conflict-free b128 reads
    constexpr int NS = BKC / 16;                  // k-groups (of 16) per stage
    __shared__ __attribute__((aligned(16))) float lds[2][MT * 16 * LDSS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * (MT * 16), z = blockIdx.z;
    // stage range of this split (stages are counted in units of BKC; a.chunks_per_split is in the same unit)
    int tot = 0;
#pragma unroll
    for (int s = 0; s < GEMM_MAX_SEG; ++s)
        if (s < a.nseg) tot += (a.seg[s].K + BKC - 1) / BKC;
    const int c_begin = z * a.chunks_per_split;
    const int c_end = min(tot, c_begin + a.chunks_per_split);

    int ncol[NTW];
    size_t ncol_c[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        ncol[nt] = n0 + (wave * NTW + nt) * 16 + li;        // this lane's W row (= output column) of tile nt
        ncol_c[nt] = ncol[nt] < a.N ? ncol[nt] : a.N - 1;
    }
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int C4 = BKC / 4;                              // float4 per staged row
    constexpr int XL = (MT * 16 * C4 + NTHR - 1) / NTHR;           // float4 staging loads per thread per stage
    f32x4 xr[XL];
    size_t xrow[XL];
    int xlds[XL], xkk[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int idx = tid + NTHR * j;
        int row = idx / C4;
        if (row > MT * 16 - 1) row = MT * 16 - 1;
        int m = m0 + row;
        if (m > a.M - 1) m = a.M - 1;
        xrow[j] = (size_t)m;
        xkk[j] = 4 * (idx % C4);
        xlds[j] = row * LDSS + xkk[j];
    }

    // stage cursor: (segment, k0) + per-lane operand pointers; inside a segment a step is "pointer += BKC floats"
    int seg = 0, k0 = 0, segK = 0;
    const float *segA = nullptr, *segB = nullptr;
    int lda = 0, ldb = 0;
    const float* wp[NTW];
    const float* xp[XL];
    auto load_seg = [&]() {
        const GemmSeg& g = a.seg[seg];
        segA = g.A; segB = g.B; lda = g.lda; ldb = g.ldb; segK = g.K;
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) wp[nt] = segB + ncol_c[nt] * ldb + k0 + 4 * lq;
#pragma unroll
        for (int j = 0; j < XL; ++j) xp[j] = segA + xrow[j] * lda + k0 + xkk[j];
    };
    auto seek = [&](int stage) {
        int c = stage;
        seg = 0;
#pragma unroll
        for (int s = 0; s < GEMM_MAX_SEG - 1; ++s) {
            if (seg == s && s < a.nseg - 1) {
                const int nst = (a.seg[s].K + BKC - 1) / BKC;
                if (c >= nst) { c -= nst; seg = s + 1; }
            }
        }
        k0 = c * BKC;
        load_seg();
    };
    auto advance = [&]() {
        k0 += BKC;
        if (k0 >= segK && seg < a.nseg - 1) { ++seg; k0 = 0; load_seg(); }
        else {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) wp[nt] += BKC;
#pragma unroll
            for (int j = 0; j < XL; ++j) xp[j] += BKC;
        }
    };
    // X first, W second: the LDS store of X only has to wait for the X loads (vmcnt is in issue order)
    auto load_stage = [&](f32x4 (&w)[NTW][NS]) {
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            if (TAIL) {
                const int k = k0 + xkk[j];
                xr[j] = ld4z<true>(k < segK ? xp[j] : xp[j] - (k - (segK - 4)), k < segK);
            } else {
                xr[j] = *reinterpret_cast<const f32x4*>(xp[j]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (TAIL) {
                    const int k = k0 + 16 * s + 4 * lq;
                    w[nt][s] = ld4z<true>(k < segK ? wp[nt] + 16 * s : wp[nt] + 16 * s - (k - (segK - 4)), k < segK);
                } else {
                    w[nt][s] = *reinterpret_cast<const f32x4*>(wp[nt] + 16 * s);
                }
            }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < XL; ++j)
            if ((MT * 16 * C4) % NTHR == 0 || tid + NTHR * j < MT * 16 * C4)
                *reinterpret_cast<f32x4*>(&lds[buf][xlds[j]]) = xr[j];
    };
    // A-fragment reads for k-group s+1 are issued before the MFMAs of group s (register double buffer).  The scheduling
    // fences pin that order: left alone, hipcc issues the reads of two groups back to back AFTER the previous group's MFMAs
    // and the first MFMA of every pair of groups waits out the whole LDS latency with the matrix pipe idle.
    auto compute = [&](int buf, const f32x4 (&w)[NTW][NS]) {
        f32x4 af[2][MT];
        const float* base = &lds[buf][li * LDSS + 4 * lq];
#pragma unroll
        for (int t = 0; t < MT; ++t) af[0][t] = *reinterpret_cast<const f32x4*>(base + t * 16 * LDSS);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < NS - 1) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    af[(s + 1) & 1][t] = *reinterpret_cast<const f32x4*>(base + t * 16 * LDSS + 16 * (s + 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1][t][e], w[nt][s][e], acc[t][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // Software pipeline (double-buffered LDS + two W register sets): while stage i is multiplied, the loads of
    // stage i+1 are in flight; they are stored to the other LDS buffer after the MFMAs.  One barrier per stage.
    const int n = c_end - c_begin;
    f32x4 wa[NTW][NS], wb[NTW][NS];
    if (n <= 0 && lflag == 0) return;
    if (n > 0) {
        seek(c_begin);
        load_stage(wa);
        if (flag_dead(lflag)) return;           // behind the first stage's loads, in front of the first write (icz_common.h)
        store_stage(0);
        __syncthreads();
        auto body = [&](int buf, const f32x4 (&wuse)[NTW][NS], f32x4 (&wload)[NTW][NS]) {
            advance();
            if (SPREAD && !TAIL && NTW == 1) {
                // Decoder-step shape (M = 64: one workgroup per CU, all in lockstep).  Issued in one burst at the top of the
                // stage, the loads put 256 x 64 KB on the fabric at once and then leave it idle while the matrix pipe runs:
                // memory time and MFMA time add up (2.65 us per stage against 1.86 us of MFMA issue, while a pure streaming
                // kernel moves the same weight bytes in 1.3 us).  So only the X loads go out at the top (their LDS store is
                // due at the end of the stage); the next stage's W loads follow one per k-group, between the MFMA groups.
#pragma unroll
                for (int j = 0; j < XL; ++j) xr[j] = *reinterpret_cast<const f32x4*>(xp[j]);
                __builtin_amdgcn_sched_barrier(0);
                f32x4 af[2][MT];
                const float* base = &lds[buf][li * LDSS + 4 * lq];
#pragma unroll
                for (int t = 0; t < MT; ++t) af[0][t] = *reinterpret_cast<const f32x4*>(base + t * 16 * LDSS);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s < NS - 1) {
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            af[(s + 1) & 1][t] = *reinterpret_cast<const f32x4*>(base + t * 16 * LDSS + 16 * (s + 1));
                    }
                    wload[0][s] = *reinterpret_cast<const f32x4*>(wp[0] + 16 * s);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s & 1][t][e], wuse[0][s][e], acc[t][0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                load_stage(wload);
                // keep the loads HERE: without the fence hipcc sinks them below the MFMAs, next to their first use
                __builtin_amdgcn_sched_barrier(0);
                compute(buf, wuse);
            }
            store_stage(buf ^ 1);
            __syncthreads();
        };
        int i = 0;
        for (; i + 2 < n; i += 2) {
            body(0, wa, wb);
            body(1, wb, wa);
        }
        if (i + 1 < n) {            // two stages left
            body(0, wa, wb);
            compute(1, wb);
        } else {                    // one stage left
            compute(0, wa);
        }
    }

    // epilogue: acc[t][nt][r] <-> row m0 + 16t + 4q + r, column ncol[nt]
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        if (ncol[nt] >= a.N) continue;
        if (a.nsplit == 1) {
            const float b = a.bias ? a.bias[ncol[nt]] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m0 + 16 * t + 4 * lq + r;
                    if (m < a.M) {
                        float* o = a.out + (size_t)m * a.ldo + ncol[nt];
                        float v = acc[t][nt][r] + b;
                        *o = a.accumulate ? (*o + v) : v;
                    }
                }
        } else {
            float* slab = a.out + (size_t)z * a.M * a.N;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int m = m0 + 16 * t + 4 * lq + r;
                    if (m < a.M) slab[(size_t)m * a.N + ncol[nt]] = acc[t][nt][r];
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// NT with split-precision operands ("x3"): C = X W^T for the skinny decoder-step shapes (M <= 64 rows per tile).
// gfx950 runs fp32-input MFMA at the vector rate (1/16 of bf16), and at 64 rows these GEMMs wait on the matrix pipe, not on
// memory.  Here every fp32 operand is cut into three bf16 pieces, x = x0 + x1 + x2 (each the bf16 rounding of what the
// previous ones left: 3 x 8 = 24 mantissa bits, the pieces are exact, the remainder is below 2^-24 |x|), and a product is
// the six piece products of order <= 2,
//     x w  ~=  x0 w0 + (x0 w1 + x1 w0) + (x1 w1 + x0 w2 + x2 w0),       dropped: x1 w2 + x2 w1 + x2 w2 <= 3 * 2^-24 |x w|,
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16: six bf16 MFMAs do the work of sixteen fp32 ones.  The weights stay fp32
// in memory (same bytes as before) and are split in registers on their way to the matrix pipe (VALU work in the shadow of
// the MFMAs); the activation tile is split once per stage when it is staged into LDS.  Operands must be finite.
// Used by the 128 x 128-tile kernels below (operand reuse: gemm_tn128_x3_kernel) and, in its own translation unit, by the
// resident-activation decoder-step kernel (gemm_resident_x3.hip).
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a, b) -> three packed bf16 pairs (a in the low half)
__device__ __forceinline__ void split3(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = cvt_pk_bf16(ra, rb);
}

// ------------------------------------------------------------------------------------------------
// NN: C = dY W   (A(m,k) = dY[m*lda+k], B(k,n) = W[k*ldb+n]).
// Mirror image of NT: the B chunk (64 k x 64 n, shared by the 4 waves) goes through LDS, each wave owns 16
// rows and all 64 columns; A fragments straight to registers.  Column tiles are interleaved: a lane's
// float4 along n at [k][4i..4i+3] supplies column 4i+j to column tile j, so one ds_read_b128 feeds 4 MFMAs.
template <bool TAIL, int BKC = GEMM_BK>
__global__ __launch_bounds__(256) void gemm_nn_kernel(GemmArgs a) {
    const int lflag = live_flag(a.live);
    constexpr int NS = BKC / 16;                 // k-groups of 16 per stage = A float4 per lane = B staging loads per thread
    __shared__ __attribute__((aligned(16))) float lds[2][BKC * GEMM_BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * GEMM_BN, m0 = blockIdx.y * GEMM_BM, z = blockIdx.z;
    int c_end;
    const int c_begin = split_range<BKC>(a, z, &c_end);
    const int mrow = m0 + wave * 16 + li;
    const size_t mrow_c = mrow < a.M ? mrow : a.M - 1;

    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 br[NS];
    f32x4 acur[NS], anxt[NS];
    // B staging: thread -> (row = idx>>4, n = n0 + 4*(idx&15)); columns beyond N are clamped (never stored)
    int bn = n0 + 4 * (tid & 15);
    if (bn > a.N - 4) bn = a.N - 4;

    ChunkCursorT<BKC> cc;
    auto load_chunk = [&](f32x4 (&av)[NS]) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = cc.k0 + 16 * s + 4 * lq;
            const int kc = k < cc.K ? k : cc.K - 4;
            av[s] = ld4z<TAIL>(cc.A + mrow_c * cc.lda + kc, k < cc.K);
        }
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            const int k = cc.k0 + (tid >> 4) + 16 * j;
            const int kc = k < cc.K ? k : cc.K - 1;
            br[j] = ld4z<TAIL>(cc.B + (size_t)kc * cc.ldb + bn, k < cc.K);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NS; ++j)
            *reinterpret_cast<f32x4*>(&lds[buf][((tid >> 4) + 16 * j) * GEMM_BN + 4 * (tid & 15)]) = br[j];
    };

    if (c_begin >= c_end && lflag == 0) return;
    if (c_begin < c_end) {
        cc.seek(a, c_begin);
        load_chunk(acur);
        if (flag_dead(lflag)) return;           // behind the first stage's loads, in front of the first write (icz_common.h)
        store_stage(0);
        __syncthreads();
        for (int c = c_begin; c < c_end; ++c) {
            const int buf = (c - c_begin) & 1;
            const bool more = (c + 1 < c_end);
            // Skinny shapes (one workgroup per CU, all in lockstep): the next stage's loads are spread over the stage instead
            // of issued in one burst (see gemm_nt_kernel): the weight tile's staging loads one per two MFMA groups over the
            // first half (their LDS store is due at the end of the stage), the A fragments one per four groups.
            const bool spread = !TAIL && BKC == 128 && a.M <= 64 && a.spread;
            if (more) {
                cc.next(a);
                if (!spread) load_chunk(anxt);
            }
            // keep the next stage's loads HERE (the compiler otherwise sinks them below the MFMAs, next to their first use)
            __builtin_amdgcn_sched_barrier(0);
            // B fragments: one ds_read_b128 per k (MFMA k index q <-> k = 16s + 4q + e, A side: component e of the lane's
            // float4) feeds 4 MFMAs.  The reads run two steps ahead of their MFMAs through a ring of three registers; the
            // fences pin that order (left alone, hipcc issues each read right before its use and every second group of
            // MFMAs waits out the LDS latency).
            constexpr int NJ = 4 * NS;
            const float* bbase = &lds[buf][(4 * lq) * GEMM_BN + 4 * li];
            f32x4 bf[3];
            bf[0] = *reinterpret_cast<const f32x4*>(bbase);
            bf[1] = *reinterpret_cast<const f32x4*>(bbase + GEMM_BN);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j + 2 < NJ) {
                    const int s2 = (j + 2) >> 2, e2 = (j + 2) & 3;
                    bf[(j + 2) % 3] = *reinterpret_cast<const f32x4*>(bbase + (16 * s2 + e2) * GEMM_BN);
                }
                if (spread && more) {
                    if ((j & 1) == 0 && (j >> 1) < NS) {
                        const int jj = j >> 1;
                        br[jj] = *reinterpret_cast<const f32x4*>(cc.B + (size_t)(cc.k0 + (tid >> 4) + 16 * jj) * cc.ldb + bn);
                    }
                    if ((j & 3) == 1) {
                        const int ss = j >> 2;
                        anxt[ss] = *reinterpret_cast<const f32x4*>(cc.A + mrow_c * cc.lda + cc.k0 + 16 * ss + 4 * lq);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(acur[j >> 2][j & 3], bf[j % 3][t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) store_stage(buf ^ 1);
            __syncthreads();
#pragma unroll
            for (int s = 0; s < NS; ++s) acur[s] = anxt[s];
        }
    }
    // epilogue: acc[t][r] <-> row m0 + 16*wave + 4q + r, column n0 + 4*i + t
    float* base = (a.nsplit == 1) ? a.out : a.out + (size_t)z * a.M * a.N;
    const int ldo = (a.nsplit == 1) ? a.ldo : a.N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int m = m0 + wave * 16 + 4 * lq + r;
        int n = n0 + 4 * li;
        if (m < a.M && n < a.N) {
            float* o = base + (size_t)m * ldo + n;
            f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
            if (a.nsplit == 1 && a.bias) {
                v[0] += a.bias[n]; v[1] += a.bias[n + 1]; v[2] += a.bias[n + 2]; v[3] += a.bias[n + 3];
            }
            if (a.nsplit == 1 && a.accumulate) {
                f32x4 old = *reinterpret_cast<f32x4*>(o);
                v += old;
            }
            *reinterpret_cast<f32x4*>(o) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TN: C = dY^T X  (A(m,k) = dY[k*lda+m], B(k,n) = X[k*ldb+n]); the reduction index k runs over rows of
// both operands (batch x time), so both are contiguous along their output index.  No LDS staging: each wave
// takes a quarter of the chunk's k range and the whole 64 x 64 tile (16 accumulators); a lane's float4 along
// m (resp. n) at row k supplies row 4i+j to row tile j (resp. column 4i+j to column tile j): 2 loads feed
// 16 MFMAs.  The four waves' partial tiles are summed through LDS at the end.
template <bool TAIL>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmArgs a) {
    if (step_dead(a.live)) return;
    __shared__ __attribute__((aligned(16))) float red[3][64 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * GEMM_BN, m0 = blockIdx.y * GEMM_BM, z = blockIdx.z;
    int c_end;
    const int c_begin = split_range(a, z, &c_end);
    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int mcol = m0 + 4 * li, ncol = n0 + 4 * li;
    if (mcol > a.M - 4) mcol = a.M - 4;      // clamped columns feed only never-stored outputs
    if (ncol > a.N - 4) ncol = a.N - 4;

    ChunkCursor cc;
    f32x4 av[4], bv[4], an[4], bn_[4];
    auto load_chunk = [&](f32x4 (&x)[4], f32x4 (&y)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = cc.k0 + 16 * wave + 4 * s + lq;
            const int kc = k < cc.K ? k : cc.K - 1;
            x[s] = ld4z<TAIL>(cc.A + (size_t)kc * cc.lda + mcol, k < cc.K);
            y[s] = ld4z<TAIL>(cc.B + (size_t)kc * cc.ldb + ncol, k < cc.K);
        }
    };
    if (c_begin < c_end) {
        cc.seek(a, c_begin);
        load_chunk(av, bv);
        for (int c = c_begin; c < c_end; ++c) {
            const bool more = (c + 1 < c_end);
            if (more) { cc.next(a); load_chunk(an, bn_); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], bv[s][u], acc[t][u], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s) { av[s] = an[s]; bv[s] = bn_[s]; }
        }
    }
    // acc[t][u][r] <-> row m0 + 4*(4q + r) + t, column n0 + 4*i + u.  Tile-local (row, col) in [0,64)^2.
    // waves 1..3 park their tiles in LDS, wave 0 sums in fixed order (bitwise reproducible).
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = 4 * (4 * lq + r) + t;
                f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
                *reinterpret_cast<f32x4*>(&red[wave - 1][row * 64 + 4 * li]) = v;
            }
    }
    __syncthreads();
    if (wave == 0) {
        float* base = (a.nsplit == 1) ? a.out : a.out + (size_t)z * a.M * a.N;
        const int ldo = (a.nsplit == 1) ? a.ldo : a.N;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int row = 4 * (4 * lq + r) + t;
                f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
#pragma unroll
                for (int w = 0; w < 3; ++w) v += *reinterpret_cast<const f32x4*>(&red[w][row * 64 + 4 * li]);
                int m = m0 + row, n = n0 + 4 * li;
                if (m < a.M && n < a.N) {
                    float* o = base + (size_t)m * ldo + n;
                    if (a.nsplit == 1 && a.accumulate) v += *reinterpret_cast<f32x4*>(o);
                    *reinterpret_cast<f32x4*>(o) = v;
                }
            }
    }
}

// ------------------------------------------------------------------------------------------------
// TN, large outputs (M, N >= 128; K a multiple of 32; no split): 128 x 128 tile per workgroup, both operand tiles
// ([32 k][128] floats each) staged through LDS and shared by the four waves, wave (wr, wc) owning the 64 x 64 quadrant.
// The 64 x 64-tile kernel above streams (64 + 64) x 4 bytes per k for 64 x 64 MACs -- 16 flop/B, i.e. ~38 GB/s per CU at
// full MFMA rate, more than a CU gets from beyond L2; this one needs half of that.  Fragments use the same interleave as
// above: the lane's float4 along m (n) at row k supplies row 4i+j (column 4i+j) of quadrant tile j, so one ds_read_b128
// per operand and k-step feeds 16 MFMAs.  Loads of the next stage are spread over the stage (one per k-step).
__global__ __launch_bounds__(256) void gemm_tn128_kernel(GemmArgs a) {
    if (step_dead(a.live)) return;
    constexpr int KC = 32, LD = 128 + 4;                       // row stride 132 dwords: the four k rows of a read hit disjoint banks
    __shared__ __attribute__((aligned(16))) float sA[2][KC * LD];
    __shared__ __attribute__((aligned(16))) float sB[2][KC * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int mbase = 64 * (wave >> 1), nbase = 64 * (wave & 1);
    const GemmSeg& g = a.seg[0];
    const int nst = g.K / KC;

    f32x4 acc[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging: thread -> 4 float4 of the A tile and 4 of the B tile per stage (row = idx >> 5, column = 4 * (idx & 31))
    const float* ap[4];
    const float* bp[4];
    int so[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int idx = tid + 256 * j, row = idx >> 5, c4 = 4 * (idx & 31);
        int mc = m0 + c4, nc = n0 + c4;
        if (mc > a.M - 4) mc = a.M - 4;                           // clamped columns feed only never-stored outputs
        if (nc > a.N - 4) nc = a.N - 4;
        ap[j] = g.A + (size_t)row * g.lda + mc;
        bp[j] = g.B + (size_t)row * g.ldb + nc;
        so[j] = row * LD + c4;
    }
    const size_t astep = (size_t)KC * g.lda, bstep = (size_t)KC * g.ldb;
    f32x4 ar[4], br[4];
    auto load_stage = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ar[j] = *reinterpret_cast<const f32x4*>(ap[j]); br[j] = *reinterpret_cast<const f32x4*>(bp[j]); }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<f32x4*>(&sA[buf][so[j]]) = ar[j];
            *reinterpret_cast<f32x4*>(&sB[buf][so[j]]) = br[j];
        }
    };
    load_stage();
    store_stage(0);
    __syncthreads();
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        const bool more = st + 1 < nst;
        if (more) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { ap[j] += astep; bp[j] += bstep; }
        }
        const float* pa = &sA[buf][lq * LD + mbase + 4 * li];
        const float* pb = &sB[buf][lq * LD + nbase + 4 * li];
        f32x4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const f32x4*>(pa);
        fb[0] = *reinterpret_cast<const f32x4*>(pb);
#pragma unroll
        for (int ks = 0; ks < KC / 4; ++ks) {
            if (ks + 1 < KC / 4) {
                fa[(ks + 1) & 1] = *reinterpret_cast<const f32x4*>(pa + 4 * (ks + 1) * LD);
                fb[(ks + 1) & 1] = *reinterpret_cast<const f32x4*>(pb + 4 * (ks + 1) * LD);
            }
            if (more) {      // the next stage's 8 loads, one per k-step
                if (ks < 4) ar[ks] = *reinterpret_cast<const f32x4*>(ap[ks]);
                else br[ks - 4] = *reinterpret_cast<const f32x4*>(bp[ks - 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[ks & 1][t], fb[ks & 1][u], acc[t][u], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
    }
    // acc[t][u][r] <-> row m0 + mbase + 4*(4q + r) + t, column n0 + nbase + 4*i + u
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + mbase + 4 * (4 * lq + r) + t, n = n0 + nbase + 4 * li;
            if (m < a.M && n < a.N) {
                f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
                float* o = a.out + (size_t)m * a.ldo + n;
                if (a.accumulate) v += *reinterpret_cast<f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = v;
            }
        }
}

// ------------------------------------------------------------------------------------------------
// TN, large outputs, split-precision operands (see gemm_nt_x3_kernel for the arithmetic): the weight-gradient GEMMs are
// matrix-pipe-bound with high operand reuse (128 x 128 tile: every staged element feeds 128 products), which is where six
// bf16 MFMAs instead of sixteen fp32 ones pay.  Both operands are k-major in memory (A(m,k) = dY[k][m], B(k,n) = X[k][n]);
// a thread loads eight consecutive k of one column with coalesced dword loads, splits them into three bf16 pieces and
// writes 16 bytes per piece to an [column][k] LDS image, from which the 32x32x16 fragments (8 consecutive k of one row /
// column) are plain 16-byte reads.  Wave (wr, wc) owns the 64 x 64 quadrant = 2 x 2 tiles of 32 x 32.
constexpr int T3_KC = 32, T3_PB = T3_KC + 8;                       // bf16 per LDS row: 80 bytes, 5 x 16 B (odd: conflict-free rows)
constexpr size_t T3_PLANE = (size_t)128 * T3_PB;                  // bf16 elements of one piece of one operand tile
constexpr size_t t3_lds(int nbuf) { return (size_t)nbuf * 2 * 3 * T3_PLANE * 2; }      // buffers x operands x pieces, bytes

// NBUF = 1: 60 KB of LDS, two workgroups per CU cover each other's barriers and waits (outputs of more than 256 tiles);
// NBUF = 2: double-buffered stages for the one-round shapes.  NW = 4: wave (wr, wc) owns a 64 x 64 quadrant; NW = 8: a
// 64 x 32 half quadrant (two waves per SIMD inside the workgroup).
// AROW = false: TN (A(m,k) = A[k lda + m]);  AROW = true: NN (A(m,k) = A[m lda + k], the batched dgrad GEMMs over all time
// steps): the A tile then comes in as two float4 per thread along k.  BROW = true as well: NT (B(k,n) = B[n ldb + k], every
// forward Linear with many rows: AoA refiner, beam-search steps), several K segments allowed when there is no split.
// blockIdx.z = split-K part (slabs [z][M][N] in a.out).
template <int NBUF, int NW, bool AROW = false, bool BROW = false>
__global__ __launch_bounds__(64 * NW) void gemm_tn128_x3_kernel(GemmArgs a) {
    if (step_dead(a.live)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char t3_smem[];
    unsigned short* const lds = reinterpret_cast<unsigned short*>(t3_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int rows_live = a.rows_live ? ((*a.rows_live + 31) & ~31) : 0x7fffffff;
    if (AROW && !BROW && m0 >= rows_live) return;            // NN: a row tile of steps the rollout never ran
    constexpr int NT = 64 * NW, NU = NW == 4 ? 2 : 1, IPT = 512 / NT;       // column tiles per wave; staging items per thread and operand
    const int mbase = NW == 4 ? 64 * (wave >> 1) : 64 * (wave >> 2), nbase = NW == 4 ? 64 * (wave & 1) : 32 * (wave & 3);
    f32x16 acc[2][NU];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][u][q] = 0.f;

    // one K segment (all of it, or this split's range when there is a single segment)
    auto run_segment = [&](const GemmSeg& g, int kbeg, int kend) {
        const int nst = (kend - kbeg) / T3_KC;
        if (nst <= 0) return;
        // staging: item = tid + NT j -> k block kb = item >> 7 (8 consecutive k), column c = item & 127  (k-major operands);
        // for a row-major A: row c = item >> 2, k block kb = item & 3 (four lanes read 128 contiguous bytes of a row)
        const float* ap[IPT];
        const float* bp[IPT];
        int so[IPT], soa[IPT];
#pragma unroll
        for (int j = 0; j < IPT; ++j) {
            const int item = tid + NT * j, kb = item >> 7, c = item & 127;
            int mc = m0 + c, nc = n0 + c;
            if (mc > a.M - 1) mc = a.M - 1;                            // clamped columns feed only never-stored outputs
            if (nc > a.N - 1) nc = a.N - 1;
            so[j] = c * T3_PB + 8 * kb;
            if (BROW) {
                const int rb = item >> 2, kq = item & 3;
                int nr = n0 + rb;
                if (nr > a.N - 1) nr = a.N - 1;
                bp[j] = g.B + (size_t)nr * g.ldb + kbeg + 8 * kq;
                so[j] = rb * T3_PB + 8 * kq;
            } else {
                bp[j] = g.B + (size_t)(kbeg + 8 * kb) * g.ldb + nc;
            }
            if (AROW) {
                const int ra = item >> 2, ka = item & 3;
                int mr = m0 + ra;
                if (mr > a.M - 1) mr = a.M - 1;
                ap[j] = g.A + (size_t)mr * g.lda + kbeg + 8 * ka;
                soa[j] = ra * T3_PB + 8 * ka;
            } else {
                ap[j] = g.A + (size_t)(kbeg + 8 * kb) * g.lda + mc;
                soa[j] = so[j];
            }
        }
        const size_t astep = AROW ? (size_t)T3_KC : (size_t)T3_KC * g.lda, bstep = BROW ? (size_t)T3_KC : (size_t)T3_KC * g.ldb;
        // register ring of two stages: iteration st multiplies LDS buffer st & 1, stores stage st + 1 (loaded one iteration
        // earlier) into the other buffer at its end and issues the loads of stage st + 2 at its top -- a stage is 48 MFMAs
        // (0.64 us), one stage of loads in flight does not cover the memory latency
        float ar0[IPT][8], br0[IPT][8], ar1[IPT][8], br1[IPT][8];
        auto load_stage = [&](float (&ar)[IPT][8], float (&br)[IPT][8]) {
#pragma unroll
            for (int j = 0; j < IPT; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) if (!BROW) br[j][e] = bp[j][(size_t)e * g.ldb];
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                if (BROW) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(bp[j]), hi = *reinterpret_cast<const f32x4*>(bp[j] + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { br[j][e] = lo[e]; br[j][4 + e] = hi[e]; }
                }
                if (AROW) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(ap[j]), hi = *reinterpret_cast<const f32x4*>(ap[j] + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ar[j][e] = lo[e]; ar[j][4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ar[j][e] = ap[j][(size_t)e * g.lda];
                }
            }
#pragma unroll
            for (int j = 0; j < IPT; ++j) { ap[j] += astep; bp[j] += bstep; }
        };
        auto put = [&](unsigned short* base, const float (&v)[8]) {
            uint32_t s0[4], s1[4], s2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(v[2 * e], v[2 * e + 1], s0[e], s1[e], s2[e]);
            *reinterpret_cast<u32x4*>(base) = (u32x4){s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<u32x4*>(base + T3_PLANE) = (u32x4){s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<u32x4*>(base + 2 * T3_PLANE) = (u32x4){s2[0], s2[1], s2[2], s2[3]};
        };
        auto store_stage = [&](int buf, const float (&ar)[IPT][8], const float (&br)[IPT][8]) {
            unsigned short* bA = lds + (size_t)buf * 6 * T3_PLANE;
            unsigned short* bB = bA + 3 * T3_PLANE;
#pragma unroll
            for (int j = 0; j < IPT; ++j) { put(bA + soa[j], ar[j]); put(bB + so[j], br[j]); }
        };
        auto compute = [&](int buf) {
            const unsigned short* pa = lds + (size_t)buf * 6 * T3_PLANE + (mbase + r) * T3_PB + 8 * h;
            const unsigned short* pb = lds + (size_t)buf * 6 * T3_PLANE + 3 * T3_PLANE + (nbase + r) * T3_PB + 8 * h;
#pragma unroll
            for (int blk = 0; blk < T3_KC / 16; ++blk) {
                bf16x8 af[3][2], bf[3][NU];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) af[p][i] = *reinterpret_cast<const bf16x8*>(pa + p * T3_PLANE + i * 32 * T3_PB + 16 * blk);
#pragma unroll
                    for (int u = 0; u < NU; ++u) bf[p][u] = *reinterpret_cast<const bf16x8*>(pb + p * T3_PLANE + u * 32 * T3_PB + 16 * blk);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int u = 0; u < NU; ++u) {       // smallest terms first
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2][i], bf[0][u], acc[i][u], 0, 0, 0);
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[2][u], acc[i][u], 0, 0, 0);
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[1][u], acc[i][u], 0, 0, 0);
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][i], bf[0][u], acc[i][u], 0, 0, 0);
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[1][u], acc[i][u], 0, 0, 0);
                        acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][i], bf[0][u], acc[i][u], 0, 0, 0);
                    }
            }
        };
        load_stage(ar0, br0);
        store_stage(0, ar0, br0);
        if (nst > 1) load_stage(ar1, br1);
        __syncthreads();
        // iteration st (even: registers 0 are free, 1 hold stage st + 1; odd: the other way round)
        auto iter = [&](int st, float (&arF)[IPT][8], float (&brF)[IPT][8], const float (&arN)[IPT][8], const float (&brN)[IPT][8]) {
            if (st + 2 < nst) load_stage(arF, brF);
            __builtin_amdgcn_sched_barrier(0);
            compute(NBUF == 2 ? (st & 1) : 0);
            if (NBUF == 1) __syncthreads();          // every wave has read this stage's fragments
            if (st + 1 < nst) store_stage(NBUF == 2 ? ((st + 1) & 1) : 0, arN, brN);
            __syncthreads();
        };
        int st = 0;
        for (; st + 2 <= nst; st += 2) {
            iter(st, ar0, br0, ar1, br1);
            iter(st + 1, ar1, br1, ar0, br0);
        }
        if (st < nst) iter(st, ar0, br0, ar1, br1);
    };
    if (a.nseg == 1) {
        // K range of this split: a.chunks_per_split counts 128-deep chunks (gemm_f32), a stage here is 32 deep
        const int kbeg = a.nsplit > 1 ? (int)blockIdx.z * a.chunks_per_split * 128 : 0;
        int kend = a.nsplit > 1 ? min(a.seg[0].K, kbeg + a.chunks_per_split * 128) : a.seg[0].K;
        if (!AROW && !BROW) kend = min(kend, rows_live);     // TN: the sum over (t, b) stops behind the last step that ran
        run_segment(a.seg[0], kbeg, kend);
    } else {
        // several K segments: the split's range [gbeg, gend) of the concatenated K is cut at the segment borders
        const int gbeg = a.nsplit > 1 ? (int)blockIdx.z * a.chunks_per_split * 128 : 0;
        const int gend = a.nsplit > 1 ? gbeg + a.chunks_per_split * 128 : 0x7fffffff;
        int off = 0;
#pragma unroll 1
        for (int sgi = 0; sgi < a.nseg; ++sgi) {
            const int K = a.seg[sgi].K;
            const int kb = max(gbeg - off, 0), ke = min(gend - off, K);
            if (kb < ke) run_segment(a.seg[sgi], kb, ke);
            off += K;
        }
    }
    // acc[i][u][q] <-> row m0 + mbase + 32 i + (q & 3) + 8 (q >> 2) + 4 h, column n0 + nbase + 32 u + r
    const bool direct = a.nsplit == 1;
    float* const outp = direct ? a.out : a.out + (size_t)blockIdx.z * a.M * a.N;
    const int ldo = direct ? a.ldo : a.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = n0 + nbase + 32 * u + r;
            if (n >= a.N) continue;
            const float bias_n = (direct && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + mbase + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m < a.M) {
                    float* o = outp + (size_t)m * ldo + n;
                    const float v = acc[i][u][q] + bias_n;
                    *o = (direct && a.accumulate) ? (*o + v) : v;
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// Optional timing of the decoder-step forward GEMMs (33..64 rows; icz_prof_select) with HIP events on the launch stream (bench.py's live
// roofline measurement).  Off by default; nothing is recorded or allocated unless enabled.
struct GemmProf {
    bool on = false;
    std::vector<hipEvent_t> ev;      // pairs
    size_t used = 0;
    double bytes = 0.0, flops = 0.0;
    unsigned seen = 0, every = 1;     // bracket every `every`-th launch (ICZ_PROF_EVERY): fewer event packets in the stream
    int select = 0;                   // 0: every skinny forward GEMM (NT, 33..64 rows); 1: the resident-activation kernel only
};
static GemmProf g_prof;
static thread_local bool g_capturing = false;
void gemm_set_capturing(bool on) { g_capturing = on; }

// ---- event pairs around groups of small kernels (icz_kprof_*, include/icz.h)
struct KProf { bool on = false; std::vector<hipEvent_t> ev[KP_GROUPS]; size_t used[KP_GROUPS] = {}; };
static KProf g_kprof;
void kprof_mark(int group, bool begin, hipStream_t st) {
    if (!g_kprof.on || g_capturing || group < 0 || group >= KP_GROUPS) return;
    std::vector<hipEvent_t>& ev = g_kprof.ev[group];
    size_t& u = g_kprof.used[group];
    if (begin) {
        if (u + 2 > ev.size()) {
            if (ev.size() >= 8192) return;
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            ev.push_back(a); ev.push_back(b);
        }
        (void)hipEventRecord(ev[u], st);
    } else if (u + 2 <= ev.size()) {
        (void)hipEventRecord(ev[u + 1], st);
        u += 2;
    }
}

static void prof_account(const GemmArgs& a) {
    double kb = 0.0, k = 0.0;
    for (int s = 0; s < a.nseg; ++s) { k += a.seg[s].K; }
    kb = ((double)a.M * k + (double)a.N * k + (double)a.M * a.N) * 4.0;   // read A once, read W once, write C once
    g_prof.bytes += kb;
    g_prof.flops += 2.0 * a.M * a.N * k;
}

size_t gemm_slab_floats(int M, int N, int nsplit) { return nsplit > 1 ? (size_t)nsplit * M * N : 0; }

// Library switches: read from the environment ONCE, at first use.  They exist for bench.py's fp32-MFMA leg and the slab A/B test.
const GemmSwitches& gemm_switches() {
    static const GemmSwitches sw = [] {
        auto on = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) != 0 : true; };
        GemmSwitches s;
        s.tn_x3 = on("ICZ_GEMM_TN_X3"); s.nn_x3 = on("ICZ_GEMM_NN_X3"); s.nt_x3big = on("ICZ_GEMM_NT_X3BIG");
        s.resident_x3 = on("ICZ_GEMM_RESIDENT_X3"); s.resident_m128 = on("ICZ_GEMM_RESIDENT_M128"); s.predict_slabs = on("ICZ_PREDICT_SLABS");
        s.resident_k512 = on("ICZ_GEMM_RESIDENT_K512");
        const char* e = getenv("ICZ_PROF_EVERY");
        s.prof_every = e && atoi(e) > 1 ? (unsigned)atoi(e) : 1u;
        const char* b = getenv("ICZ_GEMM_BIG");
        s.big_cfg = b ? atoi(b) : -1;
        return s;
    }();
    return sw;
}

// runtime override of ICZ_GEMM_BIG (tests sweep the tile configurations in one process); -2 = none
static int g_big_override = -2;
void gemm_set_big_cfg(int cfg) { g_big_override = cfg; }
int gemm_big_switch() { return g_big_override != -2 ? g_big_override : gemm_switches().big_cfg; }

// NT pipeline-stage depth: 128 when every segment's K is a multiple of 128 and the tile is full height (MT = 4)
static int nt_stage_k(const GemmArgs& a) {
    if (gemm_resident_x3_fits(a)) return 64;
    if (a.M <= 32) return 64;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % 128) return 64;
    return 128;
}
// NN pipeline-stage depth: 128 when every segment's K is a multiple of 128 (halves the barriers and gives the next
// stage's loads 128 MFMAs per wave to land behind)
static int nn_stage_k(const GemmArgs& a) {
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % 128) return 64;
    return 128;
}
// NN shapes that go to the split-precision 128 x 128-tile kernel: one segment, at least 128 rows and columns, K in whole
// 128-deep chunks, plain output (ICZ_GEMM_NN_X3=0: the fp32-MFMA kernel)
static bool nn_x3(const GemmArgs& a) {
    return gemm_switches().nn_x3 && a.nseg == 1 && a.M >= 128 && a.N >= 128 && a.seg[0].K % 128 == 0 && !a.bias && a.N % 4 == 0;
}
static int stage_k(GemmLayout layout, const GemmArgs& a) {
    return layout == GEMM_NT ? nt_stage_k(a) : (layout == GEMM_NN ? nn_stage_k(a) : GEMM_BK);
}

// spread-load variant of the fp32 NT kernel (see the kernel body).  Measured: -9.5 % at M = 64, -3 % at M = 2304 (refiner /
// prologue GEMMs), +4 % at M = 320 (beam rows)
static bool nt_spread(const GemmArgs& a) { return a.M <= 64 || a.M >= 1024; }
// NT shapes with many rows (AoA refiner, beam-search steps, prologue hoists) that go to the split-precision 128 x 128-tile
// kernel: whole 128-deep chunks, enough tiles to fill at least half of the CUs (one K segment may also be split)
static bool nt_x3big(const GemmArgs& a) {
    if (!gemm_switches().nt_x3big || a.M < 128 || a.N < 128 || gemm_resident_x3_fits(a)) return false;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % 128) return false;
    const int tiles = cdiv(a.N, 128) * cdiv(a.M, 128);
    if (a.nseg == 1) {
        if (tiles * (a.seg[0].K / 512 > 0 ? a.seg[0].K / 512 : 1) >= 128) return true;
        // few tiles but many rows (dec_att of a beam step: 640 x 1024 x 1024 = 40 tiles): 256-deep splits still fill most of the chip,
        // and the fp32 kernel needs 25 us for it (round 2's beam profile).  Round 4: from 64 such workgroups on (the query
        // projection of an AoA beam step at 320 rows: 24 tiles x 4 splits)
        return a.M >= 256 && tiles * (a.seg[0].K / 256) >= 64;
    }
    int tot = 0;                                   // several segments: their concatenation is split as well (gemm_pick_split)
    for (int s = 0; s < a.nseg; ++s) tot += a.seg[s].K / 128;
    // round 4: with few tiles (the AoA linear of a beam step, 320 x 2048 over [x_att | q]: 48 tiles) 512-deep splits count as well --
    // the fp32 kernel took 28 us per launch for it (profiles/r04_aoa_beam5_b64_kernel_stats.csv: 15 % of AoA beam search)
    const int per = (a.M >= 256 && tiles < 128) ? 4 : 8;
    return tiles * (tot / per > 0 ? tot / per : 1) >= 128;
}
static int nt_tile_n(const GemmArgs& a) { return 64; }      // fp32 NT kernel: four waves x one 16-column tile

int gemm_pick_split(const GemmArgs& a, int target_wgs, GemmLayout layout) {
    if (layout == GEMM_NT && gemm_resident_x3_fits(a)) return gemm_resident_x3_nsplit(a);      // fixed: 256-deep k ranges
    if (layout == GEMM_NT && nt_x3big(a)) {
        // 128 x 128 tiles, two workgroups per CU.  Measured at 2304 rows (tools/perf_gemm_nt_big.py, us for split 1 / 2 / 3 / 4):
        // N 1024 K 1024: 48 / 55 / 50 / 63;  N 1024 K 2048: 86 / 89 / 77 / 96;  N 2048 K 2048: 165 / 164 / 148 / 153;
        // N 3072 K 1024: 92 / 115 / 124 / 135  ->  up to 1.75 rounds of workgroups, at least 5 chunks of 128 per split
        const int tiles = cdiv(a.N, 128) * cdiv(a.M, 128);
        if (a.nseg > 1) {
            // several K segments (the LSTM gates of a beam-search step: 160 tiles at 640 rows, each pulling 4 MB through its
            // compute unit, 96 of the 256 CUs idle): one round of at most 512 workgroups, at least 8 chunks of 128 per split
            int tot = 0;
            for (int sg = 0; sg < a.nseg; ++sg) tot += a.seg[sg].K / 128;
            int s = 512 / (tiles > 0 ? tiles : 1);
            const int per = (a.M >= 256 && tiles < 128) ? 4 : 8;      // chunks of 128 per split at least (see nt_x3big)
            if (s > tot / per) s = tot / per;
            if (s < 1) s = 1;
            return cdiv(tot, cdiv(tot, s));
        }
        const int tot = a.seg[0].K / 128;
        int s = 896 / tiles;
        if (s > tot / 5) s = tot / 5;
        if (tiles < 128 && tot >= 4) s = (tot / 2 < 320 / tiles) ? tot / 2 : 320 / tiles;      // few tiles: splits of two chunks and more
        if (s < 1) s = 1;
        return cdiv(tot, cdiv(tot, s));
    }
    int tiles = cdiv(a.N, layout == GEMM_NT ? nt_tile_n(a) : GEMM_BN) * cdiv(a.M, GEMM_BM);
    int tot = total_chunks(a, stage_k(layout, a));
    int s = target_wgs / (tiles > 0 ? tiles : 1);
    if (s < 1) s = 1;
    if (s > tot) s = tot;
    if (s > 32) s = 32;
    // make every split non-empty
    int cps = cdiv(tot, s);
    return cdiv(tot, cps);
}

// Split-K factor for GEMMs with many output tiles (the batched dgrad GEMMs over all time steps): the launch runs in
// ceil(tiles * s / 256) rounds of one workgroup per CU, each round costing its K chunks plus a fixed prologue/epilogue
// (about 6 us = 2.4 stages of 128); s is chosen to minimise rounds * (chunks per split + fixed) under the slab capacity.
// E.g. 320 tiles x 79 chunks: s = 1 needs 2 rounds of 79 (the second a quarter full), s = 4 needs 5 full rounds of 20.
int gemm_pick_split_balanced(const GemmArgs& a, GemmLayout layout, size_t slab_capacity_floats) {
    if (layout == GEMM_NN && nn_x3(a)) {      // 128 x 128 tiles: about two workgroups per CU, at least 8 chunks of 128 per split
        const int tiles = cdiv(a.N, 128) * cdiv(a.M, 128), tot = a.seg[0].K / 128;
        GemmArgs probe = a;
        probe.nsplit = 1;
        int s = (gemm_big_cfg(GEMM_NN, probe) == 4 ? 768 : 512) / (tiles > 0 ? tiles : 1);      // three workgroups per CU there
        if (s > tot / 8) s = tot / 8;
        if (s < 1) s = 1;
        while (s > 1 && (size_t)s * a.M * a.N > slab_capacity_floats) --s;
        return cdiv(tot, cdiv(tot, s));
    }
    const int bk = stage_k(layout, a);
    const int tiles = cdiv(a.N, layout == GEMM_NT ? nt_tile_n(a) : GEMM_BN) * cdiv(a.M, GEMM_BM);
    const int tot = total_chunks(a, bk);
    const double fixed = 2.4 * 128.0 / bk;
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 32 && s <= tot; ++s) {
        const int cps = cdiv(tot, s);
        if (cdiv(tot, cps) != s) continue;
        if (s > 1 && (size_t)s * a.M * a.N > slab_capacity_floats) break;
        const int rounds = cdiv(tiles * s, 256);
        const double cost = rounds * (cps + fixed) + (s > 1 ? 0.25 * s : 0.0);
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

// the largest split <= nsplit whose slabs fit `capacity_floats` (no empty splits)
int gemm_fit_split(GemmLayout layout, const GemmArgs& a, int nsplit, size_t capacity_floats) {
    nsplit = gemm_normalize_split(layout, a, nsplit);
    while (nsplit > 1 && gemm_slab_floats(a.M, a.N, nsplit) > capacity_floats) nsplit = gemm_normalize_split(layout, a, nsplit - 1);
    return nsplit;
}
int gemm_normalize_split(GemmLayout layout, const GemmArgs& a, int nsplit) {
    const int tot = total_chunks(a, stage_k(layout, a));
    if (nsplit < 1) nsplit = 1;
    if (nsplit > tot) nsplit = tot;
    return cdiv(tot, cdiv(tot, nsplit));
}

static int check_args(GemmLayout layout, const GemmArgs& a) {
    ICZ_REQUIRE(a.nseg >= 1 && a.nseg <= GEMM_MAX_SEG, "gemm: nseg %d out of range", a.nseg);
    ICZ_REQUIRE(a.M > 0 && a.N > 0 && a.out, "gemm: bad M/N/out");
    ICZ_REQUIRE(a.nsplit >= 1, "gemm: nsplit %d", a.nsplit);
    ICZ_REQUIRE(a.nsplit == 1 || (!a.bias && !a.accumulate), "gemm: bias/accumulate need nsplit == 1");
    for (int s = 0; s < a.nseg; ++s) {
        const GemmSeg& g = a.seg[s];
        ICZ_REQUIRE(g.A && g.B && g.K > 0, "gemm: segment %d null operand or K<=0", s);
        ICZ_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm: segment %d operands must be 16-byte aligned", s);
        ICZ_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: segment %d leading dims must be multiples of 4 (lda %d ldb %d)", s, g.lda, g.ldb);
        ICZ_REQUIRE(!g.gather, "gemm: row gather is not supported");
        if (layout == GEMM_NT) {
            ICZ_REQUIRE(g.K % 4 == 0, "gemm NT: segment %d K=%d must be a multiple of 4", s, g.K);
        } else if (layout == GEMM_NN) {
            ICZ_REQUIRE(g.K % 4 == 0 && a.N % 4 == 0, "gemm NN: K and N must be multiples of 4 (K %d N %d)", g.K, a.N);
            ICZ_REQUIRE(!g.gather, "gemm NN: gather unsupported");
        } else {
            ICZ_REQUIRE(a.M % 4 == 0 && a.N % 4 == 0, "gemm TN: M and N must be multiples of 4 (M %d N %d)", a.M, a.N);
            ICZ_REQUIRE(!g.gather, "gemm TN: gather unsupported");
        }
    }
    if (layout != GEMM_NT && a.nsplit == 1) {
        ICZ_REQUIRE(a.ldo % 4 == 0 && ((uintptr_t)a.out & 15) == 0, "gemm: output must be float4-aligned");
    }
    return ICZ_OK;
}

// ICZ_GEMM_LOG=1: every distinct shape that reaches the 128 x 128 / large-tile split-precision kernels, once, on stderr (tuning aid)
static void log_big_shape(GemmLayout layout, const GemmArgs& a) {
    static const bool on = [] { const char* e = getenv("ICZ_GEMM_LOG"); return e && atoi(e) > 0; }();
    if (!on) return;
    static std::mutex mu;
    static std::set<std::array<int, 6>> seen;
    int K = 0;
    for (int s = 0; s < a.nseg; ++s) K += a.seg[s].K;
    std::lock_guard<std::mutex> lk(mu);
    if (seen.insert({(int)layout, a.M, a.N, K, a.nseg, a.nsplit}).second)
        fprintf(stderr, "[icz gemm] %s M=%d N=%d K=%d nseg=%d nsplit=%d rows_live=%d\n", layout == GEMM_NT ? "nt" : (layout == GEMM_NN ? "nn" : "tn"), a.M, a.N, K, a.nseg, a.nsplit, a.rows_live ? 1 : 0);
}

int gemm_f32(GemmLayout layout, const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    ICZ_TRY(check_args(layout, a));
    int tot = total_chunks(a, stage_k(layout, a));
    a.chunks_per_split = cdiv(tot, a.nsplit);
    ICZ_REQUIRE(cdiv(tot, a.chunks_per_split) == a.nsplit, "gemm: nsplit %d leaves empty splits (chunks %d)", a.nsplit, tot);
    dim3 block(256);
    bool tail = false;
    for (int s = 0; s < a.nseg; ++s) tail |= (a.seg[s].K % GEMM_BK) != 0;
    if (layout == GEMM_NT) {
        if (nt_x3big(a)) {
            static bool attr = false;
            if (!attr) {
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<1, 4, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(1)));
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<2, 8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(2)));
                attr = true;
            }
            // chunks_per_split was computed in NT stage units (128 here, the predicate guarantees it) -- the kernel's unit too
            const dim3 g128(cdiv(a.N, 128), cdiv(a.M, 128), a.nsplit);
            log_big_shape(layout, a);
            if (const int bc = gemm_big_cfg(layout, a)) return gemm_big_x3(layout, a, bc, stream);
            if ((int)(g128.x * g128.y * g128.z) > 256) hipLaunchKernelGGL((gemm_tn128_x3_kernel<1, 4, true, true>), g128, block, t3_lds(1), stream, a);
            else hipLaunchKernelGGL((gemm_tn128_x3_kernel<2, 8, true, true>), g128, dim3(512), t3_lds(2), stream, a);
            ICZ_CHECK_HIP(hipGetLastError());
            return ICZ_OK;
        }
        int mt = a.M <= 16 ? 1 : (a.M <= 32 ? 2 : 4);
        const int bn = nt_tile_n(a);
        dim3 grid(cdiv(a.N, bn), cdiv(a.M, mt * 16), a.nsplit);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        const bool resident = gemm_resident_x3_fits(a) && a.nsplit == gemm_resident_x3_nsplit(a) && a.chunks_per_split == gemm_resident_x3_stages(a);
        // inside a stream capture the two records become event nodes of the graph: every replay refreshes them
        if (g_prof.on && (g_prof.select == 1 ? resident : (mt == 4 || resident)) && (g_prof.seen++ % g_prof.every) == 0) {
            if (g_prof.used + 2 <= g_prof.ev.size()) {
                e0 = g_prof.ev[g_prof.used]; e1 = g_prof.ev[g_prof.used + 1];
                g_prof.used += 2;
                prof_account(a);
                (void)hipEventRecord(e0, stream);
            }
        }
#define ICZ_NT(MT_, NTW_, BK_) do { if (tail) hipLaunchKernelGGL((gemm_nt_kernel<MT_, NTW_, true, BK_>), grid, block, 0, stream, a); \
                                    else hipLaunchKernelGGL((gemm_nt_kernel<MT_, NTW_, false, BK_>), grid, block, 0, stream, a); } while (0)
        if (resident) {
            const int st = gemm_resident_x3(a, stream);
            if (e1) (void)hipEventRecord(e1, stream);
            return st;
        }
        if (mt == 1) ICZ_NT(1, 1, 64);
        else if (mt == 2) ICZ_NT(2, 1, 64);
        else if (nt_stage_k(a) == 128 && nt_spread(a)) hipLaunchKernelGGL((gemm_nt_kernel<4, 1, false, 128, 4, true>), grid, block, 0, stream, a);
        else if (nt_stage_k(a) == 128) ICZ_NT(4, 1, 128);
        else ICZ_NT(4, 1, 64);
#undef ICZ_NT
        if (e1) (void)hipEventRecord(e1, stream);
    } else if (layout == GEMM_NN) {
        dim3 grid(cdiv(a.N, GEMM_BN), cdiv(a.M, GEMM_BM), a.nsplit);
        a.spread = 1;
        if (nn_x3(a)) {      // batched dgrad GEMMs (all time steps at once): split-precision 128 x 128 tiles
            static bool attr = false;
            if (!attr) {
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<1, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(1)));
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<2, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(2)));
                attr = true;
            }
            const dim3 g128(cdiv(a.N, 128), cdiv(a.M, 128), a.nsplit);
            log_big_shape(layout, a);
            if (const int bc = gemm_big_cfg(layout, a)) return gemm_big_x3(layout, a, bc, stream);
            if ((int)(g128.x * g128.y * g128.z) > 256) hipLaunchKernelGGL((gemm_tn128_x3_kernel<1, 4, true>), g128, block, t3_lds(1), stream, a);
            else hipLaunchKernelGGL((gemm_tn128_x3_kernel<2, 8, true>), g128, dim3(512), t3_lds(2), stream, a);
        }
        else if (nn_stage_k(a) == 128) hipLaunchKernelGGL((gemm_nn_kernel<false, 128>), grid, block, 0, stream, a);
        else if (tail) hipLaunchKernelGGL(gemm_nn_kernel<true>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(gemm_nn_kernel<false>, grid, block, 0, stream, a);
    } else {
        // at least one 128 x 128 tile per CU, else the 64 x 64 kernel (4x the workgroups) fills the chip better
        if (a.nsplit == 1 && a.nseg == 1 && cdiv(a.M, 128) * cdiv(a.N, 128) >= 256 && a.seg[0].K % 32 == 0 && a.seg[0].K >= 64) {
            const bool x3 = gemm_switches().tn_x3;        // off: the fp32-MFMA kernel
            static bool attr = false;
            if (x3 && !attr) {
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(1)));
                ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn128_x3_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)t3_lds(2)));
                attr = true;
            }
            const dim3 g128(cdiv(a.N, 128), cdiv(a.M, 128), 1);
            const int tiles = (int)(g128.x * g128.y);
            log_big_shape(layout, a);
            if (const int bc = x3 ? gemm_big_cfg(layout, a) : 0) return gemm_big_x3(layout, a, bc, stream);
            if (!x3) hipLaunchKernelGGL(gemm_tn128_kernel, g128, block, 0, stream, a);
            // measured (4096 x {1024, 3072, 4096} x 1280, 10112 x 1024 x 1280): more than one round of tiles -> one LDS buffer and
            // two workgroups per CU (251 us / 171 TFLOP/s-equivalent at 4096 x 4096 against 369 us for the fp32 kernel); one
            // round -> two buffers and eight waves (73 against 100 us at 4096 x 1024)
            else if (tiles > 256) hipLaunchKernelGGL((gemm_tn128_x3_kernel<1, 4>), g128, block, t3_lds(1), stream, a);
            else hipLaunchKernelGGL((gemm_tn128_x3_kernel<2, 8>), g128, dim3(512), t3_lds(2), stream, a);
            ICZ_CHECK_HIP(hipGetLastError());
            return ICZ_OK;
        }
        dim3 grid(cdiv(a.N, GEMM_BN), cdiv(a.M, GEMM_BM), a.nsplit);
        if (tail) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, block, 0, stream, a);
        else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, block, 0, stream, a);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

bool gemm_prof_on() { return g_prof.on; }
void gemm_prof_select(int which) { g_prof.select = which; }
void gemm_prof_begin() {
    g_prof.on = true; g_prof.used = 0; g_prof.bytes = 0.0; g_prof.flops = 0.0; g_prof.seen = 0;
    // the whole pool is created here: events cannot be created while a stream capture is in progress
    while (g_prof.ev.size() < 8192) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; g_prof.ev.push_back(e); }
    g_prof.every = gemm_switches().prof_every;
}
int gemm_prof_end(double* avg_us, double* bytes_per_launch, double* flops_per_launch, long long* launches) {
    g_prof.on = false;
    const size_t n = g_prof.used / 2;
    double tot_ms = 0.0;
    for (size_t i = 0; i < n; ++i) {
        ICZ_CHECK_HIP(hipEventSynchronize(g_prof.ev[2 * i + 1]));
        float ms = 0.f;
        ICZ_CHECK_HIP(hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]));
        tot_ms += ms;
    }
    if (avg_us) *avg_us = n ? tot_ms * 1e3 / (double)n : 0.0;
    if (bytes_per_launch) *bytes_per_launch = n ? g_prof.bytes / (double)n : 0.0;
    if (flops_per_launch) *flops_per_launch = n ? g_prof.flops / (double)n : 0.0;
    if (launches) *launches = (long long)n;
    return ICZ_OK;
}

__global__ void prof_null_kernel(int* p) { if (p) *p = 0; }

// What an event pair costs by itself: the average pair time around an empty one-thread kernel on `stream`, launched after
// another kernel like the timed GEMMs are (the gap between the two records holds the dispatch of the bracketed kernel).
int gemm_prof_pair_overhead(hipStream_t stream, int n, double* avg_us) {
    if (n < 1) n = 1;
    if (n > 256) n = 256;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto& e : ev) ICZ_CHECK_HIP(hipEventCreate(&e));
    for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(prof_null_kernel, dim3(256), dim3(256), 0, stream, (int*)nullptr);     // predecessor on the stream
        ICZ_CHECK_HIP(hipEventRecord(ev[2 * i], stream));
        hipLaunchKernelGGL(prof_null_kernel, dim3(1), dim3(1), 0, stream, (int*)nullptr);
        ICZ_CHECK_HIP(hipEventRecord(ev[2 * i + 1], stream));
    }
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        ICZ_CHECK_HIP(hipEventSynchronize(ev[2 * i + 1]));
        float ms = 0.f;
        ICZ_CHECK_HIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        tot += ms;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (avg_us) *avg_us = tot * 1e3 / n;
    return ICZ_OK;
}

// read-only stream over a buffer: every workgroup walks chunks of 4096 float4 (64 KB), a lane keeps sixteen 16-byte loads in flight
__global__ __launch_bounds__(256) void stream_rate_kernel(const f32x4* __restrict__ p, size_t nchunks, float* __restrict__ sink) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const f32x4* q = p + c * 4096 + threadIdx.x;
        f32x4 v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = __builtin_nontemporal_load(q + i * 256);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) sink[0] = acc.x;       // never true for real data: keeps the loads alive
}

}  // namespace icz

extern "C" {
int icz_prof_stream_rate(const void* buf, size_t bytes, int32_t reps, void* stream, double* gbs) {
    ICZ_REQUIRE(buf && gbs && reps > 0 && bytes >= (size_t)65536 * 1024 && bytes % 65536 == 0, "icz_prof_stream_rate: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* sink = nullptr;
    ICZ_CHECK_HIP(hipMalloc((void**)&sink, 16));
    hipEvent_t e0, e1;
    ICZ_CHECK_HIP(hipEventCreate(&e0));
    ICZ_CHECK_HIP(hipEventCreate(&e1));
    const size_t nchunks = bytes / 65536;
    const int grid = 256 * 8;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(icz::stream_rate_kernel, dim3(grid), dim3(256), 0, st, (const icz::f32x4*)buf, nchunks, sink);
    ICZ_CHECK_HIP(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(icz::stream_rate_kernel, dim3(grid), dim3(256), 0, st, (const icz::f32x4*)buf, nchunks, sink);
    ICZ_CHECK_HIP(hipEventRecord(e1, st));
    ICZ_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    ICZ_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
    *gbs = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return ICZ_OK;
}
int icz_prof_pair_overhead(void* stream, int32_t n, double* avg_us) { return icz::gemm_prof_pair_overhead((hipStream_t)stream, n, avg_us); }
int icz_prof_begin(void) { icz::gemm_prof_begin(); return ICZ_OK; }
int icz_kprof_begin(void) {
    icz::g_kprof.on = true;
    for (size_t& u : icz::g_kprof.used) u = 0;
    return ICZ_OK;
}
int icz_kprof_end(int32_t group, double* avg_us, long long* pairs) {
    ICZ_REQUIRE(group >= 0 && group < icz::KP_GROUPS && avg_us && pairs, "icz_kprof_end: bad arguments");
    icz::g_kprof.on = false;
    ICZ_CHECK_HIP(hipDeviceSynchronize());
    const size_t n = icz::g_kprof.used[group] / 2;
    double tot = 0.0;
    for (size_t i = 0; i < n; ++i) {
        float ms = 0.f;
        ICZ_CHECK_HIP(hipEventElapsedTime(&ms, icz::g_kprof.ev[group][2 * i], icz::g_kprof.ev[group][2 * i + 1]));
        tot += ms;
    }
    *avg_us = n ? tot * 1e3 / (double)n : 0.0;
    *pairs = (long long)n;
    return ICZ_OK;
}
int icz_prof_select(int32_t which) {
    if (which < 0 || which > 1) { icz::set_error("icz_prof_select: %d (0 = all skinny forward GEMMs, 1 = resident-activation kernel)", which); return ICZ_ERR_INVALID; }
    icz::gemm_prof_select(which);
    return ICZ_OK;
}
int icz_prof_end(double* avg_us, double* bytes_per_launch, double* flops_per_launch, long long* launches) {
    return icz::gemm_prof_end(avg_us, bytes_per_launch, flops_per_launch, launches);
}
}
