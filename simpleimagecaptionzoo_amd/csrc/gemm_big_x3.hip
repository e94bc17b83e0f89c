// Split-precision GEMM for the many-row products of the path, large tiles: weight gradients dW = dy^T x over all (t, b), the batched dgrad
// products over all time steps, and every forward Linear with >= 128 rows (beam-search steps, AoA refiner, XE vocabulary projection).
//
// gemm_tn128_x3_kernel (gemm_f32.hip) is a 128 x 128 tile with two barriers per 32-deep stage; it pulls 32 KB through the compute unit
// for every 48 MFMAs of a wave and reaches 0.39 - 0.44 of the split-precision matrix peak.  Round 3 measured what
// a deeper structure is worth on PRE-SPLIT operands (tools/cxx/gemm_planes_experiment.hip: 256 x 256 tile, eight waves, one barrier
// per 16-deep k-step: 0.50) and lost the gain to the packing pass.  This kernel keeps that structure and the fp32 operands:
//
//   * BM x BN tile, WM x WN waves, wave tile (BM / WM) x (BN / WN) in 32 x 32 MFMA blocks (v_mfma_f32_32x32x16_bf16, six per product);
//   * a k-step is 16 deep.  Its LDS image is the planes experiment's: per operand, piece and 32-row block one KiB, row r / k half h
//     at chunk 2 r + (h ^ bx_sw(r)) -- fragment reads (ds_read_b128, lane = (row, half)) and both staging maps below are
//     conflict-free on it;
//   * two LDS slots, ONE barrier per k-step: during step s (MFMAs on slot s & 1) a thread cuts its 8 + 8 fp32 values of step s + 1
//     into three bf16 pieces and writes them to the other slot (free since the barrier that ended step s - 1), while its loads
//     of step s + 2 are in flight into the second half of a two-stage register ring;
//   * staging map, K-contiguous operand (NT: both, NN: A): thread -> (row = item >> 1, half = item & 1), two 16-byte loads;
//     k-major operand (TN: both, NN: B): thread -> (row = item % R, half = item / R), eight dword loads, coalesced along the rows;
//   * workgroups are numbered so that the tiles that share operand blocks run on one XCD (one L2).
//
// Arithmetic = gemm_tn128_x3_kernel's, in the same order per accumulator (k blocks of 16 ascending, piece products a2 b0, a0 b2, a1 b1,
// a1 b0, a0 b1, a0 b0): for the same split-K decomposition the results are bitwise those of the 128 x 128 kernel.
#include "gemm_f32.h"

namespace icz {
namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bx_bf16x8;
typedef __attribute__((ext_vector_type(16))) float bx_f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t bx_u32x4;

__device__ __forceinline__ uint32_t bx_cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void bx_split3(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = bx_cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = bx_cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = bx_cvt_pk_bf16(ra, rb);
}

constexpr int BX_KS = 16;          // k-step depth

// Which of its two 16-byte chunks (2 r, 2 r + 1) half h of row r takes inside the KiB of a 32-row block: h ^ bx_sw(r).  Three access
// patterns must be conflict-free (MI355X_MICROARCH.md, LDS): the fragment read (ds_read_b128: lane groups {0-3, 12-15, 20-27} and
// {4-11, 16-19, 28-31} of each half-wave, 64 banks: the two rows of a group with the same r mod 8 must differ in the bit), the
// K-contiguous staging store (ds_write_b128: groups of 8 lanes = 4 rows x 2 halves: free for any bit) and the k-major staging store
// (8 lanes = 8 consecutive rows of one half, 32 banks: rows r and r + 4 must differ).  Bit 2 ^ bit 4 of r is the solution; the planes
// experiment's bit 3 left a two-way conflict on every k-major store (SQ_LDS_BANK_CONFLICT 0.29 - 0.33 of the LDS cycles on the
// weight-gradient shapes, profiles/r05_gemm_big_pmc.txt).
__device__ __forceinline__ int bx_sw(int r) { return ((r >> 2) ^ (r >> 4)) & 1; }

// TN only: the columns of the output as up to four groups, each with its own B operand and its own output matrix (the weight gradients of
// one LSTM: d gates^T [h2 | emb | h1] land in W_ih's column blocks and in W_hh) -- one launch with 3 - 4x the tiles of the separate products
struct BxGroups {
    int count;                      // 0: plain product (a.seg[0].B, a.out)
    int start[GEMM_MAX_COLGROUPS + 1];
    const float* B[GEMM_MAX_COLGROUPS];
    float* out[GEMM_MAX_COLGROUPS];
    int ldb[GEMM_MAX_COLGROUPS], ldo[GEMM_MAX_COLGROUPS];
};

template <int BM, int BN, int WM, int WN, int WPE, bool PING, bool AROW, bool BROW>
__global__ __launch_bounds__(64 * WM * WN, WPE) void gemm_big_x3_kernel(GemmArgs a, int tiles_m, int tiles_n, BxGroups cg) {
    if (step_dead(a.live)) return;
    extern __shared__ __attribute__((aligned(1024))) unsigned char bx_smem[];
    constexpr int NW = WM * WN, NT = 64 * NW, TM = BM / WM, TN = BN / WN, FI = TM / 32, FU = TN / 32;
    constexpr int RBA = BM / 32, RBB = BN / 32, PA = 3 * RBA, PB = 3 * RBB, SLOT = (PA + PB) * 1024;
    constexpr int IA = 2 * BM / NT, IB = 2 * BN / NT;            // staging items (row, half) per thread and operand
    static_assert(IA >= 1 && IB >= 1 && (2 * BM) % NT == 0 && (2 * BN) % NT == 0, "staging map");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;

    // tile of this workgroup: consecutive workgroups go to the 8 XCDs in turn -> each XCD gets a contiguous run of a grouped
    // (GM row blocks x all column blocks) tile order
    const int tiles = tiles_m * tiles_n;
    const int z = blockIdx.x / tiles;
    int t = blockIdx.x % tiles;
    {
        const int q = tiles / 8, rr = tiles % 8, x = t % 8;
        t = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + t / 8;
    }
    constexpr int GM = 4;
    const int gsz = GM * tiles_n, grp = t / gsz, w = t % gsz;
    const int gm = min(GM, tiles_m - grp * GM);
    const int m0 = (grp * GM + w % gm) * BM, n0 = (w / gm) * BN;
    const int rows_live = a.rows_live ? ((*a.rows_live + 31) & ~31) : 0x7fffffff;
    if (AROW && !BROW && m0 >= rows_live) return;            // NN: a row tile of steps the rollout never ran
    // column group of this tile (group widths are multiples of BN): B operand, output and column origin
    const float* Bsrc = a.seg[0].B;
    float* Cdst = a.out;
    int ldb_g = a.seg[0].ldb, ldo_g = a.ldo, nloc = n0, Nloc = a.N;
    if (!AROW && !BROW && cg.count > 0) {
        int gi = 0;
#pragma unroll
        for (int j = 1; j < GEMM_MAX_COLGROUPS; ++j) gi += (j < cg.count && n0 >= cg.start[j]) ? 1 : 0;
        Bsrc = cg.B[gi]; Cdst = cg.out[gi]; ldb_g = cg.ldb[gi]; ldo_g = cg.ldo[gi];
        nloc = n0 - cg.start[gi]; Nloc = cg.start[gi + 1] - cg.start[gi];
    }

    bx_f32x16 acc[FI][FU];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int u = 0; u < FU; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][u][q] = 0.f;

    const int fo = (2 * r + (h ^ bx_sw(r))) * 16;
    const int fa = (wm * FI) * 1024 + fo, fb = (PA + wn * FU) * 1024 + fo;

    auto run_segment = [&](const GemmSeg& g, int kbeg, int kend) {
        const int nks = (kend - kbeg) / BX_KS;
        if (nks <= 0) return;
        const float* ap[IA];
        const float* bp[IB];
        int da[IA], db[IB];
#pragma unroll
        for (int j = 0; j < IA; ++j) {
            const int item = tid + NT * j;
            const int row = AROW ? item >> 1 : item % BM, hh = AROW ? item & 1 : item / BM;
            int mr = m0 + row;
            if (mr > a.M - 1) mr = a.M - 1;                        // clamped rows feed only never-stored outputs
            ap[j] = AROW ? g.A + (size_t)mr * g.lda + kbeg + 8 * hh : g.A + (size_t)(kbeg + 8 * hh) * g.lda + mr;
            const int rr = row & 31;
            da[j] = (row >> 5) * 1024 + (2 * rr + (hh ^ bx_sw(rr))) * 16;
        }
#pragma unroll
        for (int j = 0; j < IB; ++j) {
            const int item = tid + NT * j;
            const int row = BROW ? item >> 1 : item % BN, hh = BROW ? item & 1 : item / BN;
            int nr = nloc + row;
            if (nr > Nloc - 1) nr = Nloc - 1;
            const float* Bp = (!AROW && !BROW) ? Bsrc : g.B;
            const int ldbp = (!AROW && !BROW) ? ldb_g : g.ldb;
            bp[j] = BROW ? Bp + (size_t)nr * ldbp + kbeg + 8 * hh : Bp + (size_t)(kbeg + 8 * hh) * ldbp + nr;
            const int rr = row & 31;
            db[j] = PA * 1024 + (row >> 5) * 1024 + (2 * rr + (hh ^ bx_sw(rr))) * 16;
        }
        const int ldb_s = (!AROW && !BROW) ? ldb_g : g.ldb;
        const int astep = AROW ? BX_KS : BX_KS * g.lda, bstep = BROW ? BX_KS : BX_KS * ldb_s;          // floats per k-step (< 2^31 / steps: checked by the launcher)
        float ar0[IA][8], br0[IB][8], ar1[IA][8], br1[IB][8];
        auto load_stage = [&](float (&ar)[IA][8], float (&br)[IB][8], int step) {
#pragma unroll
            for (int j = 0; j < IA; ++j) {
                const float* p = ap[j] + step * astep;
                if (AROW) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ar[j][e] = lo[e]; ar[j][4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) ar[j][e] = p[(size_t)e * g.lda];
                }
            }
#pragma unroll
            for (int j = 0; j < IB; ++j) {
                const float* p = bp[j] + step * bstep;
                if (BROW) {
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { br[j][e] = lo[e]; br[j][4 + e] = hi[e]; }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) br[j][e] = p[(size_t)e * ldb_s];
                }
            }
        };
        auto put = [&](unsigned char* base, int planes_stride, const float (&v)[8]) {
            uint32_t s0[4], s1[4], s2[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bx_split3(v[2 * e], v[2 * e + 1], s0[e], s1[e], s2[e]);
            *reinterpret_cast<bx_u32x4*>(base) = (bx_u32x4){s0[0], s0[1], s0[2], s0[3]};
            *reinterpret_cast<bx_u32x4*>(base + planes_stride) = (bx_u32x4){s1[0], s1[1], s1[2], s1[3]};
            *reinterpret_cast<bx_u32x4*>(base + 2 * planes_stride) = (bx_u32x4){s2[0], s2[1], s2[2], s2[3]};
        };
        auto store_stage = [&](int slot, const float (&ar)[IA][8], const float (&br)[IB][8]) {
            unsigned char* sb = bx_smem + slot * SLOT;
#pragma unroll
            for (int j = 0; j < IA; ++j) put(sb + da[j], RBA * 1024, ar[j]);
#pragma unroll
            for (int j = 0; j < IB; ++j) put(sb + db[j], RBB * 1024, br[j]);
        };
        auto compute = [&](int slot) {
            const unsigned char* sb = bx_smem + slot * SLOT;
            constexpr int PP[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};      // smallest terms first
            if (FI >= FU) {
                // B fragments resident, A fragments streamed per 32-row block
                bx_bf16x8 bf[3][FU];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int u = 0; u < FU; ++u) bf[p][u] = *reinterpret_cast<const bx_bf16x8*>(sb + fb + (p * RBB + u) * 1024);
#pragma unroll
                for (int i = 0; i < FI; ++i) {
                    bx_bf16x8 af[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const bx_bf16x8*>(sb + fa + (p * RBA + i) * 1024);
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int u = 0; u < FU; ++u)
                            acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PP[pr][0]], bf[PP[pr][1]][u], acc[i][u], 0, 0, 0);
                }
            } else {
                bx_bf16x8 af[3][FI];
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int i = 0; i < FI; ++i) af[p][i] = *reinterpret_cast<const bx_bf16x8*>(sb + fa + (p * RBA + i) * 1024);
#pragma unroll
                for (int u = 0; u < FU; ++u) {
                    bx_bf16x8 bf[3];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const bx_bf16x8*>(sb + fb + (p * RBB + u) * 1024);
#pragma unroll
                    for (int pr = 0; pr < 6; ++pr)
#pragma unroll
                        for (int i = 0; i < FI; ++i)
                            acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PP[pr][0]][i], bf[PP[pr][1]], acc[i][u], 0, 0, 0);
                }
            }
        };
        // Loads and stores are unconditional (no branches inside a step): the loads of steps behind the segment's end re-read its last
        // step, their pieces go to a slot that is not read again.
        int ls = 0;                                              // k-step the next load_stage reads
        auto load_next = [&](float (&ar)[IA][8], float (&br)[IB][8]) {
            load_stage(ar, br, ls);
            ls = min(ls + 1, nks - 1);
        };
        load_next(ar0, br0);
        load_next(ar1, br1);
        store_stage(0, ar0, br0);
        __syncthreads();
        if (PING) {
            // ping-pong: the second half of the waves (one per SIMD, like the first) runs the same loop HALF a step behind -- while one wave
            // of a SIMD feeds the matrix pipe, the other cuts and stores.  Timeline in half steps, a barrier behind each:
            //     first half :  M(0)  S(1)  M(1)  S(2)  ...  M(n-1)  S(n)   -
            //     second half:  S(1)  M(0)  S(2)  M(1)  ...  S(n)    M(n-1) S(n+1)
            // M(s) reads slot s & 1, S(s) fills slot s & 1 with step s (both halves' shares are in before either reads it; S(n), S(n+1)
            // write re-read data nobody uses).  The second half's register ring therefore runs one step ahead.
            const int g2 = __builtin_amdgcn_readfirstlane(wave) >= NW / 2 ? 1 : 0;
            if (g2) {
                store_stage(1, ar1, br1);
                load_next(ar1, br1);
                __syncthreads();
            }
            auto iter = [&](int s, float (&arF)[IA][8], float (&brF)[IB][8], const float (&arN)[IA][8], const float (&brN)[IB][8]) {
                load_next(arF, brF);
                compute(s & 1);
                __syncthreads();
                store_stage((s + 1 + g2) & 1, arN, brN);
                __syncthreads();
            };
            int s = 0;
            for (; s + 2 <= nks; s += 2) {
                iter(s, ar0, br0, ar1, br1);
                iter(s + 1, ar1, br1, ar0, br0);
            }
            if (s < nks) iter(s, ar0, br0, ar1, br1);
            if (!g2) __syncthreads();
        } else {
            // step s: registers (s & 1) are free and take step s + 2, the other pair holds step s + 1
            auto iter = [&](int s, float (&arF)[IA][8], float (&brF)[IB][8], const float (&arN)[IA][8], const float (&brN)[IB][8]) {
                load_next(arF, brF);
                compute(s & 1);
                store_stage((s + 1) & 1, arN, brN);
                __syncthreads();
            };
            int s = 0;
            for (; s + 2 <= nks; s += 2) {
                iter(s, ar0, br0, ar1, br1);
                iter(s + 1, ar1, br1, ar0, br0);
            }
            if (s < nks) iter(s, ar0, br0, ar1, br1);
        }
    };
    if (a.nseg == 1) {
        // K range of this split: a.chunks_per_split counts 128-deep chunks (gemm_f32)
        const int kbeg = a.nsplit > 1 ? z * a.chunks_per_split * 128 : 0;
        int kend = a.nsplit > 1 ? min(a.seg[0].K, kbeg + a.chunks_per_split * 128) : a.seg[0].K;
        if (!AROW && !BROW) kend = min(kend, rows_live);     // TN: the sum over (t, b) stops behind the last step that ran
        run_segment(a.seg[0], kbeg, kend);
    } else {
        const int gbeg = a.nsplit > 1 ? z * a.chunks_per_split * 128 : 0;
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
    // acc[i][u][q] <-> row m0 + wm TM + 32 i + (q & 3) + 8 (q >> 2) + 4 h, column n0 + wn TN + 32 u + r
    const bool direct = a.nsplit == 1;
    float* const outp = direct ? Cdst : a.out + (size_t)z * a.M * a.N;
    const int ldo = direct ? ldo_g : a.N;
    const int ncol0 = direct ? nloc : n0;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int n = ncol0 + wn * TN + 32 * u + r;
            if (n >= (direct ? Nloc : a.N)) continue;
            const float bias_n = (direct && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * TM + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * h;
                if (m < a.M) {
                    float* o = outp + (size_t)m * ldo + n;
                    const float v = acc[i][u][q] + bias_n;
                    *o = (direct && a.accumulate) ? (*o + v) : v;
                }
            }
        }
}

template <int BM, int BN, int WM, int WN, int WPE, bool PING, bool AROW, bool BROW>
int launch_big(const GemmArgs& a, const BxGroups& cg, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (3 * (BM / 32) + 3 * (BN / 32)) * 1024;
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_big_x3_kernel<BM, BN, WM, WN, WPE, PING, AROW, BROW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int tiles_m = cdiv(a.M, BM), tiles_n = cdiv(a.N, BN);
    hipLaunchKernelGGL((gemm_big_x3_kernel<BM, BN, WM, WN, WPE, PING, AROW, BROW>), dim3(tiles_m * tiles_n * a.nsplit), dim3(64 * WM * WN), lds, st, a, tiles_m, tiles_n, cg);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

template <bool AROW, bool BROW>
int launch_cfg(const GemmArgs& a, int cfg, const BxGroups& cg, hipStream_t st) {
    switch (cfg) {
        case 1: return launch_big<256, 256, 2, 4, 1, false, AROW, BROW>(a, cg, st);
        case 2: return launch_big<128, 256, 2, 2, 2, false, AROW, BROW>(a, cg, st);
        case 3: return launch_big<256, 128, 2, 2, 2, false, AROW, BROW>(a, cg, st);
        case 4: return launch_big<128, 128, 2, 2, 3, false, AROW, BROW>(a, cg, st);
        default: return launch_big<256, 256, 2, 4, 1, true, AROW, BROW>(a, cg, st);
    }
}

}  // namespace

// a with nsplit / chunks_per_split set as for gemm_tn128_x3_kernel (128-deep chunks); every segment's K a multiple of 16
static int big_check(const GemmArgs& a, int cfg) {
    for (int s = 0; s < a.nseg; ++s) ICZ_REQUIRE(a.seg[s].K % BX_KS == 0, "gemm_big_x3: segment %d K = %d is not a multiple of %d", s, a.seg[s].K, BX_KS);
    ICZ_REQUIRE(cfg >= 1 && cfg <= 5, "gemm_big_x3: config %d", cfg);
    for (int s = 0; s < a.nseg; ++s)         // k-step offsets are 32-bit
        ICZ_REQUIRE((double)a.seg[s].K * (double)(a.seg[s].lda > a.seg[s].ldb ? a.seg[s].lda : a.seg[s].ldb) < 2.0e9, "gemm_big_x3: segment %d too large for 32-bit k offsets", s);
    return ICZ_OK;
}

int gemm_big_x3(GemmLayout layout, const GemmArgs& a, int cfg, hipStream_t st) {
    ICZ_TRY(big_check(a, cfg));
    BxGroups cg = {};
    if (layout == GEMM_NT) return launch_cfg<true, true>(a, cfg, cg, st);
    if (layout == GEMM_NN) return launch_cfg<true, false>(a, cfg, cg, st);
    ICZ_REQUIRE(a.nseg == 1, "gemm_big_x3 TN: one K segment");
    return launch_cfg<false, false>(a, cfg, cg, st);
}

int gemm_tn_split_pick(int M, int N, int K) {
    static const bool on = [] { const char* e = getenv("ICZ_GEMM_TN_SPLIT"); return e ? atoi(e) != 0 : true; }();      // A/B switch (read once)
    if (!on) return 1;
    if (!gemm_switches().tn_x3 || gemm_big_switch() == 0 || M < 128 || N < 128 || M % 4 || N % 4 || K % BX_KS || K < 512) return 1;
    const int tiles = cdiv(M, 128) * cdiv(N, 128), chunks = cdiv(K, 128);
    if (tiles >= 256) return 1;                       // the plain route takes it
    int s = 448 / tiles;                              // about two workgroups per CU
    if (s > chunks / 2) s = chunks / 2;               // at least two chunks (256 rows of K) per split
    if (s < 2) return 1;
    return cdiv(chunks, cdiv(chunks, s));             // no empty splits
}

int gemm_tn_split(const float* dY, int ldy, int M, const float* X, int ldx, int N, int K, int nsplit, float* slabs, const int* rows_live, hipStream_t st) {
    ICZ_REQUIRE(dY && X && slabs && nsplit > 1 && nsplit == gemm_tn_split_pick(M, N, K), "gemm_tn_split: shape / split not taken (M %d N %d K %d, %d splits)", M, N, K, nsplit);
    ICZ_REQUIRE(ldy % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)dY & 15) == 0 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)slabs & 15) == 0, "gemm_tn_split: operand alignment");
    GemmArgs a = {};
    a.nseg = 1;
    a.seg[0] = {dY, X, ldy, ldx, K, nullptr};
    a.M = M; a.N = N; a.out = slabs; a.ldo = N; a.nsplit = nsplit; a.chunks_per_split = cdiv(cdiv(K, 128), nsplit);
    a.rows_live = rows_live;
    ICZ_TRY(big_check(a, 4));
    BxGroups cg = {};
    return launch_cfg<false, false>(a, 4, cg, st);
}

bool gemm_tn_grouped_fits(int M, int K, const GemmColGroup* groups, int ngroups) {
    if (!gemm_switches().tn_x3 || gemm_big_switch() == 0 || ngroups < 1 || ngroups > GEMM_MAX_COLGROUPS || M % 4 || K % 32 || K < 64) return false;
    int N = 0;
    for (int j = 0; j < ngroups; ++j) {
        if (groups[j].cols <= 0 || groups[j].cols % 256) return false;
        N += groups[j].cols;
    }
    return cdiv(M, 128) * cdiv(N, 128) >= 256;       // as for the plain TN route: at least one 128 x 128 tile per CU
}

int gemm_tn_grouped(const float* dY, int ldy, int M, int K, const GemmColGroup* groups, int ngroups, const int* rows_live, hipStream_t st) {
    ICZ_REQUIRE(gemm_tn_grouped_fits(M, K, groups, ngroups), "gemm_tn_grouped: shape not taken (M %d K %d, %d groups)", M, K, ngroups);
    BxGroups cg = {};
    cg.count = ngroups;
    int N = 0, ldmax = ldy;
    for (int j = 0; j < ngroups; ++j) {
        const GemmColGroup& q = groups[j];
        ICZ_REQUIRE(q.B && q.out && q.ldb % 4 == 0 && q.ldo % 4 == 0 && ((uintptr_t)q.B & 15) == 0 && ((uintptr_t)q.out & 15) == 0, "gemm_tn_grouped: group %d operands", j);
        cg.start[j] = N; cg.B[j] = q.B; cg.out[j] = q.out; cg.ldb[j] = q.ldb; cg.ldo[j] = q.ldo;
        N += q.cols;
        if (q.ldb > ldmax) ldmax = q.ldb;
    }
    cg.start[ngroups] = N;
    GemmArgs a = {};
    a.nseg = 1;
    a.seg[0] = {dY, groups[0].B, ldy, ldmax, K, nullptr};
    a.M = M; a.N = N; a.out = groups[0].out; a.ldo = groups[0].ldo; a.nsplit = 1; a.chunks_per_split = cdiv(K, 128);
    a.rows_live = rows_live;
    const int cfg = gemm_big_cfg(GEMM_TN, a);
    ICZ_TRY(big_check(a, cfg ? cfg : 4));
    return launch_cfg<false, false>(a, cfg ? cfg : 4, cg, st);
}

// Which tile configuration a shape gets (0: the 128 x 128 two-barrier kernel of gemm_f32.hip).  Measured on MI355X, random operands
// (tools/perf_gemm_big.py, profiles/r05_gemm_big_sweep.txt), us 128 x 128 kernel / 256 x 256 eight waves / 128 x 128 three per CU:
//   TN 4096 x 4096 x 1280: 270 / 227 / 263     TN 4096 x 3072 x 1280: 209 / 187 / 183     TN 10112 x 1024 x 1280: 197 / 164 / 165
//   TN 4096 x 2048 x 1280: 123 / 167 / 125     TN 4096 x 1024 x 1280:  85 / 200 /  86
//   NN 1280 x 1024 x 10112: 199 (4 slabs) / 175 (8) / 169 (8)     NN 1280 x 4096 x 4096: 281 / 258 (2) / 257 (2)
//   NT 1280 x 10112 x 1024: 172 / 139 / 175    NT 2304 x 2048 x 2048: 146 / 139 (2) / 132 (4)   NT 2304 x 1024 x 1024: 47 / 58 (4) / 50 (4)
//   NT 640 x 4096 x 4096 (beam step): 132 (3) / 142 (5) / 132 (3);  in beam search itself the three-per-CU kernel is 1 - 4 % slower
int gemm_big_cfg(GemmLayout layout, const GemmArgs& a) {
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % BX_KS) return 0;          // also under a forced configuration: an ineligible shape falls back to the 128 x 128 kernel
    if (layout == GEMM_TN && a.nseg != 1) return 0;
    const int sw = gemm_big_switch();
    if (sw >= 0) return sw;
    const int t128 = cdiv(a.M, 128) * cdiv(a.N, 128), t256 = cdiv(a.M, 256) * cdiv(a.N, 256);
    const int rounds = cdiv(t256 * a.nsplit, 256);
    const double fill = (double)t256 * a.nsplit / (256.0 * rounds);
    if (layout == GEMM_TN) {
        // isolated, 192 and 160 tiles of 256 x 256 (TD-LSTM group 4096 x 3072, predict 10112 x 1024) tie with the three-per-CU kernel; inside the
        // SCST step the eight-wave kernel is the faster one (five same-box pairs: 5.78 against 5.82 ms per step with the threshold at 0.6 / 0.9)
        if (fill >= 0.6) return 1;
        return t128 >= 600 ? 4 : 0;
    }
    if (layout == GEMM_NN) return a.M >= 1024 ? 4 : 0;
    // NT: beam-search steps (a few hundred rows) stay with the 128 x 128 kernel
    if (a.M < 1024) return 0;
    if (a.nsplit == 1 && t256 >= 192 && fill >= 0.75) return 1;
    return t128 >= 256 ? 4 : 0;
}

}  // namespace icz
