// Skinny NT GEMM with split-precision operands: C[M x N] = X[M x K] W[N x K]^T for M <= 128 rows (the decoder-step GEMMs:
// LSTM gates, dec_att, predict; 64 rows per chain, 128 when the greedy and the sampled chain of an SCST step run merged).
//
// Roofline: the weight matrix is streamed from HBM exactly once per launch (N*K*4 bytes; the activations are a few hundred
// KB and stay in L2), so the kernel is HBM-bound as long as the matrix pipe keeps up.  On the fp32-input MFMA it does not
// (fp32 MFMA runs at the vector rate: 64 x 4096 x 4096 needs 13.7 us of it, the weights 8.4 us of HBM), so every fp32
// operand is multiplied as three bf16 pieces, x = x0 + x1 + x2 (each the bf16 rounding of what the previous ones left:
// 24 mantissa bits, pieces exact), a product being the six piece products of order <= 2 accumulated in fp32
// (v_mfma_f32_16x16x32_bf16; the dropped terms are below 3 * 2^-24 |x w|): 2.7x the fp32-MFMA rate at fp32-level error.
//
// Structure (one 256-thread workgroup per CU, one wave per SIMD, all 512 registers per lane):
//   * wave w owns 16*NCT adjacent output columns and ALL rows; its weight rows go HBM -> registers directly (each weight
//     element is used by one wave only: an LDS round trip would buy nothing), as fragment-shaped loads that cover 64
//     contiguous bytes of 16 rows per instruction, D - 1 stages (64 k each) ahead of their use: ~64 KB in flight per CU.
//     The fp32 -> 3 x bf16 split of the weights is VALU work in the shadow of the MFMAs of the previous k block;
//   * the activation tile of a stage (rows x 64 k, shared by the four waves) is loaded one stage ahead, split once and
//     written to LDS as three bf16 planes (double buffered, one barrier per stage), from which the A fragments are plain
//     16-byte reads (row stride 160 B: conflict-free);
//   * inside every 32-deep k block the k index is permuted (element e of lane quarter q is k = 4 q + e for e < 4 and
//     16 + 4 q + e - 4 above) so that a weight load instruction reads 64 CONTIGUOUS bytes per row; the activation planes
//     are written to LDS in the same order;
//   * split-K over blockIdx.z into slabs [z][M][N], summed in fixed order by the consumers (as gemm_f32.hip).
//
// MEASURED (round 2, MI355X, rocprofv3 kernel durations, 64 x 4096 x 4096 with split 8; tools/prof_shapes.sh,
// tools/perf_skinny_stamps.py): 23 - 25 us in every variant below against 25.8 us for the fp32-MFMA kernel, i.e. no gain that
// would justify switching the decoder steps over, hence OPT-IN (ICZ_GEMM_SKINNY_X3=1; results stay inside every parity bound:
// tests/test_gpu_butd.py runs the suite's GEMM shapes and a decode through it).  What the in-kernel clock stamps and the
// ablations show:
//   * per workgroup: ~8600 cycles of prologue (first HBM round trips), ~3200 - 4000 per 64-deep stage, ~4200 of epilogue
//     (32 four-byte store instructions per lane); eight stages per workgroup, so 30 % of a launch is fixed cost;
//   * a stage costs the same with the MFMAs removed (-12 %), with the weight loads removed (-15 %), with the activation
//     staging removed (-13 %), with two waves per SIMD (NW = 8: +-0) and with the loads spread over the MFMA groups instead of
//     issued in one burst (load-issue phase 1140 -> 190 cycles, compute phase 2600 -> 3000): what the variants share is the
//     number of BYTES a compute unit pulls through its vector-memory path per stage -- 32 KB of weights + 16 KB (fp32) or 24 KB
//     (bf16 planes) of activations, the latter re-read by each of the 256 workgroups from L2.  A wave-wide 16-byte load costs
//     the issuing wave ~70 - 80 cycles however it is placed, i.e. ~14 bytes per cycle per compute unit for weights and
//     activations TOGETHER: (32 + 16) KB / 14 B = 3400 cycles per stage, which is what every variant measures.  The same
//     arithmetic gives the fp32 kernel's time (64-column tiles: 32 KB of activations per 32 KB of weights, 4600 cycles per
//     128-deep stage, 25.8 us), so at 64 rows these GEMMs are bound by the L1 path, activations counted in, not by HBM (8.4
//     us) and not by the matrix pipe; pre-split activation planes (GemmSeg::Apl, 6 bytes per element instead of 4) therefore
//     buy back in VALU time what they cost in bytes (25.1 -> 25.0 us);
//   * wider column tiles would halve the activation share, but 256 columns per workgroup need 64 weight registers per stage
//     and ring slot: hipcc caps the kernel at 256 VGPRs (accumulators aside) and the variant spills (35 us).
#include <stdlib.h>

#include <type_traits>

#include "gemm_f32.h"
#include "split3_planes.h"

namespace icz {

// compile-time loop: f(integral_constant<int, I>) for I = 0 .. N - 1 (register arrays need constant indices)
template <int I, int N, class F>
__device__ __forceinline__ void sk_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sk_static_for<I + 1, N>(f);
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t sk_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t sk_u32x2;

__device__ __forceinline__ uint32_t sk_cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a, b) -> three packed bf16 pairs (a in the low half): p0 + p1 + p2 == the fp32 values up to 2^-24 relative
__device__ __forceinline__ void sk_split3(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = sk_cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = sk_cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = sk_cvt_pk_bf16(ra, rb);
}

constexpr int SK_BK = 64;                 // k per pipeline stage (two MFMA k blocks of 32)
constexpr int SK_PB = 80;                 // bf16 per LDS row: 160 B (64 k + 16 pad) -> conflict-free 16-byte fragment reads
constexpr size_t sk_lds_bytes(int MT) { return (size_t)2 * 3 * (16 * MT) * SK_PB * 2; }     // 2 buffers x 3 planes x rows x 160 B

template <int MT /* 16-row tiles: 4 (64 rows) or 8 (128 rows) */, int NCT /* 16-column tiles per wave */, int D /* weight ring depth */,
          int ABL = 0 /* tools/ only: 1 no MFMA, 2 no weight loads in the loop, 3 no activation staging in the loop, 4 time stamps */,
          int NW = 4 /* waves: 4, or 8 = two per SIMD, wave (cg, kh) taking k block kh of every stage for column group cg */,
          bool XPL = false /* the activations come as bf16 piece planes from their producer (GemmSeg::Apl): staged by plain copies */>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_x3_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    unsigned short* const planes = reinterpret_cast<unsigned short*>(sk_smem);
    constexpr int ROWS = 16 * MT;
    constexpr size_t PLANE = (size_t)ROWS * SK_PB;             // bf16 elements of one plane of one buffer
    constexpr int NTHR = 64 * NW, NKH = NW / 4, NB = 2 / NKH;  // k blocks (of 32) per stage and wave
    constexpr int XI = ROWS * 8 / NTHR;                        // staging items (row, 8-k group) per thread and stage
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(XI >= 1, "at least one staging item per thread");
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, kh = tid >> 8;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * (64 * NCT), z = blockIdx.z;
    int tot = 0;
#pragma unroll
    for (int s = 0; s < GEMM_MAX_SEG; ++s)
        if (s < a.nseg) tot += a.seg[s].K / SK_BK;
    const int c_begin = z * a.chunks_per_split;
    const int c_end = min(tot, c_begin + a.chunks_per_split);
    const int n = c_end - c_begin;
    // ABL 4 (tools/perf_skinny_stamps.py): wave 0 of every workgroup records the shader clock at kernel entry, after the
    // prologue, after every stage and after the epilogue into the buffer passed as a.bias (32 uint64 per workgroup)
    unsigned long long* const stamps = ABL >= 4 ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.bias)) +
                                                      32 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x) : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (ABL >= 4) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (tid == 0 && i < 32) stamps[i] = t;
        }
    };
    stamp(0);

    int ncol[NCT];
    size_t ncol_c[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) {
        ncol[c] = n0 + (wave * NCT + c) * 16 + li;                 // this lane's weight row (= output column) of column tile c
        ncol_c[c] = ncol[c] < a.N ? ncol[c] : a.N - 1;             // clamped rows feed only never-stored outputs
    }
    f32x4 acc[MT][NCT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- activation staging: item -> (row, kg): 8 consecutive k of one row = two float4; eight lanes cover a row's 256 bytes
    size_t xrow[XI];
    int xk[XI], xpos[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int item = tid + NTHR * j, row = item >> 3, kg = item & 7;
        xrow[j] = (size_t)(row < a.M ? row : a.M - 1);
        xk[j] = 8 * kg;
        // LDS position of the item's first float4 inside its row (bf16 units): block (kg >> 2), k permutation of the header
        const int kk0 = 8 * (kg & 3), half = kk0 >> 4, q0 = (kk0 & 15) >> 2;
        xpos[j] = row * SK_PB + 32 * (kg >> 2) + 8 * q0 + 4 * half;      // second float4: + 8 (next quarter, same half)
        if (XPL) xpos[j] = row * SK_PB + 8 * kg;                         // planes: 16-byte piece kg of the row, already in k order
    }

    // ---- stage cursors (segment, k offset); the weight cursor runs D - 1 stages ahead of the activation cursor's stage
    struct Cur { int seg, k0, segK; };
    Cur cx = {0, 0, 0}, cw = {0, 0, 0};
    const float* wp[NCT];
    const float* xp[XI];
    const unsigned short* xpp[XI];
    long long plstride = 0;
    auto seek = [&](Cur& c, int stage) __attribute__((always_inline)) {
        int q = stage;
        c.seg = 0;
#pragma unroll
        for (int s = 0; s < GEMM_MAX_SEG - 1; ++s) {
            if (c.seg == s && s < a.nseg - 1) {
                const int nst = a.seg[s].K / SK_BK;
                if (q >= nst) { q -= nst; c.seg = s + 1; }
            }
        }
        c.k0 = q * SK_BK;
        c.segK = a.seg[c.seg].K;
    };
    auto point_w = [&]() __attribute__((always_inline)) {
        const GemmSeg& g = a.seg[cw.seg];
#pragma unroll
        for (int c = 0; c < NCT; ++c) wp[c] = g.B + ncol_c[c] * g.ldb + cw.k0 + 32 * NB * kh + 4 * lq;
    };
    auto point_x = [&]() __attribute__((always_inline)) {
        const GemmSeg& g = a.seg[cx.seg];
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            if (XPL) xpp[j] = g.Apl + xrow[j] * g.lda + cx.k0 + xk[j];
            else xp[j] = g.A + xrow[j] * g.lda + cx.k0 + xk[j];
        }
        if (XPL) plstride = g.Apl_stride;
    };
    // inside a segment a step is "pointer += 64 floats"; the (rare) segment switch re-reads the descriptor behind a branch that
    // holds no vector-memory operation (the compiler's load counting stays exact across it)
    auto advance_w = [&]() __attribute__((always_inline)) {
        cw.k0 += SK_BK;
        if (__builtin_expect(cw.k0 >= cw.segK && cw.seg < a.nseg - 1, 0)) {
            ++cw.seg; cw.k0 = 0; cw.segK = a.seg[cw.seg].K;
            point_w();
        } else {
#pragma unroll
            for (int c = 0; c < NCT; ++c) wp[c] += SK_BK;
        }
    };
    auto advance_x = [&]() __attribute__((always_inline)) {
        cx.k0 += SK_BK;
        if (__builtin_expect(cx.k0 >= cx.segK && cx.seg < a.nseg - 1, 0)) {
            ++cx.seg; cx.k0 = 0; cx.segK = a.seg[cx.seg].K;
            point_x();
        } else {
#pragma unroll
            for (int j = 0; j < XI; ++j) { if (XPL) xpp[j] += SK_BK; else xp[j] += SK_BK; }
        }
    };

    // weight ring: w[slot][block][column tile][lo / hi float4]  (k = 32 blk + 4 q + e  and  32 blk + 16 + 4 q + e)
    f32x4 w[D][NB][NCT][2];
    f32x4 xr[XI][2];
    sk_u32x4 xq[XI][3];
    auto load_w = [&](auto slot) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        sk_static_for<0, NB * NCT>([&](auto q) {
            constexpr int b = decltype(q)::value / NCT, c = decltype(q)::value % NCT;
            w[S][b][c][0] = *reinterpret_cast<const f32x4*>(wp[c] + 32 * b);
            w[S][b][c][1] = *reinterpret_cast<const f32x4*>(wp[c] + 32 * b + 16);
        });
    };
    auto load_x = [&]() __attribute__((always_inline)) {
        sk_static_for<0, XI>([&](auto jj) {
            constexpr int j = decltype(jj)::value;
            if constexpr (XPL) {
                xq[j][0] = *reinterpret_cast<const sk_u32x4*>(xpp[j]);
                xq[j][1] = *reinterpret_cast<const sk_u32x4*>(xpp[j] + plstride);
                xq[j][2] = *reinterpret_cast<const sk_u32x4*>(xpp[j] + 2 * plstride);
            } else {
                xr[j][0] = *reinterpret_cast<const f32x4*>(xp[j]);
                xr[j][1] = *reinterpret_cast<const f32x4*>(xp[j] + 4);
            }
        });
    };
    // the same loads one at a time (spread over the MFMA groups of a stage: issued in one burst at the stage's top they stall
    // the wave for ~80 cycles each while nothing else runs): weight load i = (block, column tile, lo / hi); activation load i
    constexpr int LW = NB * NCT * 2, LX = XPL ? 3 * XI : 2 * XI;
    auto load_w1 = [&](auto slot, auto idx) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value, i = decltype(idx)::value, b = i / (2 * NCT), c = (i / 2) % NCT, h = i & 1;
        w[S][b][c][h] = *reinterpret_cast<const f32x4*>(wp[c] + 32 * b + 16 * h);
    };
    auto load_x1 = [&](auto idx) __attribute__((always_inline)) {
        constexpr int i = decltype(idx)::value;
        if constexpr (XPL) xq[i / 3][i % 3] = *reinterpret_cast<const sk_u32x4*>(xpp[i / 3] + (i % 3) * plstride);
        else xr[i / 2][i & 1] = *reinterpret_cast<const f32x4*>(xp[i / 2] + 4 * (i & 1));
    };
    // The split work is cut into pieces of one packed pair (11 VALU instructions) so that it can be dealt out over the MFMA
    // groups of a stage:
    //   weight piece (block b, column tile c, word i = 0..3): floats 2 i, 2 i + 1 of the lane's eight -> word i of the three
    //                 B-fragment planes;
    //   activation piece u = 0 .. 4 XI - 1: item u >> 2, float4 (u >> 1) & 1, pair u & 1; the second pair of a float4 also
    //                 stores the 8-byte pieces of the three planes.
    sk_u32x4 bq[2][NCT][3];                // B fragments (three planes) of the block in use [0] and of the next one [1]
    uint32_t xw[3][2];
    // every index below is a compile-time constant (integral_constant arguments, sk_static_for loops): register arrays
    // indexed by anything else end up in scratch memory
    auto w_piece = [&](auto slot, auto blk, auto piece, auto which) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value, B_ = decltype(blk)::value, P = decltype(piece)::value, Wh = decltype(which)::value;
        constexpr int c = P >> 2, i = P & 3;
        uint32_t p0, p1, p2;
        sk_split3(w[S][B_][c][i >> 1][2 * (i & 1)], w[S][B_][c][i >> 1][2 * (i & 1) + 1], p0, p1, p2);
        bq[Wh][c][0][i] = p0; bq[Wh][c][1][i] = p1; bq[Wh][c][2][i] = p2;
    };
    auto x_piece = [&](auto piece, int buf) __attribute__((always_inline)) {
        constexpr int u = decltype(piece)::value, j = u >> 2, hlf = (u >> 1) & 1, pr = u & 1;
        if constexpr (XPL) {        // plane (u & 3) of item j, one 16-byte copy; every fourth piece is empty
            if constexpr ((u & 3) < 3)
                *reinterpret_cast<sk_u32x4*>(planes + (size_t)buf * 3 * PLANE + (u & 3) * PLANE + xpos[j]) = xq[j][u & 3];
            return;
        }
        sk_split3(xr[j][hlf][2 * pr], xr[j][hlf][2 * pr + 1], xw[0][pr], xw[1][pr], xw[2][pr]);
        if (pr == 1) {
            unsigned short* o = planes + (size_t)buf * 3 * PLANE + xpos[j] + 8 * hlf;
            *reinterpret_cast<sk_u32x2*>(o) = (sk_u32x2){xw[0][0], xw[0][1]};
            *reinterpret_cast<sk_u32x2*>(o + PLANE) = (sk_u32x2){xw[1][0], xw[1][1]};
            *reinterpret_cast<sk_u32x2*>(o + 2 * PLANE) = (sk_u32x2){xw[2][0], xw[2][1]};
        }
    };
    auto store_x = [&](int buf) __attribute__((always_inline)) {
        sk_static_for<0, 4 * XI>([&](auto u) { x_piece(u, buf); });
    };
    constexpr int WP = 4 * NCT;            // weight pieces per block
    // One 64-deep stage from ring slot S and LDS buffer `buf`, as 2 MT groups of 6 NCT MFMAs (one 16-row tile of one k block).
    // Software pipeline, pinned group by group (the compiler otherwise runs all VALU work, then all MFMAs):
    //   * the A fragments of group g + 1 are read at the top of group g;
    //   * the B fragments of the NEXT block (block 1 of this stage, then block 0 of the next stage: NEXT_W) are split during
    //     this block's groups, WP / MT pieces per group;
    //   * the activations of the next stage (NEXT_X) are split and written to the other LDS buffer, one piece per group.
    // bq[0] holds this stage's block 0 on entry and the next stage's block 0 on exit.
    auto compute = [&](auto slot, auto next_w, auto next_x, auto spread, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value;
        constexpr bool NEXT_W = decltype(next_w)::value, NEXT_X = decltype(next_x)::value;
        constexpr bool SPREAD = decltype(spread)::value;      // this stage also issues the loads of stage s + D - 1 / s + 1, group by group
        constexpr int G = NB * MT, HG = G / 2, XP = 4 * XI;
        const unsigned short* abase = planes + (size_t)buf * 3 * PLANE + li * SK_PB + 8 * lq + 32 * NB * kh;
        sk_bf16x8 af[2][3];
        sk_static_for<0, 3>([&](auto pp) { af[0][decltype(pp)::value] = *reinterpret_cast<const sk_bf16x8*>(abase + decltype(pp)::value * PLANE); });
        sk_static_for<0, NB * MT>([&](auto gc) {
            constexpr int g = decltype(gc)::value, b = g / MT, t = g % MT;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g + 1 < NB * MT) {
                constexpr int b1 = (g + 1) / MT, t1 = (g + 1) % MT;
                const unsigned short* ap = abase + (size_t)t1 * 16 * SK_PB + 32 * b1;
                sk_static_for<0, 3>([&](auto pp) { af[(g + 1) & 1][decltype(pp)::value] = *reinterpret_cast<const sk_bf16x8*>(ap + decltype(pp)::value * PLANE); });
            }
            // VALU work dealt to this group: the next block's B fragments (this stage's, or block 0 of the next stage's slot)
            if constexpr (b + 1 < NB || NEXT_W) {
                sk_static_for<t * WP / MT, (t + 1) * WP / MT>([&](auto pc) {
                    if constexpr (b + 1 < NB) w_piece(slot, std::integral_constant<int, b + 1>{}, pc, std::integral_constant<int, 1>{});
                    else w_piece(std::integral_constant<int, (S + 1) % D>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 1>{});
                });
            }
            constexpr bool XON = NEXT_X && ABL != 3 && ABL != 13 && ABL != 15;
            if constexpr (SPREAD) {
                if constexpr (ABL != 2 && ABL != 12 && ABL != 15)
                    sk_static_for<g * LW / G, (g + 1) * LW / G>([&](auto ic) { load_w1(std::integral_constant<int, (S + D - 1) % D>{}, ic); });
                if constexpr (XON && g < HG) sk_static_for<g * LX / HG, (g + 1) * LX / HG>([&](auto ic) { load_x1(ic); });
                if constexpr (XON && g >= HG) sk_static_for<(g - HG) * XP / HG, (g - HG + 1) * XP / HG>([&](auto ic) { x_piece(ic, buf ^ 1); });
            } else if constexpr (XON) {
                x_piece(gc, buf ^ 1);
            }
            sk_static_for<0, NCT>([&](auto cc) {       // smallest terms first
                constexpr int c = decltype(cc)::value;
                const sk_bf16x8 b0 = __builtin_bit_cast(sk_bf16x8, bq[0][c][0]), b1_ = __builtin_bit_cast(sk_bf16x8, bq[0][c][1]),
                                b2 = __builtin_bit_cast(sk_bf16x8, bq[0][c][2]);
                f32x4 v = acc[t][c];
                if constexpr (ABL == 1 || ABL == 11) {
                    v[0] += __builtin_bit_cast(f32x4, af[g & 1][0])[0] + __builtin_bit_cast(f32x4, b0)[1] + __builtin_bit_cast(f32x4, af[g & 1][1])[2] +
                            __builtin_bit_cast(f32x4, b1_)[3] + __builtin_bit_cast(f32x4, af[g & 1][2])[0] + __builtin_bit_cast(f32x4, b2)[1];
                    acc[t][c] = v;
                    return;
                }
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][2], b0, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b2, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][1], b1_, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][1], b0, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b1_, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b0, v, 0, 0, 0);
                acc[t][c] = v;
            });
            // issue order inside the group: the fragment reads first, then every MFMA followed by its share of the VALU work
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            if constexpr (SPREAD) __builtin_amdgcn_sched_group_barrier(0x020, (g + 1) * LW / G - g * LW / G + (g < HG ? (g + 1) * LX / HG - g * LX / HG : 0), 0);
#pragma unroll
            for (int i = 0; i < 6 * NCT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
            if constexpr (t == MT - 1) {        // the next block's fragments take over
                __builtin_amdgcn_sched_barrier(0);
                sk_static_for<0, NCT * 3>([&](auto q) { bq[0][decltype(q)::value / 3][decltype(q)::value % 3] = bq[1][decltype(q)::value / 3][decltype(q)::value % 3]; });
            }
        });
    };

    if (n > 0) {
        seek(cx, c_begin);
        seek(cw, c_begin);
        point_x();
        point_w();
        load_x();
        // prologue: the first D - 1 stages of weights, the B fragments of stage 0 / block 0, the activations of stage 0
        sk_static_for<0, D - 1>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if (I < n) {
                if (I > 0) advance_w();
                load_w(ic);
            }
        });
        store_x(0);
        sk_static_for<0, WP>([&](auto pc) { w_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 0>{}); });
        __syncthreads();
        stamp(1);
        // stage s: activations of s + 1 and weights of s + D - 1 go out at its top (activations first: vmcnt retires in
        // issue order, and the activations are needed first).  FULL = both exist: the steady-state instance has no branch.
        auto stage = [&](auto slot, auto full, int s) __attribute__((always_inline)) {
            constexpr int S = decltype(slot)::value;
            constexpr bool FULL = decltype(full)::value;
            const bool nx = FULL || s + 1 < n, nw = FULL || s + D - 1 < n;
            if (FULL) {          // steady state: the loads go out group by group inside compute()
                advance_x();
                advance_w();
                __builtin_amdgcn_sched_barrier(0);
                stamp(2 + 3 * s);
                compute(slot, std::true_type{}, std::true_type{}, std::true_type{}, s & 1);
            } else {
                if (nx && ABL != 3 && ABL != 13 && ABL != 15) { advance_x(); load_x(); }
                if (nw && ABL != 2 && ABL != 12 && ABL != 15) {
                    advance_w();
                    load_w(std::integral_constant<int, (S + D - 1) % D>{});
                }
                // the loads go out HERE: left alone, the scheduler sinks them to their first use (shorter live ranges)
                __builtin_amdgcn_sched_barrier(0);
                stamp(2 + 3 * s);
                if (nx) compute(slot, std::true_type{}, std::true_type{}, std::false_type{}, s & 1);      // a next stage exists: its weights are loaded
                else compute(slot, std::false_type{}, std::false_type{}, std::false_type{}, s & 1);
            }
            stamp(3 + 3 * s);
            __syncthreads();
            stamp(4 + 3 * s);
        };
        int s = 0;
        for (; s + 2 * D - 1 <= n; s += D)          // every stage of the group has its activations and weights to prefetch
            sk_static_for<0, D>([&](auto ic) { stage(ic, std::true_type{}, s + decltype(ic)::value); });
        for (; s < n; s += D)
            sk_static_for<0, D>([&](auto ic) {
                constexpr int I = decltype(ic)::value;
                if (s + I < n) stage(ic, std::false_type{}, s + I);
            });
    }

    // ---- the two k halves meet in LDS (the plane buffers are free after the last barrier): waves kh = 1 park their tiles
    float* const red = reinterpret_cast<float*>(sk_smem);
    if constexpr (NKH == 2) {
        if (kh == 1) {
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j) red[(size_t)(16 * t + 4 * lq + j) * (64 * NCT) + (wave * NCT + c) * 16 + li] = acc[t][c][j];
        }
        __syncthreads();
    }
    // ---- epilogue: acc[t][c][j] <-> row 16 t + 4 q + j, column ncol[c]
    const bool direct = a.nsplit == 1;
    float* const outp = direct ? a.out : a.out + (size_t)z * a.M * a.N;
    const int ldo = direct ? a.ldo : a.N;
    if (kh == 0) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            if (ncol[c] >= a.N) continue;
            const float bias = (direct && a.bias && ABL < 4) ? a.bias[ncol[c]] : 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int m = 16 * t + 4 * lq + j;
                    if (m < a.M) {
                        float* o = outp + (size_t)m * ldo + ncol[c];
                        float v = acc[t][c][j] + bias;
                        if constexpr (NKH == 2) v += red[(size_t)m * (64 * NCT) + (wave * NCT + c) * 16 + li];
                        *o = (direct && a.accumulate) ? (*o + v) : v;
                    }
                }
        }
    }
    stamp(31);
}

static int sk_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

// ------------------------------------------------------------------------------------------------
// Second structure, for the two LSTM-gate GEMMs of a decoder step (N = 4 H output columns, K = 3 H / 4 H): the activations
// of the workgroup's WHOLE k range (NSR stages of 64) are split ONCE and stay in LDS as three bf16 planes, and the workgroup
// then streams TPW column tiles of 128 one after the other through them.  What this buys, by the MEASURED model of the
// header (a compute unit moves ~14 bytes per cycle through its vector-memory path, activations included):
//   * activation bytes per weight byte 0.25 instead of 0.5 (one 64 x 256 fp32 read per 2 x 128 x 256 weights), and the
//     activation split done once per workgroup instead of once per stage;
//   * no barrier after the first one: the planes are read-only, the four waves run free through the TPW * NSR pipeline steps
//     (weight ring as above, loads spread over the MFMA groups), each wave's accumulators leaving through its own LDS strip as
//     full 16-byte row segments when a tile is done.
// Everything is compile-time (steps, ring slots, tile boundaries): no branch with a vector-memory operation in it.
// Launch: grid (N / (128 TPW), 1, chunks / NSR); every workgroup has exactly NSR chunks (the launcher checks).
constexpr int RS_PB = 64 * 4 + 16;         // bf16 per LDS row for NSR = 4: 544 B -> conflict-free 16-byte fragment reads
template <int MT, int NSR, int TPW, int D>
constexpr size_t rs_lds_bytes() { return (size_t)3 * (16 * MT) * RS_PB * 2 + (size_t)4 * (16 * MT) * 32 * 4; }

template <int MT, int NSR, int TPW, int D, bool STAMPS = false>
__global__ __launch_bounds__(256) void gemm_resident_x3_kernel(GemmArgs a) {
    static_assert(NSR == 4, "row stride RS_PB is laid out for four stages");
    unsigned long long* const stamps = STAMPS ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.bias)) +
                                                    32 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x) : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (STAMPS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0 && i < 32) stamps[i] = t;
        }
    };
    stamp(0);
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    unsigned short* const planes = reinterpret_cast<unsigned short*>(sk_smem);
    constexpr int ROWS = 16 * MT, NCT = 2, NT = TPW * NSR;
    constexpr size_t PLANE = (size_t)ROWS * RS_PB;
    constexpr int XI = ROWS * 8 / 256;                         // staging items (row, 8-k group) per thread and stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int n0 = blockIdx.x * (128 * TPW), z = blockIdx.z;
    const int c_begin = z * NSR;
    float* const strip = reinterpret_cast<float*>(sk_smem + 3 * PLANE * 2) + (size_t)wave * ROWS * 32;      // this wave's epilogue staging

    struct Cur { int seg, k0; };
    auto seek = [&](int stage) __attribute__((always_inline)) {
        Cur c = {0, 0};
        int q = stage;
#pragma unroll
        for (int sg = 0; sg < GEMM_MAX_SEG - 1; ++sg) {
            if (c.seg == sg && sg < a.nseg - 1) {
                const int nst = a.seg[sg].K / SK_BK;
                if (q >= nst) { q -= nst; c.seg = sg + 1; }
            }
        }
        c.k0 = q * SK_BK;
        return c;
    };

    // ---- phase B: NT pipeline steps, step i = (tile i / NSR, stage i % NSR)
    f32x4 acc[MT][NCT];
    f32x4 w[D][2][NCT][2];
    sk_u32x4 bq[2][NCT][3];
    const float* wp[NCT];
    auto point_w = [&](auto stepc) __attribute__((always_inline)) {       // weight pointers of pipeline step i
        constexpr int i = decltype(stepc)::value, tile = i / NSR, st = i % NSR;
        const Cur c = seek(c_begin + st);
        const GemmSeg& g = a.seg[c.seg];
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) {
            const int col = n0 + 128 * tile + (wave * NCT + cc) * 16 + li;
            wp[cc] = g.B + (size_t)(col < a.N ? col : a.N - 1) * g.ldb + c.k0 + 4 * lq;
        }
    };
    constexpr int LW = 2 * NCT * 2, WP = 4 * NCT, G = 2 * MT;
    auto load_w1 = [&](auto slot, auto idx) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value, i = decltype(idx)::value, b = i / (2 * NCT), c = (i / 2) % NCT, h = i & 1;
        w[S][b][c][h] = *reinterpret_cast<const f32x4*>(wp[c] + 32 * b + 16 * h);
    };
    auto w_piece = [&](auto slot, auto blk, auto piece, auto which) __attribute__((always_inline)) {
        constexpr int S = decltype(slot)::value, B_ = decltype(blk)::value, P = decltype(piece)::value, Wh = decltype(which)::value;
        constexpr int c = P >> 2, i = P & 3;
        uint32_t p0, p1, p2;
        sk_split3(w[S][B_][c][i >> 1][2 * (i & 1)], w[S][B_][c][i >> 1][2 * (i & 1) + 1], p0, p1, p2);
        bq[Wh][c][0][i] = p0; bq[Wh][c][1][i] = p1; bq[Wh][c][2][i] = p2;
    };
    // ---- phase A: the activations of the whole k range -> three bf16 planes in LDS (k order permuted inside 32-blocks)
    {
        f32x4 xr[NSR][XI][2];
        sk_static_for<0, NSR>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            const Cur c = seek(c_begin + st);
            const GemmSeg& g = a.seg[c.seg];
            sk_static_for<0, XI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int item = tid + 256 * j, row = item >> 3, kg = item & 7;
                const float* xp = g.A + (size_t)(row < a.M ? row : a.M - 1) * g.lda + c.k0 + 8 * kg;
                xr[st][j][0] = *reinterpret_cast<const f32x4*>(xp);
                xr[st][j][1] = *reinterpret_cast<const f32x4*>(xp + 4);
            });
        });
        // the weights of the first D - 1 pipeline steps go out now: their HBM latency passes behind the split below
        sk_static_for<0, D - 1>([&](auto ic) {
            point_w(ic);
            sk_static_for<0, LW>([&](auto lc) { load_w1(ic, lc); });
        });
        __builtin_amdgcn_sched_barrier(0);
        sk_static_for<0, NSR>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            sk_static_for<0, XI>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int item = tid + 256 * j, row = item >> 3, kg = item & 7;
                const int kk0 = 8 * (kg & 3), half = kk0 >> 4, q0 = (kk0 & 15) >> 2;
                unsigned short* o0 = planes + (size_t)row * RS_PB + 64 * st + 32 * (kg >> 2) + 8 * q0 + 4 * half;
                sk_static_for<0, 2>([&](auto hc) {
                    constexpr int hlf = decltype(hc)::value;
                    uint32_t a0, a1, a2, b0, b1, b2;
                    sk_split3(xr[st][j][hlf][0], xr[st][j][hlf][1], a0, a1, a2);
                    sk_split3(xr[st][j][hlf][2], xr[st][j][hlf][3], b0, b1, b2);
                    unsigned short* o = o0 + 8 * hlf;
                    *reinterpret_cast<sk_u32x2*>(o) = (sk_u32x2){a0, b0};
                    *reinterpret_cast<sk_u32x2*>(o + PLANE) = (sk_u32x2){a1, b1};
                    *reinterpret_cast<sk_u32x2*>(o + 2 * PLANE) = (sk_u32x2){a2, b2};
                });
            });
        });
    }

    stamp(1);
    __syncthreads();                        // the planes are complete (the only barrier of the kernel)
    stamp(2);
    sk_static_for<0, WP>([&](auto pc) { w_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 0>{}); });

    sk_static_for<0, NT>([&](auto stepc) {
        constexpr int i = decltype(stepc)::value, S = i % D, tile = i / NSR, st = i % NSR;
        constexpr bool HAS_LOAD = i + D - 1 < NT, HAS_NEXT = i + 1 < NT;
        if constexpr (st == 0) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (HAS_LOAD) point_w(std::integral_constant<int, i + D - 1>{});
        const unsigned short* abase = planes + (size_t)li * RS_PB + 64 * st + 8 * lq;
        sk_bf16x8 af[2][3];
        sk_static_for<0, 3>([&](auto pp) { af[0][decltype(pp)::value] = *reinterpret_cast<const sk_bf16x8*>(abase + decltype(pp)::value * PLANE); });
        sk_static_for<0, G>([&](auto gc) {
            constexpr int g = decltype(gc)::value, b = g / MT, t = g % MT;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (g + 1 < G) {
                constexpr int b1 = (g + 1) / MT, t1 = (g + 1) % MT;
                const unsigned short* ap = abase + (size_t)t1 * 16 * RS_PB + 32 * b1;
                sk_static_for<0, 3>([&](auto pp) { af[(g + 1) & 1][decltype(pp)::value] = *reinterpret_cast<const sk_bf16x8*>(ap + decltype(pp)::value * PLANE); });
            }
            if constexpr (HAS_LOAD)
                sk_static_for<g * LW / G, (g + 1) * LW / G>([&](auto lc) { load_w1(std::integral_constant<int, (S + D - 1) % D>{}, lc); });
            if constexpr (b == 0 || HAS_NEXT) {
                sk_static_for<t * WP / MT, (t + 1) * WP / MT>([&](auto pc) {
                    if constexpr (b == 0) w_piece(std::integral_constant<int, S>{}, std::integral_constant<int, 1>{}, pc, std::integral_constant<int, 1>{});
                    else w_piece(std::integral_constant<int, (S + 1) % D>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 1>{});
                });
            }
            sk_static_for<0, NCT>([&](auto cc) {       // smallest terms first
                constexpr int c = decltype(cc)::value;
                const sk_bf16x8 b0 = __builtin_bit_cast(sk_bf16x8, bq[0][c][0]), b1_ = __builtin_bit_cast(sk_bf16x8, bq[0][c][1]),
                                b2 = __builtin_bit_cast(sk_bf16x8, bq[0][c][2]);
                f32x4 v = acc[t][c];
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][2], b0, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b2, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][1], b1_, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][1], b0, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b1_, v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[g & 1][0], b0, v, 0, 0, 0);
                acc[t][c] = v;
            });
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            if constexpr (HAS_LOAD) __builtin_amdgcn_sched_group_barrier(0x020, (g + 1) * LW / G - g * LW / G, 0);
#pragma unroll
            for (int k = 0; k < 6 * NCT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            }
            if constexpr (t == MT - 1) {
                __builtin_amdgcn_sched_barrier(0);
                sk_static_for<0, NCT * 3>([&](auto q) { bq[0][decltype(q)::value / 3][decltype(q)::value % 3] = bq[1][decltype(q)::value / 3][decltype(q)::value % 3]; });
            }
        });
        stamp(3 + 2 * i);
        if constexpr (st == NSR - 1) {
            // ---- the tile is done: accumulators -> this wave's LDS strip [ROWS][32] -> 16-byte row segments of the slab / output
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int j = 0; j < 4; ++j) strip[(16 * t + 4 * lq + j) * 32 + 16 * c + li] = acc[t][c][j];
            const bool direct = a.nsplit == 1;
            float* const outp = direct ? a.out : a.out + (size_t)z * a.M * a.N;
            const int ldo = direct ? a.ldo : a.N;
            const int colb = n0 + 128 * tile + 32 * wave + 4 * (lane & 7);
#pragma unroll
            for (int it = 0; it < ROWS / 8; ++it) {
                const int m = (lane >> 3) + 8 * it;
                f32x4 v = *reinterpret_cast<const f32x4*>(strip + m * 32 + 4 * (lane & 7));
                if (m < a.M && colb + 3 < a.N) {
                    if (direct && a.bias && !STAMPS) v += *reinterpret_cast<const f32x4*>(a.bias + colb);
                    *reinterpret_cast<f32x4*>(outp + (size_t)m * ldo + colb) = v;
                } else if (m < a.M) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (colb + e < a.N) outp[(size_t)m * ldo + colb + e] = v[e] + ((direct && a.bias) ? a.bias[colb + e] : 0.f);
                }
            }
            stamp(4 + 2 * i);
        }
    });
    stamp(31);
}

// shapes the resident-activation kernel takes: <= 64 rows, N a multiple of 4 and at least 2048 wide, whole 256-deep k ranges
bool gemm_resident_x3_fits(const GemmArgs& a) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("ICZ_GEMM_RESIDENT_X3"); on = e ? atoi(e) : 1; }
    if (!on || a.M <= 32 || a.M > 64 || a.N < 2048 || a.N % 4 || a.accumulate) return false;
    int tot = 0;
    for (int s = 0; s < a.nseg; ++s) {
        if (a.seg[s].K % 64 || a.seg[s].gather) return false;
        tot += a.seg[s].K / 64;
    }
    return tot % 4 == 0 && tot >= 8;
}
int gemm_resident_x3_nsplit(const GemmArgs& a) {
    int tot = 0;
    for (int s = 0; s < a.nseg; ++s) tot += a.seg[s].K / 64;
    return tot / 4;
}
static unsigned long long* g_sk_stamps = nullptr;       // development only (ICZ_SKINNY_ABL=4): 32 stamps for up to 4096 workgroups
int gemm_resident_x3(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (sk_env("ICZ_SKINNY_ABL", 0) == 4) {
        if (!g_sk_stamps) ICZ_CHECK_HIP(hipMalloc((void**)&g_sk_stamps, sizeof(unsigned long long) * 32 * 4096));
        a.bias = reinterpret_cast<const float*>(g_sk_stamps);
        constexpr size_t lds2 = rs_lds_bytes<4, 4, 2, 3>();
        static bool attr2 = false;
        if (!attr2) { ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_kernel<4, 4, 2, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2)); attr2 = true; }
        hipLaunchKernelGGL((gemm_resident_x3_kernel<4, 4, 2, 3, true>), dim3(cdiv(a.N, 256), 1, a.nsplit), dim3(256), lds2, stream, a);
        ICZ_CHECK_HIP(hipGetLastError());
        return ICZ_OK;
    }
    ICZ_REQUIRE(gemm_resident_x3_fits(a) && a.nsplit == gemm_resident_x3_nsplit(a) && a.chunks_per_split == 4,
                "gemm_resident_x3: launch does not match the kernel's fixed decomposition (nsplit %d)", a.nsplit);
    ICZ_REQUIRE(a.nsplit == 1 || a.ldo == a.N || true, "");
    static bool attr = false;
    constexpr size_t lds = rs_lds_bytes<4, 4, 2, 3>();
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_kernel<4, 4, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_resident_x3_kernel<4, 4, 2, 3>), dim3(cdiv(a.N, 256), 1, a.nsplit), dim3(256), lds, stream, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int gemm_predict(const float* x, int H, const float* w_pred, const float* bias, int rows, int V, int Vp, float* logits, int ldl,
                 float* ws, size_t ws_cap_floats, int* pred_nsplit, hipStream_t st) {
    static int pred_slabs = -1;
    if (pred_slabs < 0) pred_slabs = sk_env("ICZ_PREDICT_SLABS", 1);
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {x, w_pred, H, H, H, nullptr};
    g.M = rows; g.N = Vp; g.out = ws; g.ldo = Vp;
    if (pred_slabs && pred_nsplit && ws && gemm_resident_x3_fits(g) && gemm_slab_floats(rows, Vp, gemm_resident_x3_nsplit(g)) <= ws_cap_floats) {
        g.nsplit = gemm_resident_x3_nsplit(g);
        *pred_nsplit = g.nsplit;
    } else {           // K = H is short: no split-K, bias fused
        g.N = V; g.out = logits; g.ldo = ldl; g.bias = bias;
        g.nsplit = 1;
        if (pred_nsplit) *pred_nsplit = 1;
    }
    return gemm_f32(GEMM_NT, g, st);
}

// ------------------------------------------------------------------------------------------------

bool gemm_skinny_x3_enabled() {
    static int on = -1;
    if (on < 0) on = sk_env("ICZ_GEMM_SKINNY_X3", 0);       // opt-in: see the MEASURED paragraph of the header
    return on != 0;
}

// shapes this kernel takes: NT, 33..128 rows, whole 64-deep chunks, plain operands
bool gemm_skinny_x3_fits(const GemmArgs& a) {
    if (!gemm_skinny_x3_enabled()) return false;
    if (a.M <= 32 || a.M > 128 || a.N < 64) return false;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % SK_BK || a.seg[s].gather) return false;
    return true;
}

int gemm_skinny_x3_tile_n(const GemmArgs& a) {
    static int force = -1;
    if (force < 0) force = sk_env("ICZ_GEMM_SKINNY_NCT", 0);
    if (force == 1) return 64;
    if (force == 2) return 128;
    // 128 columns per workgroup (half the activation traffic per weight byte) where split-K still yields ~256 workgroups;
    // narrow outputs and the un-split vocabulary projection (N = 10102, K = 1024: 158 workgroups of 64 columns) take 64
    return (a.N >= 2048 && a.N <= 8192) ? 128 : 64;
}

template <int MT, int NCT, int D, int ABL, int NW = 4, bool XPL = false>
static int sk_launch(const GemmArgs& a, dim3 grid, hipStream_t stream) {
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_skinny_x3_kernel<MT, NCT, D, ABL, NW, XPL>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)sk_lds_bytes(MT)));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_skinny_x3_kernel<MT, NCT, D, ABL, NW, XPL>), grid, dim3(64 * NW), sk_lds_bytes(MT), stream, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int gemm_skinny_x3(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    if (sk_env("ICZ_SKINNY_ABL", 0) >= 4) {
        if (!g_sk_stamps) ICZ_CHECK_HIP(hipMalloc((void**)&g_sk_stamps, sizeof(unsigned long long) * 32 * 4096));
        a.bias = reinterpret_cast<const float*>(g_sk_stamps);
    }
    const int bn = gemm_skinny_x3_tile_n(a);
    const dim3 grid(cdiv(a.N, bn), 1, a.nsplit);
    static int abl = -1, depth = -1, nw = -1;
    if (abl < 0) { abl = sk_env("ICZ_SKINNY_ABL", 0); depth = sk_env("ICZ_SKINNY_D", 3); nw = sk_env("ICZ_SKINNY_NW", 8); }
    const bool tall = a.M > 64;
    bool xpl = true;
    for (int sg = 0; sg < a.nseg; ++sg) xpl = xpl && a.seg[sg].Apl != nullptr;
    if (xpl && abl == 0) {      // activations pre-split by their producers
        if (nw == 8) {
            if (tall) { if (bn == 128) return sk_launch<8, 2, 3, 0, 8, true>(a, grid, stream); else return sk_launch<8, 1, 3, 0, 8, true>(a, grid, stream); }
            else { if (bn == 128) return sk_launch<4, 2, 3, 0, 8, true>(a, grid, stream); else return sk_launch<4, 1, 3, 0, 8, true>(a, grid, stream); }
        }
        if (tall) { if (bn == 128) return sk_launch<8, 2, 3, 0, 4, true>(a, grid, stream); else return sk_launch<8, 1, 3, 0, 4, true>(a, grid, stream); }
        else { if (bn == 128) return sk_launch<4, 2, 3, 0, 4, true>(a, grid, stream); else return sk_launch<4, 1, 3, 0, 4, true>(a, grid, stream); }
    }
    if (xpl && abl == 4) return sk_launch<4, 2, 3, 4, 4, true>(a, grid, stream);
#define ICZ_SK(MT_, NCT_, D_, ABL_) return sk_launch<MT_, NCT_, D_, ABL_>(a, grid, stream)
    if (abl == 0 && nw == 8 && depth == 3) {
        {
            if (tall) { if (bn == 128) return sk_launch<8, 2, 3, 0, 8>(a, grid, stream); else return sk_launch<8, 1, 3, 0, 8>(a, grid, stream); }
            else { if (bn == 128) return sk_launch<4, 2, 3, 0, 8>(a, grid, stream); else return sk_launch<4, 1, 3, 0, 8>(a, grid, stream); }
        }
    }
    if (abl == 11) return sk_launch<4, 2, 3, 11, 4>(a, grid, stream);
    if (abl == 12) return sk_launch<4, 2, 3, 12, 4>(a, grid, stream);
    if (abl == 13) return sk_launch<4, 2, 3, 13, 4>(a, grid, stream);
    if (abl == 15) return sk_launch<4, 2, 3, 15, 4>(a, grid, stream);
    if (abl == 0 && depth == 3) {
        if (tall) { if (bn == 128) ICZ_SK(8, 2, 3, 0); else ICZ_SK(8, 1, 3, 0); }
        else { if (bn == 128) ICZ_SK(4, 2, 3, 0); else ICZ_SK(4, 1, 3, 0); }
    }
    // development variants (tools/perf_skinny_stamps.py): time stamps and ablations at 64 rows, 128-column tiles, 4 waves
    if (abl == 4) ICZ_SK(4, 2, 3, 4);
#undef ICZ_SK
    set_error("gemm_skinny_x3: no such variant (ICZ_SKINNY_ABL=%d ICZ_SKINNY_D=%d)", abl, depth);
    return ICZ_ERR_INVALID;
}

}  // namespace icz

// x -> its three bf16 piece planes (split3_planes.h layout), for tensors whose producer does not write them itself
__global__ __launch_bounds__(256) void split3_planes_kernel(const float* __restrict__ x, int rows, int K, int ld, icz::Planes pl) {
    const int row = blockIdx.y, k = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (k >= K) return;
    const icz::f32x4 v = *reinterpret_cast<const icz::f32x4*>(x + (size_t)row * ld + k);
    icz::sp_store4(pl, (size_t)row * ld, k, v[0], v[1], v[2], v[3]);
}
namespace icz {
int split3_planes(const float* x, int rows, int K, int ld, Planes pl, hipStream_t st) {
    ICZ_REQUIRE(x && pl.base && rows > 0 && K > 0 && K % 4 == 0 && ld % 4 == 0, "split3_planes: bad arguments");
    hipLaunchKernelGGL(split3_planes_kernel, dim3(cdiv(K, 1024), rows), dim3(256), 0, st, x, rows, K, ld, pl);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}
}  // namespace icz

// development only: the stamps of the last ICZ_SKINNY_ABL=4 launch (tools/perf_skinny_stamps.py)
extern "C" int icz_debug_skinny_stamps(unsigned long long* out_host, int n_workgroups) {
    if (!icz::g_sk_stamps || n_workgroups > 4096) return -1;
    return hipMemcpy(out_host, icz::g_sk_stamps, sizeof(unsigned long long) * 32 * n_workgroups, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
