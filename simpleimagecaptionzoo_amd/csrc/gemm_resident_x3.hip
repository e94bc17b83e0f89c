// Decoder-step NT GEMM with split-precision operands and LDS-resident activations: C[M x N] = X[M x K] W[N x K]^T for 33..64
// rows (the LSTM gates and the vocabulary projection of one decoder step; and, on transposed weight copies, the per-step dgrad
// GEMMs of BPTT).  gfx950 only.
//
// Roofline: the weight matrix is streamed from HBM exactly once per launch (N*K*4 bytes; the activations are a few hundred
// KB), so the kernel is HBM-bound as long as the matrix pipe keeps up.  On the fp32-input MFMA it does not (fp32 MFMA runs at
// the vector rate: 64 x 4096 x 4096 needs 13.7 us of it, the weights 8.4 us of HBM), so every fp32 operand is multiplied as
// three bf16 pieces, x = x0 + x1 + x2 (each the bf16 rounding of what the previous ones left: 24 mantissa bits, pieces exact),
// a product being the six piece products of order <= 2 accumulated in fp32 (v_mfma_f32_16x16x32_bf16; the dropped terms are
// below 3 * 2^-24 |x w|): 2.7x the fp32-MFMA rate at fp32-level error (tests: every kernel within 3e-6 of max|C| of float64).
//
// Structure (one 256-thread workgroup per CU, one wave per SIMD, all 512 registers per lane):
//   * a workgroup owns one k range of NSR 64-deep stages and TPW column tiles of 128.  The activations of its WHOLE k range are
//     split ONCE and stay in LDS as three bf16 planes; the workgroup then streams its column tiles one after the other through
//     them.  Activation bytes per weight byte: 0.25 (one 64 x 256 fp32 read per 2 x 128 x 256 weights);
//   * wave w owns 32 adjacent output columns of a tile and ALL rows; its weight rows go HBM -> registers directly (each weight
//     element is used by one wave only: an LDS round trip would buy nothing), as fragment-shaped loads that cover 64 contiguous
//     bytes of 16 rows per instruction, D - 1 pipeline steps ahead of their use.  The fp32 -> 3 x bf16 split of the weights is
//     VALU work in the shadow of the MFMAs of the previous k block;
//   * inside every 32-deep k block the k index is permuted (element e of lane quarter q is k = 4 q + e for e < 4 and
//     16 + 4 q + e - 4 above) so that a weight load instruction reads 64 CONTIGUOUS bytes per row; the activation planes are
//     written to LDS in the same order (row stride 64 NSR + 16 bf16: conflict-free 16-byte fragment reads);
//   * one barrier in the whole kernel (the planes are read-only afterwards); each wave's accumulators leave through its own
//     LDS strip as full 16-byte row segments when a tile is done;
//   * split-K slabs [z][M][N], summed in fixed order by the consumers (LSTM pointwise, argmax / multinomial, BPTT pointwise):
//     bitwise reproducible, no float atomics.
// Everything is compile-time (steps, ring slots, tile boundaries): no branch with a vector-memory operation in it.
// Launch: 1-D grid of (column groups) x (k ranges) workgroups, every one with exactly NSR stages (the launcher checks).
//
// Round 6 -- what ships at <= 64 rows wherever the stage count is a multiple of eight (every product of the BASELINE models): a workgroup
// owns a 512-deep k range (NSR = 8) and ONE column tile, i.e. the same 256 KB of weights as before, twice the activation block (128 KB,
// an L2 hit) and HALF the slab (32 KB): 6 / 8 / 2 split-K slabs instead of 12 / 16 / 4 for the TD gates, the LM gates and the vocabulary
// projection, on the same 192 / 256 / 158 workgroups.  The planes hold half a range (HS = 4 stages, 98 KB) at a time; the second half's
// activations are loaded into registers during the first half's steps (one float4 per thread every other MFMA group), wait there, and
// are cut into the same planes between two barriers in front of step 4; the accumulators stay live across the halves, one epilogue.
// MEASURED (profiles/r06_resident_k512_stamps.txt, median cycles per workgroup, LM gates 64 x 4096 x 4096): steps 3.5 k, 3 x 2.8 k,
// 5.2 k (the re-split: +3 k), 2.2 k, 1.9 k, 2.1 k + one epilogue 2.0 k = 35.6 k against 34.8 k for the 256-deep form -- the GEMM itself
// is 2 - 5 % SLOWER, and the step around it faster: the pointwise / select / BPTT kernels sum half the slabs and the launch edge
// behind the GEMM has half the dirty bytes to write back.  Same box, three alternations (profiles/r06_resident_k512_ab.log): SCST step
// 5.69 -> 5.50 ms, rollouts 2.79 -> 2.70, backward 2.49 -> 2.40.  (First version, the second half's loads in one burst behind the
// barrier: step 0 took 5.1 k cycles instead of 3.5 k, SCST step 5.79 -> 5.67 ms.)  The data-movement emulation had predicted the
// direction (tools/cxx/persistent_step.hip `half`: 72.9 -> 67.9 us per greedy step, profiles/r06_persistent_step_half_slabs.txt).
// ICZ_GEMM_RESIDENT_K512=0 selects the 256-deep form everywhere.
//
// MEASURED (MI355X, in-kernel clock stamps under -DICZ_DEV, tools/perf_skinny_stamps.py; rocprofv3 for whole launches):
//   * round 2: per workgroup ~9.5-11.5 k cycles from entry to the barrier, ~2.1-3.2 k per pipeline step (32 KB of weights),
//     ~1.7-2.1 k per tile epilogue; the main loop runs at the chip's HBM rate (256 KB per CU in ~19 k cycles), the fixed part
//     is a third of the launch.  Round 3 split the prologue: the 16 activation loads per thread take 2.9-4.2 k cycles TO ISSUE
//     and the 16 weight loads behind them another 4.0-5.4 k, the data is there ~200 cycles later, the split + LDS writes take
//     2.7 k: a compute unit's memory pipeline accepts ~14-16 bytes per cycle whatever the bytes are, so activation bytes cost
//     as much as weight bytes.  Tried in round 3 without gain: whole 128-byte lines per activation load instruction (kept, neutral),
//     XCD-aware placement of the workgroups that share an activation block (prologue unchanged, K = 4096 main loop slower: L2
//     channel conflicts), non-temporal weight loads (+-3 %), three stages per k range where that fills the chip (see "Decomposition"),
//     and staging the activations one 64-deep stage at a time beside the MFMAs of the first column tile (only 16 KB + the first
//     step's weights before the first MFMA: the barrier came 3.5 k cycles earlier, the first tile's steps took 4.4 k longer -- the
//     bytes a CU has to pull are the same, and ~4 k cycles pass between kernel entry and the 16th load instruction whatever is loaded);
//   * what did not help in round 2 (kept out): 8 waves, 256-column tiles (spills), bf16 planes written by the producers
//     (6 instead of 4 bytes per element through the same pipeline), weights through LDS.
#include <stdlib.h>

#include <type_traits>

#include "gemm_f32.h"

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
// the same instruction through the conversion builtin: one the machine scheduler knows as a VALU instruction.  Used where the cut stands
// alone (the activation planes: prologue and re-split, -0.4 k cycles each); inside the main loop the hand-placed scheduling groups were
// tuned around the INLINEASM nodes and the builtin makes steps 1 - 7 slower (stamps: profiles/r06_resident_k512_stamps.txt)
__device__ __forceinline__ uint32_t sk_cvt_pk_bf16_b(float lo, float hi) {
    typedef __bf16 sk_bf16x2_ __attribute__((ext_vector_type(2)));
    typedef float sk_f32x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((sk_f32x2_){lo, hi}, sk_bf16x2_));
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

__device__ __forceinline__ void sk_split3_b(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = sk_cvt_pk_bf16_b(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = sk_cvt_pk_bf16_b(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = sk_cvt_pk_bf16_b(ra, rb);
}

constexpr int SK_BK = 64;                 // k per pipeline stage (two MFMA k blocks of 32)
// bf16 per LDS row of the resident planes: 64 NSR + 16 -> a row stride of 8 (mod 16) dwords: the 16-byte fragment reads of the
// 16 rows x 4 quarters of a wave touch every bank once per 16-lane group
constexpr int rs_pb(int nsr) { return 64 * nsr + 16; }
template <int MT, int NSR>
constexpr size_t rs_lds_bytes() { return (size_t)3 * (16 * MT) * rs_pb(NSR) * 2 + (size_t)4 * (16 * MT) * 32 * 4; }

// one workgroup's work: pair = its index among the (k range, column group) pairs of the problem `a`
// HS = stages whose planes are resident at a time.  HS == NSR: the whole k range (rounds 2 - 5).  NSR == 2 HS, TPW == 1 (round 6): a
// 512-deep k range on ONE column tile, its planes in two halves -- the second half's activations are loaded into registers behind the
// barrier, wait there during the first half's steps and are cut into the same LDS planes between two barriers; the accumulators stay
// live across the halves.  Half the split-K slabs of the 256-deep decomposition for the same 256 KB of weights per workgroup.
template <int MT, int NSR, int TPW, int D, bool STAMPS, int HS = NSR>
__device__ __forceinline__ void resident_x3_body(const GemmArgs& a, const int pair) {
    static_assert(HS >= 2 && HS <= 4 && (NSR == HS || (NSR == 2 * HS && TPW == 1)), "two to four resident 64-deep stages; two halves only on one tile");
    constexpr int RS_PB = rs_pb(HS);
    unsigned long long* const stamps = STAMPS ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.bias)) + 32 * (size_t)blockIdx.x : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (STAMPS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0 && i < 32) stamps[i] = t;
        }
    };
    stamp(0);
    const int lflag = live_flag(a.live);
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    unsigned short* const planes = reinterpret_cast<unsigned short*>(sk_smem);
    constexpr int ROWS = 16 * MT, NCT = 2, NT = TPW * NSR;
    constexpr size_t PLANE = (size_t)ROWS * RS_PB;
    constexpr int XL = ROWS * 16 / 256;                        // activation float4 per thread and 64-deep stage
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    // (column group, k range) = (b % ncg, b / ncg): workgroups b and b + 8 share an XCD (round-robin dispatch), so an XCD's L2 sees
    // every k range.  Measured round 3: giving each XCD whole k ranges instead (its L2 then fetches an activation block once, not
    // all of them) left the prologue unchanged and made the K = 4096 main loop 1.5 - 2x slower per step: with a power-of-two row
    // stride an XCD that streams only two k ranges uses a fraction of its L2 channels.
    const int ncg = (a.N + 128 * TPW - 1) / (128 * TPW);
    const int z = pair / ncg, n0 = (pair % ncg) * (128 * TPW);
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
        // a wave-wide load covers four rows x 256 contiguous bytes (whole 128-byte lines: lanes 0..15 = the 16 float4 of a row's stage)
        f32x4 xr[HS][XL];
        sk_static_for<0, HS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            const Cur c = seek(c_begin + st);
            const GemmSeg& g = a.seg[c.seg];
            sk_static_for<0, XL>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int f = tid + 256 * j, row = f >> 4, c4 = f & 15;
                xr[st][j] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(row < a.M ? row : a.M - 1) * g.lda + c.k0 + 4 * c4);
            });
        });
        if constexpr (STAMPS) { __builtin_amdgcn_sched_barrier(0); stamp(20); __builtin_amdgcn_sched_barrier(0); }
        // the first step's weights go out now: their HBM latency passes behind the split below.  The second step's follow AFTER the
        // split (round 3: with both up front, 128 KB of loads stood between kernel entry and the first MFMA; deferring 32 KB of them
        // took ~0.5 us off every launch: greedy step 96.9 -> 95.4 us, SCST rollouts and backward -1 % each, same box)
        point_w(std::integral_constant<int, 0>{});
        sk_static_for<0, LW>([&](auto lc) { load_w1(std::integral_constant<int, 0>{}, lc); });
        __builtin_amdgcn_sched_barrier(0);
        // early-out of a rollout / BPTT step behind the reference's break, behind the issue of the activation and first weight loads
        // (live_flag / flag_dead, icz_common.h): nothing has been written yet
        if (flag_dead(lflag)) return;
        if constexpr (STAMPS) {
            stamp(21);
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            stamp(22);
            __builtin_amdgcn_sched_barrier(0);
        }
        sk_static_for<0, HS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            sk_static_for<0, XL>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int f = tid + 256 * j, row = f >> 4, c4 = f & 15, kq = c4 & 7;
                // float4 kq of a 32-block holds k = 4 kq .. 4 kq + 3: fragment position 8 (kq & 3) + 4 (kq >> 2) (see the header)
                unsigned short* o = planes + (size_t)row * RS_PB + 64 * st + 32 * (c4 >> 3) + 8 * (kq & 3) + 4 * (kq >> 2);
                uint32_t a0, a1, a2, b0, b1, b2;
                sk_split3_b(xr[st][j][0], xr[st][j][1], a0, a1, a2);
                sk_split3_b(xr[st][j][2], xr[st][j][3], b0, b1, b2);
                *reinterpret_cast<sk_u32x2*>(o) = (sk_u32x2){a0, b0};
                *reinterpret_cast<sk_u32x2*>(o + PLANE) = (sk_u32x2){a1, b1};
                *reinterpret_cast<sk_u32x2*>(o + 2 * PLANE) = (sk_u32x2){a2, b2};
            });
        });
    }
    // second half of a 512-deep range: its activations wait in registers until the first half's steps are done with the planes
    f32x4 xr2[HS][XL];

    __builtin_amdgcn_sched_barrier(0);
    sk_static_for<1, D - 1>([&](auto ic) {
        point_w(ic);
        sk_static_for<0, LW>([&](auto lc) { load_w1(ic, lc); });
    });
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    __syncthreads();                        // the planes are complete (HS == NSR: the only barrier of the kernel)
    stamp(2);
    // (the second half's activation loads ride inside steps 0 .. HS - 1, one float4 per thread every other MFMA group: issued in one
    // burst here -- 64 KB per workgroup through a memory pipe that takes ~16 B per cycle -- they made step 0 take 5.1 k cycles instead
    // of 3.2 k, in-kernel stamps round 6)
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
        if constexpr (NSR != HS && i == HS) {
            // ---- the first half is done with the planes: cut the second half into them (two barriers; the weight ring runs on)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            sk_static_for<0, HS>([&](auto sc) {
                constexpr int s2 = decltype(sc)::value;
                sk_static_for<0, XL>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int f = tid + 256 * j, row = f >> 4, c4 = f & 15, kq = c4 & 7;
                    unsigned short* o = planes + (size_t)row * RS_PB + 64 * s2 + 32 * (c4 >> 3) + 8 * (kq & 3) + 4 * (kq >> 2);
                    uint32_t a0, a1, a2, b0, b1, b2;
                    sk_split3_b(xr2[s2][j][0], xr2[s2][j][1], a0, a1, a2);
                    sk_split3_b(xr2[s2][j][2], xr2[s2][j][3], b0, b1, b2);
                    *reinterpret_cast<sk_u32x2*>(o) = (sk_u32x2){a0, b0};
                    *reinterpret_cast<sk_u32x2*>(o + PLANE) = (sk_u32x2){a1, b1};
                    *reinterpret_cast<sk_u32x2*>(o + 2 * PLANE) = (sk_u32x2){a2, b2};
                });
            });
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (HAS_LOAD) point_w(std::integral_constant<int, i + D - 1>{});
        constexpr bool XLOAD = NSR != HS && i < HS;          // this step carries the loads of stage HS + i (second half)
        const float* xbase = nullptr;
        int xlda = 0;
        if constexpr (XLOAD) {
            const Cur c2 = seek(c_begin + HS + i);
            xbase = a.seg[c2.seg].A + c2.k0;
            xlda = a.seg[c2.seg].lda;
        }
        const unsigned short* abase = planes + (size_t)li * RS_PB + 64 * (st % HS) + 8 * lq;
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
            constexpr bool XL1 = XLOAD && g % (G / XL) == 0;
            if constexpr (XL1) {
                constexpr int j = g / (G / XL);
                const int f = tid + 256 * j, row = f >> 4, c4 = f & 15;
                xr2[i % HS][j] = *reinterpret_cast<const f32x4*>(xbase + (size_t)(row < a.M ? row : a.M - 1) * xlda + 4 * c4);
            }
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
            if constexpr (HAS_LOAD || XL1) __builtin_amdgcn_sched_group_barrier(0x020, (HAS_LOAD ? (g + 1) * LW / G - g * LW / G : 0) + (XL1 ? 1 : 0), 0);
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

template <int MT, int NSR, int TPW, int D, bool STAMPS = false, int HS = NSR>
__global__ __launch_bounds__(256) void gemm_resident_x3_kernel(GemmArgs a) {
    resident_x3_body<MT, NSR, TPW, D, STAMPS, HS>(a, blockIdx.x);
}
// Two independent problems in one launch (the two recurrent dgrad products of a BPTT step, d h2 and d h1: 128 + 64 workgroups --
// alone neither fills the chip, and every launch pays the same fixed prologue): workgroups [0, first) take `a`, the rest `b`.
struct GemmPair { GemmArgs a, b; int first; };
template <int MT, int NSR, int TPW, int D, int HS = NSR>
__global__ __launch_bounds__(256) void gemm_resident_x3_pair_kernel(GemmPair g) {
    if ((int)blockIdx.x < g.first) resident_x3_body<MT, NSR, TPW, D, false, HS>(g.a, blockIdx.x);
    else resident_x3_body<MT, NSR, TPW, D, false, HS>(g.b, blockIdx.x - g.first);
}

// ------------------------------------------------------------------------------------------------
// 65..128 rows (round 4): the greedy baseline and the sampled rollout of an SCST step as ONE chain of 2 x 64 decoder rows, so that
// the weights are streamed once per step pair instead of once per chain.  Same decomposition as above (a workgroup owns a 256-deep
// k range x two 128-column tiles = 256 KB of weights, 256 / 192 / 160 workgroups), same fragment-shaped weight loads, same slabs.
// What changes with 128 rows: the three bf16 planes of a 256-deep range would need 196 KB of LDS, so the planes hold HALF a range
// (128 deep: 108 KB) at a time and BOTH column tiles keep their accumulators live (128 VGPRs): pipeline steps run (half, tile,
// stage) = (0,0,0) (0,0,1) (0,1,0) (0,1,1) | re-split | (1,0,2) (1,0,3) (1,1,2) (1,1,3).  The second half's activations are loaded
// into registers beside the first two steps' MFMAs and wait there; the re-split costs two barriers.  The MFMA operands are swapped
// (A = weights, B = activations) so that a lane holds four consecutive columns of a row and a tile leaves straight from the
// accumulators as 16-byte stores, tile 0 beside tile 1's last two steps.
// Bytes through a compute unit per workgroup: 256 KB weights + 128 KB activations + 128 KB slab tile = 512 KB for 128 rows against
// 2 x 384 KB for two 64-row launches; matrix time 8 steps x 192 MFMAs x 16 cycles = 24.6 k cycles per wave -- at 128 rows the main
// loop is paced by the matrix pipe (the weight split of a k block, ~90 VALU instructions, now hides behind 96 MFMAs instead of 48).
// pipeline step i of the 128-row kernel -> column tile, stage of the k range (two tiles x two stages per resident half)
constexpr int m128_tile_of(int i) { return (i % 4) / 2; }
constexpr int m128_stage_of(int i) { return (i / 4) * 2 + i % 2; }
template <int D, bool STAMPS>
__device__ __forceinline__ void resident_x3_m128_body(const GemmArgs& a, const int pair) {
    constexpr int MT = 8, NSR = 4, HS = 2, TPW = 2, NCT = 2, ROWS = 16 * MT, NT = TPW * NSR;
    constexpr int RS_PB = rs_pb(HS);
    constexpr size_t PLANE = (size_t)ROWS * RS_PB;
    constexpr int XL = ROWS * 16 / 256;                        // activation float4 per thread and 64-deep stage
    unsigned long long* const stamps = STAMPS ? reinterpret_cast<unsigned long long*>(const_cast<float*>(a.bias)) + 32 * (size_t)blockIdx.x : nullptr;
    auto stamp = [&](int i) __attribute__((always_inline)) {
        if constexpr (STAMPS) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            if (threadIdx.x == 0 && i < 32) stamps[i] = t;
        }
    };
    stamp(0);
    if (step_dead(a.live)) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    unsigned short* const planes = reinterpret_cast<unsigned short*>(sk_smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lq = lane >> 4;
    const int ncg = (a.N + 128 * TPW - 1) / (128 * TPW);
    const int z = pair / ncg, n0 = (pair % ncg) * (128 * TPW);
    const int c_begin = z * NSR;

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

    f32x4 acc[MT][TPW * NCT];
    f32x4 w[D][2][NCT][2];
    sk_u32x4 bq[2][NCT][3];
    const float* wp[NCT];
    auto point_w = [&](auto stepc) __attribute__((always_inline)) {
        constexpr int i = decltype(stepc)::value;
        const Cur c = seek(c_begin + m128_stage_of(i));
        const GemmSeg& g = a.seg[c.seg];
#pragma unroll
        for (int cc = 0; cc < NCT; ++cc) {
            const int col = n0 + 128 * m128_tile_of(i) + (wave * NCT + cc) * 16 + li;
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
    // activations of one half (HS stages from `first`): global -> registers, registers -> the three planes
    auto load_x = [&](f32x4 (&xr)[HS][XL], int first) __attribute__((always_inline)) {
        sk_static_for<0, HS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            const Cur c = seek(c_begin + first + st);
            const GemmSeg& g = a.seg[c.seg];
            sk_static_for<0, XL>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int f = tid + 256 * j, row = f >> 4, c4 = f & 15;
                xr[st][j] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(row < a.M ? row : a.M - 1) * g.lda + c.k0 + 4 * c4);
            });
        });
    };
    auto load_x1 = [&](f32x4 (&xr)[HS][XL], int first, auto idx) __attribute__((always_inline)) {      // one float4 of load_x
        constexpr int st = decltype(idx)::value / XL, j = decltype(idx)::value % XL;
        const Cur c = seek(c_begin + first + st);
        const GemmSeg& g = a.seg[c.seg];
        const int f = tid + 256 * j, row = f >> 4, c4 = f & 15;
        xr[st][j] = *reinterpret_cast<const f32x4*>(g.A + (size_t)(row < a.M ? row : a.M - 1) * g.lda + c.k0 + 4 * c4);
    };
    auto split_x = [&](f32x4 (&xr)[HS][XL]) __attribute__((always_inline)) {
        sk_static_for<0, HS>([&](auto sc) {
            constexpr int st = decltype(sc)::value;
            sk_static_for<0, XL>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const int f = tid + 256 * j, row = f >> 4, c4 = f & 15, kq = c4 & 7;
                unsigned short* o = planes + (size_t)row * RS_PB + 64 * st + 32 * (c4 >> 3) + 8 * (kq & 3) + 4 * (kq >> 2);
                uint32_t a0, a1, a2, b0, b1, b2;
                sk_split3(xr[st][j][0], xr[st][j][1], a0, a1, a2);
                sk_split3(xr[st][j][2], xr[st][j][3], b0, b1, b2);
                *reinterpret_cast<sk_u32x2*>(o) = (sk_u32x2){a0, b0};
                *reinterpret_cast<sk_u32x2*>(o + PLANE) = (sk_u32x2){a1, b1};
                *reinterpret_cast<sk_u32x2*>(o + 2 * PLANE) = (sk_u32x2){a2, b2};
            });
        });
    };

    // the second half's activations: loaded beside the MFMAs of steps 0 and 1 (one float4 per even row-tile group; the weight loads
    // take the odd ones), they wait in registers until the re-split.  (First version: all 16 loads in front of the barrier --
    // entry -> barrier 12.2 k cycles at K = 4096 with 192 KB of loads queued in front of the first MFMA.)
    f32x4 xr2[HS][XL];
    {
        f32x4 xr[HS][XL];
        load_x(xr, 0);
        point_w(std::integral_constant<int, 0>{});
        sk_static_for<0, LW>([&](auto lc) { load_w1(std::integral_constant<int, 0>{}, lc); });
        __builtin_amdgcn_sched_barrier(0);
        split_x(xr);
    }
    __builtin_amdgcn_sched_barrier(0);
    sk_static_for<1, D - 1>([&](auto ic) {
        point_w(ic);
        sk_static_for<0, LW>([&](auto lc) { load_w1(ic, lc); });
    });
    __builtin_amdgcn_sched_barrier(0);
    stamp(1);
    __syncthreads();
    stamp(2);
    sk_static_for<0, WP>([&](auto pc) { w_piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 0>{}); });
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int c = 0; c < TPW * NCT; ++c) acc[t][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool direct = a.nsplit == 1;
    float* const outp = direct ? a.out : a.out + (size_t)z * a.M * a.N;
    const int ldo = direct ? a.ldo : a.N;
    auto store_tile = [&](auto tc) __attribute__((always_inline)) {
        constexpr int tile = decltype(tc)::value;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const int colb = n0 + 128 * tile + (wave * NCT + c) * 16 + 4 * lq;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const int m = 16 * t + li;
                f32x4 v = acc[t][tile * NCT + c];
                if (m < a.M && colb + 3 < a.N) {
                    if (direct && a.bias && !STAMPS) v += *reinterpret_cast<const f32x4*>(a.bias + colb);
                    *reinterpret_cast<f32x4*>(outp + (size_t)m * ldo + colb) = v;
                } else if (m < a.M) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (colb + e < a.N) outp[(size_t)m * ldo + colb + e] = v[e] + ((direct && a.bias) ? a.bias[colb + e] : 0.f);
                }
            }
        }
    };

    sk_static_for<0, NT>([&](auto stepc) {
        constexpr int i = decltype(stepc)::value, S = i % D, tile = m128_tile_of(i), hst = i % HS;
        constexpr bool HAS_LOAD = i + D - 1 < NT, HAS_NEXT = i + 1 < NT;
        if constexpr (i == TPW * HS) {
            // ---- the first half of the k range is done for both tiles: the planes take the second half
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            split_x(xr2);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            stamp(19);
        }
        if constexpr (HAS_LOAD) point_w(std::integral_constant<int, i + D - 1>{});
        const unsigned short* abase = planes + (size_t)li * RS_PB + 64 * hst + 8 * lq;
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
            constexpr bool X2_LOAD = i < 2 && g % 2 == 0;       // 2 steps x 8 even groups = the HS * XL = 16 float4 of the second half
            if constexpr (HAS_LOAD)
                sk_static_for<g * LW / G, (g + 1) * LW / G>([&](auto lc) { load_w1(std::integral_constant<int, (S + D - 1) % D>{}, lc); });
            if constexpr (X2_LOAD) load_x1(xr2, HS, std::integral_constant<int, i * (G / 2) + g / 2>{});
            if constexpr (b == 0 || HAS_NEXT) {
                sk_static_for<t * WP / MT, (t + 1) * WP / MT>([&](auto pc) {
                    if constexpr (b == 0) w_piece(std::integral_constant<int, S>{}, std::integral_constant<int, 1>{}, pc, std::integral_constant<int, 1>{});
                    else w_piece(std::integral_constant<int, (S + 1) % D>{}, std::integral_constant<int, 0>{}, pc, std::integral_constant<int, 1>{});
                });
            }
            // operands swapped against the 64-row kernel: A = weight fragment (16 output columns), B = activation fragment (16 rows),
            // so a lane ends up with FOUR CONSECUTIVE COLUMNS of one row (D[n = 4 lq + r][m = li]) and the tile leaves as 16-byte stores
            // straight from the accumulators -- no LDS strip, so tile 0 can leave while tile 1 still multiplies
            sk_static_for<0, NCT>([&](auto cc) {       // smallest terms first
                constexpr int c = decltype(cc)::value;
                const sk_bf16x8 b0 = __builtin_bit_cast(sk_bf16x8, bq[0][c][0]), b1_ = __builtin_bit_cast(sk_bf16x8, bq[0][c][1]),
                                b2 = __builtin_bit_cast(sk_bf16x8, bq[0][c][2]);
                f32x4 v = acc[t][tile * NCT + c];
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g & 1][2], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2, af[g & 1][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1_, af[g & 1][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g & 1][1], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1_, af[g & 1][0], v, 0, 0, 0);
                v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, af[g & 1][0], v, 0, 0, 0);
                acc[t][tile * NCT + c] = v;
            });
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            if constexpr (HAS_LOAD || X2_LOAD)
                __builtin_amdgcn_sched_group_barrier(0x020, (HAS_LOAD ? (g + 1) * LW / G - g * LW / G : 0) + (X2_LOAD ? 1 : 0), 0);
#pragma unroll
            for (int k = 0; k < 6 * NCT; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            }
            if constexpr (t == MT - 1) {
                __builtin_amdgcn_sched_barrier(0);
                sk_static_for<0, NCT * 3>([&](auto q) { bq[0][decltype(q)::value / 3][decltype(q)::value % 3] = bq[1][decltype(q)::value / 3][decltype(q)::value % 3]; });
            }
        });
        stamp(3 + i);
        if constexpr (i >= TPW * HS && i % HS == HS - 1) {
            // second half, the tile's last stage: its accumulators leave now; tile 0's stores run beside tile 1's last two steps
            store_tile(std::integral_constant<int, tile>{});
            stamp(12 + tile);
        }
    });
    stamp(31);
}

template <int D, bool STAMPS = false>
__global__ __launch_bounds__(256) void gemm_resident_x3_m128_kernel(GemmArgs a) {
    resident_x3_m128_body<D, STAMPS>(a, blockIdx.x);
}
constexpr size_t rs_m128_lds_bytes() { return (size_t)3 * 128 * rs_pb(2) * 2; }

// ------------------------------------------------------------------------------------------------
// Decomposition.  tot = K / 64 stages in all; a workgroup takes NSR = 4 of them (a 256-deep k range) and TPW = 2 column tiles of
// 128: TD gates (K = 3072) 12 ranges x 16 column groups = 192 workgroups, LM gates (K = 4096) 256, vocabulary projection 160.
// MEASURED (round 3, same box): three-stage ranges for the TD gates (256 workgroups of 192 KB of weights instead of 192 of 256 KB)
// take 17 % fewer cycles per workgroup (33.3 k -> 27.6 k) and make the decode SLOWER: greedy 64 x 20 steps 98.9 -> 100.8 us per
// step, SCST rollouts 3.09 -> 3.13 ms -- four more slabs through the LSTM pointwise kernel, and no idle CUs left for the other
// chain of the rollout pair.  Not kept; the kernel body stays generic in NSR.  The other direction -- TPW = 4 column tiles per
// workgroup (half the workgroups: 96 / 128 / 80, every launch on at most half of the CUs, the activation block amortised over twice
// the weights) -- is worse by more: rollouts 2.75 -> 3.25 ms, greedy 95.5 -> 127 us per step with all three GEMMs on it, 2.98 ms /
// 109 us with the vocabulary projection alone (same box, bitwise the same results): two half-chip GEMMs of the two chains side by
// side do not make up for a launch that takes 1.7x as long.  And TWO workgroups per CU (so that one chain's GEMM can stream while the
// other's is in its prologue / epilogue): NSR = 2 x TPW = 4 -- 128-deep ranges of 512 columns, the same 256 KB of weights per
// workgroup, 55 KB of planes + half-width epilogue strips = 72 KB of LDS, 200 VGPRs -- doubles the slabs (24 - 32 per gate GEMM)
// and loses everywhere: rollouts 2.79 -> 3.55 ms, greedy 96 -> 110 us per step (same box, suite green with it).
static int rs_total_stages(const GemmArgs& a) {
    int tot = 0;
    for (int s = 0; s < a.nseg; ++s) tot += a.seg[s].K / SK_BK;
    return tot;
}
// (round 6) <= 64 rows, whole 512-deep ranges: eight stages on ONE column tile (192 / 256 / 158 workgroups for the TD gates, the LM gates
// and the vocabulary projection -- as before -- with half the split-K slabs: 6 / 8 / 2)
int gemm_resident_x3_stages(const GemmArgs& a) {
    const int tot = rs_total_stages(a);
    if (gemm_switches().resident_k512 && a.M <= 64 && tot % 8 == 0) return 8;
    return tot % 4 == 0 ? 4 : 0;
}
// shapes the kernel can take: 33..64 rows, N a multiple of 4, whole 64-deep stages that split into ranges
static bool rs_shape_ok(const GemmArgs& a) {
    if (!gemm_switches().resident_x3 || a.M <= 32 || a.M > (gemm_switches().resident_m128 ? 128 : 64) || a.N % 4 || a.accumulate) return false;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].K % SK_BK || a.seg[s].gather) return false;
    return rs_total_stages(a) >= 8 && gemm_resident_x3_stages(a) > 0;
}
// ... and the ones gemm_f32 routes to it by itself: at least 2048 columns (narrower outputs leave most of the chip idle at 256
// columns per workgroup: dec_att stays on the fp32 kernel)
bool gemm_resident_x3_fits(const GemmArgs& a) { return a.N >= 2048 && rs_shape_ok(a); }
int gemm_resident_x3_nsplit(const GemmArgs& a) { return rs_total_stages(a) / gemm_resident_x3_stages(a); }

template <int NSR, bool STAMPS>
static int rs_launch(const GemmArgs& a, hipStream_t stream) {
    constexpr size_t lds = rs_lds_bytes<4, NSR>();
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_kernel<4, NSR, 2, 3, STAMPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_resident_x3_kernel<4, NSR, 2, 3, STAMPS>), dim3(cdiv(a.N, 256) * a.nsplit), dim3(256), lds, stream, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

#ifndef RS_K512_D
#define RS_K512_D 3          // depth of the weight ring of the 512-deep form (pipeline steps in flight + 1); 4 measured: +5 - 8 % cycles per workgroup (a longer prologue), profiles/r06_resident_k512_stamps.txt
#endif
template <bool STAMPS>
static int rs_launch_k512(const GemmArgs& a, hipStream_t stream) {
    constexpr size_t lds = rs_lds_bytes<4, 4>();
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_kernel<4, 8, 1, RS_K512_D, STAMPS, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_resident_x3_kernel<4, 8, 1, RS_K512_D, STAMPS, 4>), dim3(cdiv(a.N, 128) * a.nsplit), dim3(256), lds, stream, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

template <bool STAMPS>
static int rs_launch_m128(const GemmArgs& a, hipStream_t stream) {
    constexpr size_t lds = rs_m128_lds_bytes();
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_m128_kernel<3, STAMPS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    hipLaunchKernelGGL((gemm_resident_x3_m128_kernel<3, STAMPS>), dim3(cdiv(a.N, 256) * a.nsplit), dim3(256), lds, stream, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

#ifdef ICZ_DEV
static unsigned long long* g_sk_stamps = nullptr;       // development builds: 32 stamps for up to 4096 workgroups
static int rs_dev_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#endif

int gemm_resident_x3(const GemmArgs& a_in, hipStream_t stream) {
    GemmArgs a = a_in;
    const int nsr = gemm_resident_x3_stages(a);
    ICZ_REQUIRE(rs_shape_ok(a) && a.nsplit == gemm_resident_x3_nsplit(a) && a.chunks_per_split == nsr,
                "gemm_resident_x3: launch does not match the kernel's fixed decomposition (nsplit %d)", a.nsplit);
    ICZ_REQUIRE(a.nsplit > 1 || a.ldo >= a.N, "gemm_resident_x3: output row stride %d below N = %d", a.ldo, a.N);
#ifdef ICZ_DEV
    if (rs_dev_env("ICZ_DEV_STAMPS")) {
        if (!g_sk_stamps) ICZ_CHECK_HIP(hipMalloc((void**)&g_sk_stamps, sizeof(unsigned long long) * 32 * 4096));
        a.bias = reinterpret_cast<const float*>(g_sk_stamps);
        return a.M > 64 ? rs_launch_m128<true>(a, stream) : (nsr == 8 ? rs_launch_k512<true>(a, stream) : rs_launch<4, true>(a, stream));
    }
#endif
    return a.M > 64 ? rs_launch_m128<false>(a, stream) : (nsr == 8 ? rs_launch_k512<false>(a, stream) : rs_launch<4, false>(a, stream));
}

// the pair launch: both problems in the kernel's decomposition with four stages per k range, slabs out (nsplit > 1)
bool gemm_resident_x3_pair_fits(const GemmArgs& a, const GemmArgs& b) {
    return a.M <= 64 && b.M <= 64 && rs_shape_ok(a) && rs_shape_ok(b) && gemm_resident_x3_stages(a) == gemm_resident_x3_stages(b) && a.N >= 512 && b.N >= 512;
}
int gemm_resident_x3_pair(const GemmArgs& a_in, const GemmArgs& b_in, hipStream_t stream) {
    ICZ_REQUIRE(gemm_resident_x3_pair_fits(a_in, b_in), "gemm_resident_x3_pair: shapes outside the kernel's decomposition");
    GemmPair g;
    g.a = a_in; g.b = b_in;
    const int nsr = gemm_resident_x3_stages(g.a), tilew = nsr == 8 ? 128 : 256;
    for (GemmArgs* p : {&g.a, &g.b}) {
        p->nsplit = gemm_resident_x3_nsplit(*p);
        p->chunks_per_split = nsr;
        p->bias = nullptr;
        ICZ_REQUIRE(p->out && p->nsplit > 1, "gemm_resident_x3_pair: slab output expected");
    }
    g.first = cdiv(g.a.N, tilew) * g.a.nsplit;
    constexpr size_t lds = rs_lds_bytes<4, 4>();
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_pair_kernel<4, 4, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_resident_x3_pair_kernel<4, 8, 1, RS_K512_D, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const dim3 grid(g.first + cdiv(g.b.N, tilew) * g.b.nsplit);
    if (nsr == 8) hipLaunchKernelGGL((gemm_resident_x3_pair_kernel<4, 8, 1, RS_K512_D, 4>), grid, dim3(256), lds, stream, g);
    else hipLaunchKernelGGL((gemm_resident_x3_pair_kernel<4, 4, 2, 3>), grid, dim3(256), lds, stream, g);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int gemm_predict(const float* x, int H, const float* w_pred, const float* bias, int rows, int V, int Vp, float* logits, int ldl,
                 float* ws, size_t ws_cap_floats, int* pred_nsplit, hipStream_t st, const int* live) {
    GemmArgs g = {};
    g.nseg = 1;
    g.live = live;
    g.seg[0] = {x, w_pred, H, H, H, nullptr};
    g.M = rows; g.N = Vp; g.out = ws; g.ldo = Vp;
    // (one k range only -- K = 512 on 512-deep ranges, NIC's vocabulary projection -- is not a slab product: finished logits, bias fused)
    if (gemm_switches().predict_slabs && pred_nsplit && ws && gemm_resident_x3_fits(g) && gemm_resident_x3_nsplit(g) > 1 &&
        gemm_slab_floats(rows, Vp, gemm_resident_x3_nsplit(g)) <= ws_cap_floats) {
        g.nsplit = gemm_resident_x3_nsplit(g);
        *pred_nsplit = g.nsplit;
    } else {           // K = H is short: no split-K, bias fused
        g.N = V; g.out = logits; g.ldo = ldl; g.bias = bias;
        g.nsplit = 1;
        if (pred_nsplit) *pred_nsplit = 1;
    }
    return gemm_f32(GEMM_NT, g, st);
}

}  // namespace icz

#ifdef ICZ_DEV
// development builds only: the stamps of the last ICZ_DEV_STAMPS=1 launch (tools/perf_skinny_stamps.py)
extern "C" int icz_debug_skinny_stamps(unsigned long long* out_host, int n_workgroups) {
    if (!icz::g_sk_stamps || n_workgroups > 4096) return -1;
    return hipMemcpy(out_host, icz::g_sk_stamps, sizeof(unsigned long long) * 32 * n_workgroups, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
