// EXPERIMENT OF ROUND 3 -- NOT PART OF libicz.so.  Measured and rejected: the GEMM alone is 1.1 - 1.3x
// faster than gemm_tn128_x3_kernel (4096 x 4096 x 1280: 206 against 265 us = 1.25 PFLOP/s of bf16 MFMA, 50 % of the dense peak;
// 640 x 4096 x 4096 with 5 splits: ~114 against ~130 us), but the packing pass (11.6 us per 4096 x 1280 activation operand, 29 - 40 us
// per 4096 x 4096 weight matrix) takes most of that back wherever the operand is not static: net -13 % at 4096 x 4096 x 1280, +-0 to
// +25 % at the 640- and 2304-row shapes (tools/perf_planes.py, gpurun_out/r3/e7_*.log of the session).
// To rerun: copy this file and its header into simpleimagecaptionzoo_amd/csrc/ (as gemm_planes.hip / .h), add it to SRCS, and route
// icz_gemm_f32 (butd.hip) through planes_pack x 2 + gemm_planes when ICZ_DEV_PLANES=<config> is set, workspace = planes_bytes(M, K) +
// planes_bytes(N, K); split-K slabs go through slab_reduce_kernel as for the other kernels.
//
// Split-precision GEMM on PRE-SPLIT operands ("planes"): the many-row GEMMs of the path (weight gradients and batched dgrad
// over all time steps, beam-search steps at 640 rows, the AoA refiner at 2304 rows, XE forward over all time steps).
//
// gemm_tn128_x3_kernel (gemm_f32.hip) loads fp32 tiles, cuts every element into three bf16 pieces in the CU and writes them to
// LDS -- once per TILE that uses the element (32 times for a 4096-wide output) -- in a 128 x 128, two-barriers-per-stage loop that
// reaches 39 % of the bf16 matrix peak.  Removing the VALU split alone bought 4 - 10 % (round 3, `ICZ_DEV_NOSPLIT` experiment):
// the loop structure is the limit, not the split.  Here the operands are cut ONCE by a packing pass into bf16 planes laid out in
// global memory exactly as the GEMM wants them in LDS, so the GEMM's staging is a plain 1-KiB-per-wave-instruction LDS-DMA copy
// (`global_load_lds_dwordx4`: no VGPRs, no VALU, no ds_write), kept in flight across the loop's one barrier per k-step with
// counted `s_waitcnt vmcnt` waits.
//
// Plane layout of an operand with R rows (padded to Rp, a multiple of 256) and KS k-steps of 16:
//     plane p (0 = leading piece), k-step s, row r, k half h (k = 16 s + 8 h .. + 7)  ->  16 bytes at
//     (((p KS + s) Rp + (r & ~31)) 32 + chunk(r & 31, h) 16,      chunk(r, h) = 2 r + (h ^ ((r >> 3) & 1))
// i.e. one k-step of 32 consecutive rows is 1 KiB, and inside it the two halves of rows 8..15 / 24..31 are swapped so that the
// 16 lanes of a ds_read_b128 group (rows 0..15 of one half) cover all 64 LDS banks.  Pad rows and pad k are zero.
// Both operands are K-contiguous planes (A: rows = M, B: rows = N), whatever the layout of the fp32 source (the packing pass
// transposes k-major sources); several K segments are simply consecutive k-steps.
//
// Arithmetic = gemm_tn128_x3_kernel's: per 16-deep k block the six piece products a2 b0, a0 b2, a1 b1, a1 b0, a0 b1, a0 b0 in that
// order into one fp32 accumulator (v_mfma_f32_32x32x16_bf16).
#include "gemm_planes_experiment.h"

namespace icz {

typedef __attribute__((ext_vector_type(8))) __bf16 pl_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pl_f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t pl_u32x4;

__device__ __forceinline__ uint32_t pl_cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void pl_split3(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = pl_cvt_pk_bf16(a, b);
    float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);
    p1 = pl_cvt_pk_bf16(ra, rb);
    ra -= __uint_as_float(p1 << 16);
    rb -= __uint_as_float(p1 & 0xffff0000u);
    p2 = pl_cvt_pk_bf16(ra, rb);
}

// ---------------------------------------------------------------------------------------------------------
// Packing pass.  One thread = one row x one k-step (16 values -> 3 x 32 bytes).
struct PackArgs {
    const float* src;
    int ld, rows, K;          // valid rows / k of the source
    int kmajor;               // 0: element(r, k) = src[r ld + k];  1: src[k ld + r]
    uint16_t* dst;            // plane 0 of the operand
    int Rp, KS, ks_off;       // padded rows, k-steps of the whole operand, first k-step of this segment
    int relu;                 // max(x, 0) on load (sources that are consumed through a relu)
};

__device__ __forceinline__ void pack_store(const PackArgs& a, int row, int s, const float (&v)[16]) {
    uint32_t q[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pl_split3(v[2 * e], v[2 * e + 1], q[0][e], q[1][e], q[2][e]);
    const int r = row & 31;
    const int sw = (r >> 3) & 1;
    const size_t plane = (size_t)a.KS * a.Rp * 16;                                   // uint16 elements of one plane
    uint16_t* base = a.dst + ((size_t)(a.ks_off + s) * a.Rp + (row & ~31)) * 16 + (size_t)r * 16;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        pl_u32x4* o = reinterpret_cast<pl_u32x4*>(base + p * plane);
        o[sw] = (pl_u32x4){q[p][0], q[p][1], q[p][2], q[p][3]};                       // half 0 -> chunk 2 r + sw
        o[sw ^ 1] = (pl_u32x4){q[p][4], q[p][5], q[p][6], q[p][7]};
    }
}

// K-contiguous source: thread t of 256 -> row t >> 2 of 64, k-step t & 3 of 4 (four lanes read 256 contiguous bytes of a row)
__global__ __launch_bounds__(256) void pack_rows_kernel(PackArgs a) {
    const int row = blockIdx.y * 64 + (threadIdx.x >> 2), s = blockIdx.x * 4 + (threadIdx.x & 3);
    if (s >= (a.K + 15) / 16) return;                     // beyond the segment's last (partial) k-step
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = 0.f;
    if (row < a.rows) {
        const float* p = a.src + (size_t)row * a.ld + s * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (s * 16 + 4 * j < a.K) {                   // K % 4 == 0
                const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * j + e] = a.relu ? fmaxf(x[e], 0.f) : x[e];
            }
    }
    pack_store(a, row, s, v);
}
// k-major source: thread t of 256 -> row t & 63 of 64 (coalesced along the rows), k-step t >> 6 of 4
__global__ __launch_bounds__(256) void pack_kmajor_kernel(PackArgs a) {
    const int row = blockIdx.y * 64 + (threadIdx.x & 63), s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= (a.K + 15) / 16) return;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = s * 16 + e;
        float x = (row < a.rows && k < a.K) ? a.src[(size_t)k * a.ld + row] : 0.f;
        v[e] = a.relu ? fmaxf(x, 0.f) : x;
    }
    pack_store(a, row, s, v);
}

size_t planes_bytes(int rows, int K) { return (size_t)3 * ((K + 15) / 16) * round_up(rows, 256) * 32; }

int planes_pack(const float* src, int ld, int rows, int K, bool kmajor, void* dst, int rows_padded, int ks_total, int ks_off, bool relu,
                hipStream_t st) {
    ICZ_REQUIRE(src && dst && rows > 0 && K > 0, "planes_pack: bad arguments");
    ICZ_REQUIRE(rows_padded % 256 == 0 && rows_padded >= rows, "planes_pack: padded rows %d", rows_padded);
    ICZ_REQUIRE(kmajor || (K % 4 == 0 && ld % 4 == 0 && ((uintptr_t)src & 15) == 0), "planes_pack: K-contiguous sources must be float4-aligned");
    const int ks = (K + 15) / 16;
    ICZ_REQUIRE(ks_off >= 0 && ks_off + ks <= ks_total, "planes_pack: k-steps %d + %d of %d", ks_off, ks, ks_total);
    PackArgs a = {src, ld, rows, K, kmajor ? 1 : 0, reinterpret_cast<uint16_t*>(dst), rows_padded, ks_total, ks_off, relu ? 1 : 0};
    const dim3 grid(cdiv(ks, 4), rows_padded / 64);
    if (kmajor) hipLaunchKernelGGL(pack_kmajor_kernel, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(pack_rows_kernel, grid, dim3(256), 0, st, a);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// ---------------------------------------------------------------------------------------------------------
// The GEMM.  BM x BN tile, WM x WN waves (wave tile BM / WM x BN / WN in 32 x 32 MFMA tiles), NSLOT k-steps of LDS ring
// (prefetch distance NSLOT - 1), one barrier per k-step.
struct PlanesKArgs {
    const uint16_t* A;
    const uint16_t* B;
    int Mp, Np, KS;
    int M, N;
    float* out;
    int ldo;
    const float* bias;
    int accumulate;
    int nsplit, ks_per_split;
    int tiles_m, tiles_n;
};

template <int N_>
__device__ __forceinline__ void pl_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int BM, int BN, int WM, int WN, int NSLOT, int WPE>
__global__ __launch_bounds__(64 * WM * WN, WPE) void gemm_planes_kernel(PlanesKArgs a) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char pl_smem[];
    constexpr int NW = WM * WN, TM = BM / WM, TN = BN / WN, FI = TM / 32, FU = TN / 32;
    constexpr int PA = 3 * (BM / 32), PB = 3 * (BN / 32), PIECES = PA + PB, G = PIECES / NW;
    static_assert(PIECES % NW == 0, "pieces per k-step must divide over the waves");
    static_assert(NSLOT == 2 || NSLOT == 3, "ring depth");
    constexpr int SLOT = PIECES * 1024, DIST = NSLOT - 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // tile of this workgroup: consecutive workgroups go to the 8 XCDs in turn -> give each XCD a contiguous run of a grouped
    // (4 row blocks x all column blocks) tile order, so that the tiles that share operand blocks share an L2
    const int tiles = a.tiles_m * a.tiles_n;
    const int z = blockIdx.x / tiles;
    int t = blockIdx.x % tiles;
    {
        const int q = tiles / 8, rr = tiles % 8, x = t % 8;
        t = (x < rr ? x * (q + 1) : rr * (q + 1) + (x - rr) * q) + t / 8;
    }
    constexpr int GM = 4;
    const int gsz = GM * a.tiles_n, grp = t / gsz, w = t % gsz;
    const int gm = min(GM, a.tiles_m - grp * GM);
    const int m0 = (grp * GM + w % gm) * BM, n0 = (w / gm) * BN;
    const int ks_beg = z * a.ks_per_split, ks_end = min(a.KS, ks_beg + a.ks_per_split);
    const int nks = ks_end - ks_beg;

    // this wave's pieces of a k-step: q = wave G + j
    const uint16_t* src[G];
    size_t stp[G];
    int dst[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
        const int q = wave * G + j;
        if (q < PA) {
            const int p = q / (BM / 32), rb = q % (BM / 32);
            src[j] = a.A + ((size_t)(p * a.KS + ks_beg) * a.Mp + m0 + rb * 32) * 16 + lane * 8;
            stp[j] = (size_t)a.Mp * 16;
        } else {
            const int qb = q - PA, p = qb / (BN / 32), rb = qb % (BN / 32);
            src[j] = a.B + ((size_t)(p * a.KS + ks_beg) * a.Np + n0 + rb * 32) * 16 + lane * 8;
            stp[j] = (size_t)a.Np * 16;
        }
        dst[j] = q * 1024;
    }
    auto issue = [&](int slot) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[j],
                                             (__attribute__((address_space(3))) void*)(pl_smem + slot * SLOT + dst[j]), 16, 0, 0);
            src[j] += stp[j];
        }
    };

    pl_f32x16 acc[FI][FU];
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int u = 0; u < FU; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][u][q] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int fo = (2 * r + (h ^ ((r >> 3) & 1))) * 16;
    const int fa = (wm * FI) * 1024 + fo, fb = (PA + wn * FU) * 1024 + fo;

    if (nks > 0) {
        issue(0);
        if (DIST > 1 && nks > 1) issue(1);
        if (DIST > 1 && nks > 1) pl_wait_vm<G>(); else pl_wait_vm<0>();
        __builtin_amdgcn_s_barrier();
    }
    int slot = 0, fill = DIST % NSLOT;
    for (int s = 0; s < nks; ++s) {
        if (s + DIST < nks) issue(fill);
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* sb = pl_smem + slot * SLOT;
        pl_bf16x8 af[3][FI], bf[3][FU];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int u = 0; u < FU; ++u) bf[p][u] = *reinterpret_cast<const pl_bf16x8*>(sb + fb + (p * (BN / 32) + u) * 1024);
#pragma unroll
            for (int i = 0; i < FI; ++i) af[p][i] = *reinterpret_cast<const pl_bf16x8*>(sb + fa + (p * (BM / 32) + i) * 1024);
        }
        constexpr int PP[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};      // smallest terms first
#pragma unroll
        for (int pr = 0; pr < 6; ++pr)
#pragma unroll
            for (int i = 0; i < FI; ++i)
#pragma unroll
                for (int u = 0; u < FU; ++u)
                    acc[i][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[PP[pr][0]][i], bf[PP[pr][1]][u], acc[i][u], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nks) {
            if (DIST > 1 && s + 2 < nks) pl_wait_vm<G>(); else pl_wait_vm<0>();
        }
        __builtin_amdgcn_s_barrier();
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        fill = fill + 1 == NSLOT ? 0 : fill + 1;
    }

    // acc[i][u][q] <-> row m0 + wm TM + 32 i + (q & 3) + 8 (q >> 2) + 4 h, column n0 + wn TN + 32 u + r
    const bool direct = a.nsplit == 1;
    float* const outp = direct ? a.out : a.out + (size_t)z * a.M * a.N;
    const int ldo = direct ? a.ldo : a.N;
#pragma unroll
    for (int i = 0; i < FI; ++i)
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int n = n0 + wn * TN + 32 * u + r;
            if (n >= a.N) continue;
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

template <int BM, int BN, int WM, int WN, int NSLOT, int WPE>
static int launch_planes(const PlanesGemm& g, hipStream_t st) {
    constexpr int PIECES = 3 * (BM / 32) + 3 * (BN / 32);
    constexpr size_t lds = (size_t)NSLOT * PIECES * 1024;
    static bool attr = false;
    if (!attr) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes_kernel<BM, BN, WM, WN, NSLOT, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    PlanesKArgs k = {};
    k.A = reinterpret_cast<const uint16_t*>(g.A); k.B = reinterpret_cast<const uint16_t*>(g.B);
    k.Mp = g.Mp; k.Np = g.Np; k.KS = g.KS; k.M = g.M; k.N = g.N; k.out = g.out; k.ldo = g.ldo; k.bias = g.bias; k.accumulate = g.accumulate;
    k.nsplit = g.nsplit;
    k.ks_per_split = cdiv(g.KS, g.nsplit);
    ICZ_REQUIRE(cdiv(g.KS, k.ks_per_split) == g.nsplit, "gemm_planes: nsplit %d leaves empty splits of %d k-steps", g.nsplit, g.KS);
    k.tiles_m = cdiv(g.M, BM); k.tiles_n = cdiv(g.N, BN);
    ICZ_REQUIRE(k.tiles_m * BM <= g.Mp && k.tiles_n * BN <= g.Np, "gemm_planes: planes padded to %d x %d rows, tiles need %d x %d", g.Mp, g.Np, k.tiles_m * BM, k.tiles_n * BN);
    hipLaunchKernelGGL((gemm_planes_kernel<BM, BN, WM, WN, NSLOT, WPE>), dim3(k.tiles_m * k.tiles_n * g.nsplit), dim3(64 * WM * WN), lds, st, k);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int gemm_planes(const PlanesGemm& g, hipStream_t st) {
    ICZ_REQUIRE(g.A && g.B && g.out && g.M > 0 && g.N > 0 && g.KS > 0, "gemm_planes: bad arguments");
    ICZ_REQUIRE(g.Mp % 256 == 0 && g.Np % 256 == 0, "gemm_planes: planes must be padded to 256 rows");
    ICZ_REQUIRE(g.nsplit >= 1 && (g.nsplit == 1 || (!g.bias && !g.accumulate)), "gemm_planes: bias / accumulate need nsplit == 1");
    switch (g.config) {
        case 1: return launch_planes<128, 256, 2, 2, 2, 2>(g, st);
        case 2: return launch_planes<128, 128, 2, 2, 3, 2>(g, st);
        case 3: return launch_planes<256, 256, 2, 4, 2, 2>(g, st);
        default: return launch_planes<256, 256, 2, 4, 3, 2>(g, st);
    }
}

}  // namespace icz
