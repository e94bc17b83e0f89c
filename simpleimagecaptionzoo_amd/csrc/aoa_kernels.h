// Kernels specific to the AoA model family (Models/AoA_Model.py): custom LayerNorm, multi-head dot-product attention
// (R x R self-attention in the refiner, 1 x R in the decoder; R = 36, 49, or up to 128 with per-image counts), GLU gate, general-p dropout.
#pragma once
#include "butd_kernels.h"

namespace icz {
namespace {

// Dropout with an arbitrary drop probability p (AoA uses 0.1 / 0.3 / 0.5): explicit keep-mask or Philox word >= p*2^32.
struct DropP {
    int mode;                 // 0 off, 1 explicit, 2 Philox
    const uint8_t* mask;
    const uint64_t* seed_p;
    uint32_t stream, step;
    uint32_t thresh;          // floor(p * 2^32)
    float scale;              // 1 / (1 - p)
    uint64_t idx0;            // elements in front of idx0 pass unchanged and the rest count from it: the evaluation-mode half of a paired refiner pass
    __device__ __forceinline__ float apply(float x, uint64_t idx) const {
        if (mode == 0 || idx < idx0) return x;
        idx -= idx0;
        bool keep;
        if (mode == 1) keep = mask[idx] != 0;
        else {
            const uint64_t g = idx >> 2;
            uint4_ c = {(uint32_t)g, (uint32_t)(g >> 32), step, stream};
            const uint64_t seed = *seed_p;
            uint4_ r = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            const uint32_t w = (idx & 3) == 0 ? r.x : ((idx & 3) == 1 ? r.y : ((idx & 3) == 2 ? r.z : r.w));
            keep = w >= thresh;
        }
        return keep ? x * scale : 0.f;
    }
};

// Region rows of a batch.  Fixed region sets: row (img, r) of the [n_img, R, *] tensors is img * R + r.  With per-image
// region counts ('adaptive' features) the refiner works on the PACKED valid rows only -- image img owns rows
// off[img] .. off[img] + count[img] -- as the reference does for the projection (pack_wrapper, AoA_Model.py:650-653); the
// padded rows it carries through the refiner never reach a result (their keys are masked everywhere).  rowmap[packed row] =
// img * R + r is the row's padded index, which still addresses the dropout masks / Philox counters.
struct RegionRows {
    const int32_t* off;        // [n_img + 1] or null (fixed)
    const int32_t* rowmap;     // [total rows] or null (fixed)
    const int32_t* lens;       // [n_img] or null (fixed)
    int R;
    __device__ __forceinline__ size_t first(int img) const { return off ? (size_t)off[img] : (size_t)img * R; }
    __device__ __forceinline__ int count(int img) const { return lens ? lens[img] : R; }
    __device__ __forceinline__ size_t padded(size_t row) const { return rowmap ? (size_t)rowmap[row] : row; }
};

// off = exclusive prefix sum of the counts, rowmap = padded index of every packed row (one workgroup)
__global__ __launch_bounds__(256) void aoa_offsets_kernel(const int32_t* __restrict__ lens, int n_img, int R, int32_t* __restrict__ off,
                                                          int32_t* __restrict__ rowmap) {
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < n_img; ++i) { off[i] = acc; acc += lens[i]; }
        off[n_img] = acc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_img * R; i += 256) {
        const int img = i / R, r = i % R;
        if (r < lens[img]) rowmap[off[img] + r] = i;
    }
}

// packed[row] = padded[rowmap[row]]   (rows of n floats, n % 4 == 0)
__global__ __launch_bounds__(256) void aoa_pack_rows_kernel(const float* __restrict__ padded, const int32_t* __restrict__ rowmap,
                                                            float* __restrict__ packed, size_t rows, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n4 = n >> 2;
    if (i >= rows * n4) return;
    const size_t row = i / n4, c = (i % n4) * 4;
    *reinterpret_cast<f32x4*>(packed + row * n + c) = *reinterpret_cast<const f32x4*>(padded + (size_t)rowmap[row] * n + c);
}

// padded[rowmap[row]] = packed[row]   (the padded tensor is zeroed beforehand)
__global__ __launch_bounds__(256) void aoa_unpack_rows_kernel(const float* __restrict__ packed, const int32_t* __restrict__ rowmap,
                                                              float* __restrict__ padded, size_t rows, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, n4 = n >> 2;
    if (i >= rows * n4) return;
    const size_t row = i / n4, c = (i % n4) * 4;
    *reinterpret_cast<f32x4*>(padded + (size_t)rowmap[row] * n + c) = *reinterpret_cast<const f32x4*>(packed + row * n + c);
}

// y = drop(relu(x)) (feature projection epilogue, AoA_Model.py:661-665); y may be x
__global__ __launch_bounds__(256) void relu_drop_kernel(const float* x, float* y, size_t n, DropP dp, RegionRows rr, int Hd) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t row = i / Hd;
    y[i] = dp.apply(fmaxf(x[i], 0.f), rr.padded(row) * Hd + i % Hd);
}

// Custom LayerNorm (AoA_Model.py:14-25): y = gain * (x - mean) / (std_unbiased + eps) + bias.  One wave per row; the row is
// read once with 16-byte loads and kept in registers (n <= 2048, n % 4 == 0; longer rows are re-read).
// Optionally stores (mean, 1/(std+eps)) per row for the backward pass.
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gain,
                                                         const float* __restrict__ bias, float* __restrict__ y, int rows, int n,
                                                         float* __restrict__ stats, const int* __restrict__ live = nullptr) {
    if (step_dead(live)) return;               // a rollout step behind the reference's break (icz_common.h)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);      // one wave per row
    if (row >= rows) return;
    const float* xr = x + (size_t)row * n;
    float* yr = y + (size_t)row * n;
    constexpr int NV = 8;                        // 8 float4 per lane = 2048 floats per row in registers
    if (n <= 256 * NV && (n & 3) == 0) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            v[u] = c < n ? *reinterpret_cast<const f32x4*>(xr + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
        }
        const float mean = wave_sum(s) / (float)n;
        float q = 0.f;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            if (4 * lane + 256 * u < n) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = v[u][j] - mean; q += d * d; }
            }
        }
        const float stdv = sqrtf(wave_sum(q) / (float)(n - 1));
        const float inv = 1.0f / (stdv + 1e-6f);
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int c = 4 * lane + 256 * u;
            if (c < n) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
                const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = g[j] * (v[u][j] - mean) * inv + b[j];
                *reinterpret_cast<f32x4*>(yr + c) = o;
            }
        }
        if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = inv; }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < n; c += 64) s += xr[c];
    const float mean = wave_sum(s) / (float)n;
    float v = 0.f;
    for (int c = lane; c < n; c += 64) { const float d = xr[c] - mean; v += d * d; }
    const float stdv = sqrtf(wave_sum(v) / (float)(n - 1));
    const float inv = 1.0f / (stdv + 1e-6f);
    for (int c = lane; c < n; c += 64) yr[c] = gain[c] * (xr[c] - mean) * inv + bias[c];
    if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = inv; }
}

// LDS row pitches of the attention tiles, in floats: a multiple of 4 (16-byte row reads) whose count of 16-byte slots is
// odd, so that consecutive rows start in different bank groups
__host__ __device__ __forceinline__ int aoa_pitch(int n) {
    int n4 = (n + 3) / 4;
    if ((n4 & 1) == 0) ++n4;
    return 4 * n4;
}

// Refiner self-attention (AoA_Model.py:41-69,113-117), one workgroup per (image, head):
//   S = Q_h K_h^T / sqrt(d);  S[:, r] = -1e9 where bu_mask[r] == 0;  P = softmax_rows(S);  P = drop(P, 0.1);  O_h = P V_h
// Q, K, V: [n_img, R, Hd] with head h in columns [h*d, (h+1)*d).  R <= 128.  The K and V head tiles stay in LDS while the
// queries go through in chunks of QC rows (QC = R when everything fits: 36 and 49 regions).  bu_masks are prefix masks
// (AoA_Engine.py:37-40), given as the valid count per image: exp(-1e9 - max) is exactly 0 in fp32, so the masked keys are
// skipped rather than computed.
// Both products are register-blocked: a thread owns a 4 x 4 block of S (rows q, q + nq/4, ...; keys r, r + len/4, ...) or a
// 4 x 4 block of O (4 strided rows x 4 adjacent columns) and walks the reduction dimension four at a time with 16-byte LDS
// reads -- 8 reads per 64 FMAs, where one element per thread needs 2 reads per FMA and leaves the kernel LDS-bound.
// (round 5) The same attention on the fp32 matrix pipe for region counts up to 64 (36 boxes, the 7 x 7 grid): the register-blocked kernel
// above keeps 81 of 256 threads busy in S = Q K^T at 36 regions and took 29 us per refiner layer at 64 images (58 us for the paired pass of
// an SCST step) for 75 MB of traffic.  Here a wave owns 16 x 16 tiles: S tiles with v_mfma_f32_16x16x4_f32 (exact fp32 products) straight
// from 16-byte row loads of Q and K (lane (i, g) holds row i, k = 16 j + 4 g + e -- the k order of a fragment is free as long as both
// operands use it), the scaled scores through a small LDS tile for the row softmax (one key per lane) and dropout, then O = P V with P
// fragments from LDS and V fragments as dword loads along the head dimension.  LDS: RP x (RP + 4) floats (10 KB at 36 regions).
// Same index for the dropout draw of P[q][k] as above.  d = Hd / NH must be a multiple of 16.
__global__ __launch_bounds__(256) void mha_self_mfma_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                            float* __restrict__ O, int R, int Hd, int NH, RegionRows rr, DropP dp, int ldi) {
    extern __shared__ __attribute__((aligned(16))) float sm_mha2[];
    const int img = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int d = Hd / NH;
    const int len = rr.count(img);
    const int nrow = rr.off ? len : R;                       // query rows of this image
    const int nt = (max(len, nrow) + 15) >> 4, RP = nt * 16, lp = RP + 4;
    float* sp = sm_mha2;                                     // [RP][lp]
    const size_t base = rr.first(img) * Hd + (size_t)hd * d;
    const size_t ibase = rr.first(img) * ldi + (size_t)hd * d;
    const float scale = 1.0f / sqrtf((float)d);
    // ---- S = Q K^T / sqrt(d): tile (qt, kt) = tile index / nt, % nt
    for (int tile = wave; tile < nt * nt; tile += 4) {
        const int qt = tile / nt, kt = tile % nt;
        const float* qp = Q + ibase + (size_t)min(qt * 16 + li, nrow - 1) * ldi + 4 * lg;      // rows behind the last one repeat it: never stored
        const float* kp = K + ibase + (size_t)min(kt * 16 + li, len - 1) * ldi + 4 * lg;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j0 = 0; j0 < d; j0 += 64) {
            f32x4 qf[4], kf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qf[j] = *reinterpret_cast<const f32x4*>(qp + j0 + 16 * j);
                kf[j] = *reinterpret_cast<const f32x4*>(kp + j0 + 16 * j);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[j][e], kf[j][e], acc, 0, 0, 0);
        }
        // acc[r] <-> query qt * 16 + 4 lg + r, key kt * 16 + li
#pragma unroll
        for (int r = 0; r < 4; ++r) sp[(qt * 16 + 4 * lg + r) * lp + kt * 16 + li] = acc[r] * scale;
    }
    __syncthreads();
    // ---- softmax over the valid keys of every query row (one wave per row, one key per lane), dropout, zeros behind the last key
    for (int q = wave; q < RP; q += 4) {
        const float v = lane < len ? sp[q * lp + lane] : -INFINITY;
        const float mx = wave_max(v);
        const float e = lane < len ? expf(v - mx) : 0.f;
        const float sum = wave_sum(e);
        const uint64_t idx = (((uint64_t)img * NH + hd) * R + q) * R;
        if (lane < RP) sp[q * lp + lane] = (lane < len && q < nrow) ? dp.apply(e / sum, idx + lane) : 0.f;
    }
    __syncthreads();
    // ---- O = P V: tile (qt, dt), dt over the head dimension in blocks of 16
    const int ndt = d >> 4;
    for (int tile = wave; tile < nt * ndt; tile += 4) {
        const int qt = tile / ndt, dt = tile % ndt;
        const float* pp = sp + (qt * 16 + li) * lp + 4 * lg;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < nt; ++j) {
            const f32x4 pf = *reinterpret_cast<const f32x4*>(pp + 16 * j);
            float vf[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vf[e] = V[ibase + (size_t)min(16 * j + 4 * lg + e, len - 1) * ldi + dt * 16 + li];      // P is zero behind the last key
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pf[e], vf[e], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + 4 * lg + r;
            if (q < nrow) O[base + (size_t)q * Hd + dt * 16 + li] = acc[r];
        }
    }
}

// The three projections of a refiner layer as one [3 Hd, Hd] weight and one [3 Hd] bias (one GEMM of N = 3 Hd instead of three
// of N = Hd, which leave half of the CUs idle): copied from the bound parameters at every weight refresh.
constexpr int AOA_QKV_MAX_LAYERS = 8;
struct QkvPackTable {
    const float* w[AOA_QKV_MAX_LAYERS][3];
    const float* b[AOA_QKV_MAX_LAYERS][3];
    float* wdst[AOA_QKV_MAX_LAYERS];
    float* bdst[AOA_QKV_MAX_LAYERS];
};
__global__ __launch_bounds__(256) void aoa_qkv_pack_kernel(QkvPackTable t, int Hd) {
    const int l = blockIdx.z, which = blockIdx.y;
    const size_t n4 = (size_t)Hd * Hd / 4, i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) reinterpret_cast<f32x4*>(t.wdst[l] + (size_t)which * Hd * Hd)[i] = reinterpret_cast<const f32x4*>(t.w[l][which])[i];
    if (i < (size_t)Hd) t.bdst[l][(size_t)which * Hd + i] = t.b[l][which][i];
}

// Q, K, V rows are `ldi` floats apart (3 Hd when they are the column blocks of one fused projection), O rows Hd.
__global__ __launch_bounds__(256) void mha_self_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                       float* __restrict__ O, int R, int Hd, int NH, int QC, RegionRows rr, DropP dp, int ldi) {
    extern __shared__ __attribute__((aligned(16))) float sm_mha[];     // K,V tiles [R4][ld], Q chunk [QC4][ld], P [QC4][lp]
    const int img = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
    const int d = Hd / NH, d4 = d >> 2, ld = aoa_pitch(d), lp = aoa_pitch(R);
    const int R4 = (R + 3) & ~3, QC4 = (QC + 3) & ~3;
    const int len = rr.count(img), len4 = (len + 3) & ~3;
    const int nrow = rr.off ? len : R;                       // query rows of this image: the packed layout has no padded ones
    float* sk = sm_mha;
    float* sv = sk + R4 * ld;
    float* sq = sv + R4 * ld;
    float* sp = sq + QC4 * ld;
    const size_t base = rr.first(img) * Hd + (size_t)hd * d;           // output rows
    const size_t ibase = rr.first(img) * ldi + (size_t)hd * d;         // Q / K / V rows
    for (int i = tid; i < len4 * d4; i += 256) {
        const int r = i / d4, j = (i % d4) * 4;
        f32x4 kk = {0.f, 0.f, 0.f, 0.f}, vv = kk;              // rows [len, len4) are read by the last block of four: zeros
        if (r < len) {
            kk = *reinterpret_cast<const f32x4*>(K + ibase + (size_t)r * ldi + j);
            vv = *reinterpret_cast<const f32x4*>(V + ibase + (size_t)r * ldi + j);
        }
        *reinterpret_cast<f32x4*>(sk + r * ld + j) = kk;
        *reinterpret_cast<f32x4*>(sv + r * ld + j) = vv;
    }
    const float scale = 1.0f / sqrtf((float)d);
    const int lane = tid & 63, wave = tid >> 6;
    const int nrt = len4 >> 2;
    for (int q0 = 0; q0 < nrow; q0 += QC) {
        const int nq = min(QC, nrow - q0), nqt = (nq + 3) >> 2;
        for (int i = tid; i < nq * d4; i += 256) {
            const int r = i / d4, j = (i % d4) * 4;
            *reinterpret_cast<f32x4*>(sq + r * ld + j) = *reinterpret_cast<const f32x4*>(Q + ibase + (size_t)(q0 + r) * ldi + j);
        }
        __syncthreads();
        for (int i = tid; i < nqt * nrt; i += 256) {
            const int qt = i / nrt, rt = i % nrt;
            const float *qp[4], *kp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                qp[c] = sq + min(qt + c * nqt, nq - 1) * ld;
                kp[c] = sk + min(rt + c * nrt, len - 1) * ld;
            }
            float acc[4][4] = {};
            for (int j = 0; j < d; j += 4) {
                f32x4 qv[4], kv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    qv[c] = *reinterpret_cast<const f32x4*>(qp[c] + j);
                    kv[c] = *reinterpret_cast<const f32x4*>(kp[c] + j);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[a][b] += qv[a][e] * kv[b][e];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int q = qt + a * nqt;
                if (q >= nq) continue;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = rt + b * nrt;
                    if (r < len) sp[q * lp + r] = acc[a][b] * scale;
                }
            }
        }
        __syncthreads();
        // softmax per query row: one wave per row, two keys per lane; columns [len, len4) are zeroed for the blocked P V
        for (int q = wave; q < nq; q += 4) {
            const int r1 = lane + 64;
            const float v0 = lane < len ? sp[q * lp + lane] : -INFINITY;
            const float v1 = r1 < len ? sp[q * lp + r1] : -INFINITY;
            const float mx = wave_max(fmaxf(v0, v1));
            const float e0 = lane < len ? expf(v0 - mx) : 0.f;
            const float e1 = r1 < len ? expf(v1 - mx) : 0.f;
            const float sum = wave_sum(e0 + e1);
            const uint64_t idx = (((uint64_t)img * NH + hd) * R + (q0 + q)) * R;
            if (lane < len4) sp[q * lp + lane] = lane < len ? dp.apply(e0 / sum, idx + lane) : 0.f;
            if (r1 < len4) sp[q * lp + r1] = r1 < len ? dp.apply(e1 / sum, idx + r1) : 0.f;
        }
        __syncthreads();
        for (int i = tid; i < nqt * d4; i += 256) {
            const int qt = i / d4, j = (i % d4) * 4;
            const float* pp[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) pp[c] = sp + min(qt + c * nqt, nq - 1) * lp;
            f32x4 acc[4] = {};
            for (int r = 0; r < len4; r += 4) {
                f32x4 pv[4], vv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    pv[c] = *reinterpret_cast<const f32x4*>(pp[c] + r);
                    vv[c] = *reinterpret_cast<const f32x4*>(sv + (r + c) * ld + j);
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[a] += pv[a][e] * vv[e];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int q = qt + a * nqt;
                if (q < nq) *reinterpret_cast<f32x4*>(O + base + (size_t)(q0 + q) * Hd + j) = acc[a];
            }
        }
        __syncthreads();
    }
}

// dropout over the concatenation [a | b] (each [rows, Hd]) with one mask of width 2*Hd (dropout_aoa, AoA_Model.py:118)
__global__ __launch_bounds__(256) void drop_concat_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ ad,
                                                          float* __restrict__ bd, size_t rows, int Hd, RegionRows rr, DropP dp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Hd) return;
    const size_t prow = rr.padded(i / Hd);
    const int c = (int)(i % Hd);
    ad[i] = dp.apply(a[i], prow * 2 * Hd + c);
    bd[i] = dp.apply(b[i], prow * 2 * Hd + Hd + c);
}

// GLU gate + sublayer residual (AoA_Model.py:83,39): x_out = x + drop(z[:, :Hd] * sigmoid(z[:, Hd:]), p)
__global__ __launch_bounds__(256) void glu_residual_kernel(const float* __restrict__ z, const float* __restrict__ x, float* __restrict__ out,
                                                           size_t rows, int Hd, RegionRows rr, DropP dp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * Hd) return;
    const size_t row = i / Hd;
    const int c = (int)(i % Hd);
    const float y = z[row * 2 * Hd + c] * sigmoidf_(z[row * 2 * Hd + Hd + c]);
    out[i] = x[i] + dp.apply(y, rr.padded(row) * Hd + c);
}

// Decoder: u = mean_feat[img] + drop(ctx_prev, 0.5)   (AoA_Model.py:321-323)
__global__ __launch_bounds__(256) void aoa_u_kernel(const float* __restrict__ meanf, const int32_t* __restrict__ img_of_row,
                                                    const float* __restrict__ ctx_prev, float* __restrict__ u, int rows, int Hd, DropP dp) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * Hd) return;
    const int row = (int)(i / Hd), c = (int)(i % Hd);
    const int img = img_of_row ? img_of_row[row] : row;
    u[i] = meanf[(size_t)img * Hd + c] + dp.apply(ctx_prev[i], i);
}

// K and V head tiles [R][d] -> LDS [R][d+1] by the workgroup's NT threads: 16-byte loads, eight in flight per thread (d % 4 == 0)
template <int NT>
__device__ __forceinline__ void aoa_stage_kv(const float* __restrict__ K, const float* __restrict__ V, float* sk, float* sv, int R, int d,
                                             int Hd, int tid) {
    const int d4 = d >> 2, n = R * d4, ld = d + 1;
    for (int i0 = tid; i0 < n; i0 += 4 * NT) {
        f32x4 kk[4], vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + NT * u, n - 1), r = i / d4, j = (i % d4) * 4;
            kk[u] = *reinterpret_cast<const f32x4*>(K + (size_t)r * Hd + j);
            vv[u] = *reinterpret_cast<const f32x4*>(V + (size_t)r * Hd + j);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + NT * u;
            if (i < n) {
                const int r = i / d4, j = (i % d4) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) { sk[r * ld + j + c] = kk[u][c]; sv[r * ld + j + c] = vv[u][c]; }
            }
        }
    }
}

// Decoder attention, one query per row (AoA_Model.py:329-334 -> :90-120), one workgroup of 256 threads per (row, head):
//   s_r = Qp_h . Kd_h[r] / sqrt(d);  P = softmax_R(s);  Pd = drop(P, 0.1);  x_h = sum_r Pd_r Vd_h[r]
// Kd / Vd: region rows x Hd (linear_K / linear_V of the refined features, hoisted: time-invariant).  R <= 128: thread r of the
// first two waves owns key r; all four waves stage the image's K / V head tiles (most of the kernel's time: each tile is
// used by one query only); with region counts only the valid rows are staged and the masked keys get P = 0.
// Saves P and Pd ([rows, NH, R]) when requested (backward).
// qns > 1: the query projection arrives as qns split-K slabs in `Qp` (slab z at + z * q_stride, no bias): the workgroup sums
// its head's d columns in slab order, adds q_bias and leaves the finished values in Qp_store (backward reads them) -- the
// slab_reduce launch between the projection GEMM and this kernel (40 per SCST step) is gone.
__global__ __launch_bounds__(256) void aoa_dec_attn_kernel(const float* __restrict__ Qp, const float* __restrict__ Kd,
                                                           const float* __restrict__ Vd, const int32_t* __restrict__ img_of_row,
                                                           float* __restrict__ xatt, float* __restrict__ P_out, float* __restrict__ Pd_out,
                                                           int R, int Hd, int NH, RegionRows rr, DropP dp,
                                                           int qns = 1, size_t q_stride = 0, const float* __restrict__ q_bias = nullptr,
                                                           float* __restrict__ Qp_store = nullptr, const int* __restrict__ live = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm_da[];    // K tile [R][d+1], V tile [R][d+1], q [d], p [128], red [4]
    if (step_dead(live)) return;               // a rollout step behind the reference's break (icz_common.h)
    const int row = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = Hd / NH, ld = d + 1;
    float* sk = sm_da;
    float* sv = sk + R * ld;
    float* sq = sv + R * ld;
    float* sp = sq + d;
    float* red = sp + 128;
    const int img = img_of_row ? img_of_row[row] : row;
    const int len = rr.count(img);
    const size_t base = rr.first(img) * Hd + (size_t)hd * d;
    aoa_stage_kv<256>(Kd + base, Vd + base, sk, sv, len, d, Hd, tid);
    for (int j = tid; j < d; j += 256) {
        const size_t qi = (size_t)row * Hd + (size_t)hd * d + j;
        float q = Qp[qi];
        if (qns > 1) {
            for (int z = 1; z < qns; ++z) q += Qp[(size_t)z * q_stride + qi];
            q += q_bias[hd * d + j];
            Qp_store[qi] = q;
        }
        sq[j] = q;
    }
    __syncthreads();
    float s = -INFINITY;
    if (tid < len) {
        float acc = 0.f;
        for (int j = 0; j < d; ++j) acc += sq[j] * sk[tid * ld + j];
        s = acc / sqrtf((float)d);
    }
    const float wmx = wave_max(s);
    if (lane == 0) red[wave] = wmx;
    __syncthreads();
    const float mx = fmaxf(red[0], red[1]);
    const float e = tid < len ? expf(s - mx) : 0.f;
    const float wsum = wave_sum(e);
    __syncthreads();
    if (lane == 0) red[wave] = wsum;
    __syncthreads();
    const float sum = red[0] + red[1];
    if (tid < 128) {
        const float p = e / sum;
        const uint64_t pidx = ((uint64_t)row * NH + hd) * R + tid;
        const float pd = tid < len ? dp.apply(p, pidx) : 0.f;
        sp[tid] = pd;
        if (tid < R) {
            if (P_out) P_out[pidx] = p;
            if (Pd_out) Pd_out[pidx] = pd;
        }
    }
    __syncthreads();
    for (int j = tid; j < d; j += 256) {
        float acc = 0.f;
        for (int r = 0; r < len; ++r) acc += sp[r] * sv[r * ld + j];
        xatt[(size_t)row * Hd + (size_t)hd * d + j] = acc;
    }
}

// Decoder GLU (no residual): ctx = z[:, :Hd] * sigmoid(z[:, Hd:]);  ctxdrop = drop(ctx, 0.5) for `predict`.  Optionally also
// the next step's LSTM input u_next = mean_feat[img] + drop(ctx, 0.5) with THAT step's dropout stream (aoa_u_kernel of step
// t + 1, AoA_Model.py:321-323): the chains that feed ctx straight into the next step (greedy, sampled, teacher-forced) save a
// launch per step; beam search re-gathers ctx by source beam first and keeps aoa_u_kernel.
__global__ __launch_bounds__(256) void aoa_glu_kernel(const float* __restrict__ zslab, int ns, const float* __restrict__ zbias,
                                                      float* __restrict__ z_out, float* __restrict__ ctx, float* __restrict__ ctxdrop,
                                                      int rows, int Hd, DropP dp, const float* __restrict__ meanf,
                                                      const int32_t* __restrict__ img_of_row, float* __restrict__ u_next, DropP dp_u,
                                                      const int* __restrict__ live = nullptr) {
    if (step_dead(live)) return;               // a rollout step behind the reference's break (icz_common.h)
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * Hd) return;
    const size_t row = i / Hd;
    const int c = (int)(i % Hd);
    const size_t MN = (size_t)rows * 2 * Hd;
    const float a = sum_slabs1(zslab, ns, MN, row * 2 * Hd + c) + zbias[c];
    const float b = sum_slabs1(zslab, ns, MN, row * 2 * Hd + Hd + c) + zbias[Hd + c];
    if (z_out) { z_out[row * 2 * Hd + c] = a; z_out[row * 2 * Hd + Hd + c] = b; }
    const float y = a * sigmoidf_(b);
    ctx[i] = y;
    ctxdrop[i] = dp.apply(y, i);
    if (u_next) {
        const int img = img_of_row ? img_of_row[row] : (int)row;
        u_next[i] = meanf[(size_t)img * Hd + c] + dp_u.apply(y, i);
    }
}

// mean over the (valid) regions of an image  (AoA_Model.py:250-253)
__global__ __launch_bounds__(256) void mean_rows_kernel(const float* __restrict__ x, float* __restrict__ m, int Hd, RegionRows rr) {
    const int img = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Hd) return;
    const int len = rr.count(img);
    const size_t r0 = rr.first(img);
    float s = 0.f;
    for (int r = 0; r < len; ++r) s += x[(r0 + r) * Hd + c];
    m[(size_t)img * Hd + c] = s / (float)len;
}

}  // namespace
}  // namespace icz
