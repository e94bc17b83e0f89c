// Device kernels of the BUTD decoder step (forward).  All HBM-bound byte movers: coalesced 16-byte accesses,
// wave-shuffle reductions, no atomics (bitwise reproducible).  Reference line numbers: Models/BUTD_Model.py.
#pragma once
#include <stdlib.h>

#include "icz_common.h"
#include "rng.h"

namespace icz {
namespace {   // internal linkage: this header is included by several translation units

// ---------------------------------------------------------------------------------------------------------
// weight_norm (old style, dim 0): w[r,:] = v[r,:] * g[r] / ||v[r,:]||   (:43-45, :84).  One wave per row.
// Also keeps ||v[r]|| for the backward pass.
__global__ __launch_bounds__(256) void weight_norm_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                          float* __restrict__ w, float* __restrict__ norm,
                                                          int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* vr = v + (size_t)row * cols;
    float ss = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float s = g[row] / nrm;
    float* wr = w + (size_t)row * cols;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        *reinterpret_cast<f32x4*>(wr + c) = x * s;
    }
    if (lane == 0) norm[row] = nrm;
}

// The four weight-normed layers of the decoder in ONE launch (blocks laid out job after job, 4 rows per block).
struct WeightNormJob { const float* v; const float* g; float* w; float* norm; int rows, cols, block0; };
struct WeightNormTable { WeightNormJob j[4]; int count; };
__global__ __launch_bounds__(256) void weight_norm_multi_kernel(WeightNormTable tab) {
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < tab.count; ++i)
        if ((int)blockIdx.x >= tab.j[i].block0) k = i;
    const WeightNormJob& jb = tab.j[k];
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - jb.block0) * 4 + (threadIdx.x >> 6);
    if (row >= jb.rows) return;
    const float* vr = jb.v + (size_t)row * jb.cols;
    float ss = 0.f;
    for (int c = lane * 4; c < jb.cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        ss += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
    ss = wave_sum(ss);
    const float nrm = sqrtf(ss);
    const float s = jb.g[row] / nrm;
    float* wr = jb.w + (size_t)row * jb.cols;
    for (int c = lane * 4; c < jb.cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        *reinterpret_cast<f32x4*>(wr + c) = x * s;
    }
    if (lane == 0) jb.norm[row] = nrm;
}

// ---------------------------------------------------------------------------------------------------------
// mean over regions (:117,:167,:205,:251): mean[b,d] = sum_r feats[b,r,d] / R
__global__ __launch_bounds__(256) void mean_feats_kernel(const float* __restrict__ feats, float* __restrict__ mean,
                                                         int R, int D) {
    const int b = blockIdx.y;
    const int d = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (d >= D) return;
    const float* f = feats + (size_t)b * R * D + d;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) s += *reinterpret_cast<const f32x4*>(f + (size_t)r * D);
    const float fr = (float)R;
    f32x4 o = {s[0] / fr, s[1] / fr, s[2] / fr, s[3] / fr};
    *reinterpret_cast<f32x4*>(mean + (size_t)b * D + d) = o;
}

// ---------------------------------------------------------------------------------------------------------
// sum split-K slabs (+ bias):  out[m, n] = sum_z slab[z, m, n] + bias[n]
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, int nsplit, size_t MN, int N,
                                                          const float* __restrict__ bias, float* __restrict__ out) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= MN) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(slab + i);
    for (int z = 1; z < nsplit; ++z) s += *reinterpret_cast<const f32x4*>(slab + (size_t)z * MN + i);
    if (bias) {
        int n = (int)(i % N);
        s += *reinterpret_cast<const f32x4*>(bias + n);
    }
    *reinterpret_cast<f32x4*>(out + i) = s;
}

// ---------------------------------------------------------------------------------------------------------
// Embedding -> ReLU -> Dropout (:77-81):  emb[row,:] = relu(E[it[row],:]) * keep * 2
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ table, const int64_t* __restrict__ it,
                                                    float* __restrict__ emb, int rows, int E, DropCfg dc, int relu = 1,
                                                    const int* __restrict__ live = nullptr) {
    if (step_dead(live)) return;               // a rollout step behind the reference's break (icz_common.h)
    const int row = blockIdx.y;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    f32x4 x = *reinterpret_cast<const f32x4*>(table + (size_t)it[row] * E + e);
    const int drow = row - dc.row0;                    // (DropCfg::row0: evaluation-mode rows of a merged chain in front)
    if (drow < 0) dc.mode = 0;
    uint32_t k = dc.mode ? dc.keep4((uint64_t)drow * E + e) : 0xFu;
    const float sc = dc.mode ? 2.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = ((k >> j) & 1u) ? (relu ? fmaxf(x[j], 0.f) : x[j]) * sc : 0.f;
    *reinterpret_cast<f32x4*>(emb + (size_t)row * E + e) = x;
}

// The same for every time step of a teacher-forced forward pass at once (the tokens are all known up front): block (x, t B + b),
// active while b < rows_t[t]; dropout of step t = the per-step configuration (Philox step t / mask slice t of [T][B][E]).
constexpr int EMB_MAX_T = 128;
struct EmbRows { int n[EMB_MAX_T]; };
__global__ __launch_bounds__(256) void embed_steps_kernel(const float* __restrict__ table, const int64_t* __restrict__ tok,
                                                          float* __restrict__ emb, int B, int E, EmbRows rows, DropCfg dc) {
    const int t = blockIdx.y / B, b = blockIdx.y % B;
    if (b >= rows.n[t]) return;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    dc.step = (uint32_t)t;
    if (dc.mode == 1) dc.mask += (size_t)t * B * E;
    f32x4 x = *reinterpret_cast<const f32x4*>(table + (size_t)tok[blockIdx.y] * E + e);
    uint32_t k = dc.mode ? dc.keep4((uint64_t)b * E + e) : 0xFu;
    const float sc = dc.mode ? 2.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = ((k >> j) & 1u) ? fmaxf(x[j], 0.f) * sc : 0.f;
    *reinterpret_cast<f32x4*>(emb + (size_t)blockIdx.y * E + e) = x;
}

// ---------------------------------------------------------------------------------------------------------
// LSTMCell pointwise part (:82-83; gate order i,f,g,o):
//   gates[row, :] = sum_z slab[z,row,:] + pre[pre_row,:] (optional) + b_ih + b_hh
//   c' = sig(f) c + sig(i) tanh(g);  h' = sig(o) tanh(c')
// Optionally stores the activated gates (for backward) and a dropped copy of h' (input of `predict`, :146).
struct LstmPointArgs {
    const float* slab; int nsplit;
    const float* pre; const int32_t* pre_row;     // pre may be null; pre_row null = identity
    const float* b_ih; const float* b_hh;
    const float* c_prev; float* h_out; float* c_out;
    float* gates_out;                              // [rows,4H] activated i,f,g,o or null
    float* hdrop_out;                              // [rows,H] or null
    int rows, H;
    const int* live;                               // step_dead(live): return at entry (icz_common.h)
};
// sum of ns split-K slabs at one element, loads issued four at a time (independent), added in slab order
__device__ __forceinline__ float sum_slabs1(const float* __restrict__ p, int ns, size_t stride, size_t off) {
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= ns; z += 4) {
        const float a = p[(size_t)z * stride + off], b = p[(size_t)(z + 1) * stride + off];
        const float c = p[(size_t)(z + 2) * stride + off], d = p[(size_t)(z + 3) * stride + off];
        s += a; s += b; s += c; s += d;
    }
    for (; z < ns; ++z) s += p[(size_t)z * stride + off];
    return s;
}

// grid (H/256, rows): one hidden unit per thread (4-byte accesses, 256 B per wave instruction, 4x the workgroups of
// a float4 layout -- at 64 rows the kernel is latency-bound, not bandwidth-bound)
__global__ __launch_bounds__(256) void lstm_point_kernel(LstmPointArgs a, DropCfg dc) {
    if (step_dead(a.live)) return;
    const int row = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    const int H = a.H, G = 4 * H;
    const size_t MN = (size_t)a.rows * G;
    float gt[4];
    const int pr = (a.pre && a.pre_row) ? a.pre_row[row] : row;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)row * G + q * H + j;
        float s = sum_slabs1(a.slab, a.nsplit, MN, off);
        if (a.pre) s += a.pre[(size_t)pr * G + q * H + j];
        s += a.b_ih[q * H + j];
        s += a.b_hh[q * H + j];
        gt[q] = s;
    }
    const float cp = a.c_prev[(size_t)row * H + j];
    const float gi = sigmoidf_(gt[0]), gf = sigmoidf_(gt[1]), gg = tanhf(gt[2]), go = sigmoidf_(gt[3]);
    const float cn = gf * cp + gi * gg;
    const float hn = go * tanhf(cn);
    a.h_out[(size_t)row * H + j] = hn;
    a.c_out[(size_t)row * H + j] = cn;
    if (a.gates_out) {
        float* go_ = a.gates_out + (size_t)row * G + j;
        go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
    }
    if (a.hdrop_out) {
        float hd = hn;
        const int drow = row - dc.row0;                // (DropCfg::row0)
        if (dc.mode && drow >= 0) hd = dc.keep((uint64_t)drow * H + j) ? hn * 2.0f : 0.f;
        a.hdrop_out[(size_t)row * H + j] = hd;
    }
}

// Gate-per-wave form, grid (H / 256, rows), 256 threads: wave q sums gate q of the workgroup's 256 hidden units (16 bytes per
// lane and slab, the loads of up to eight slabs independent), the four gates meet in LDS and thread j finishes unit j.  The
// resident-activation gate GEMM (gemm_resident_x3.hip) leaves 12 - 16 split-K slabs (16.8 MB at 64 rows): the one-unit kernel
// above then issues 64 four-byte wave loads per thread (8.3 us), a four-units-per-thread form with one wave per workgroup
// has ONE wave per compute unit waiting on 64 KB (9.5 us in the SCST trace); here four waves per compute unit wait on 16 KB
// each (5.1 us).  Same summation order per element (slabs ascending, pre, b_ih, b_hh): bit-identical to the kernel above.
__global__ __launch_bounds__(256) void lstm_point_gw_kernel(LstmPointArgs a, DropCfg dc) {
    __shared__ __attribute__((aligned(16))) float sg[4][256];
    const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int H = a.H, G = 4 * H;
    const int j0 = blockIdx.x * 256, j4 = j0 + lane * 4;
    const size_t MN = (size_t)a.rows * G;
    // early-out of a dead rollout step (live_flag / flag_dead, icz_common.h): tested once, behind the first batch of slab loads
    const int lflag = live_flag(a.live);
    bool tested = false;
#define ICZ_LIVE_TEST() do { if (!tested) { if (flag_dead(lflag)) return; tested = true; } } while (0)
    if (j4 < H) {
        const size_t off = (size_t)row * G + (size_t)q * H + j4;
        const float* p = a.slab + off;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= a.nsplit; z += 8) {
            f32x4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(z + i) * MN);
            ICZ_LIVE_TEST();
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
        }
        for (; z + 4 <= a.nsplit; z += 4) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (size_t)(z + i) * MN);
            ICZ_LIVE_TEST();
#pragma unroll
            for (int i = 0; i < 4; ++i) s += v[i];
        }
        for (; z < a.nsplit; ++z) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + (size_t)z * MN);
            ICZ_LIVE_TEST();
            s += v;
        }
        if (a.pre) {
            const int pr = a.pre_row ? a.pre_row[row] : row;
            s += *reinterpret_cast<const f32x4*>(a.pre + (size_t)pr * G + (size_t)q * H + j4);
        }
        s += *reinterpret_cast<const f32x4*>(a.b_ih + (size_t)q * H + j4);
        s += *reinterpret_cast<const f32x4*>(a.b_hh + (size_t)q * H + j4);
        *reinterpret_cast<f32x4*>(&sg[q][lane * 4]) = s;
    }
    ICZ_LIVE_TEST();          // (lanes past H, no slabs)
#undef ICZ_LIVE_TEST
    __syncthreads();
    const int j = j0 + tid;
    if (j >= H) return;
    const float cp = a.c_prev[(size_t)row * H + j];
    const float gi = sigmoidf_(sg[0][tid]), gf = sigmoidf_(sg[1][tid]), gg = tanhf(sg[2][tid]), go = sigmoidf_(sg[3][tid]);
    const float cn = gf * cp + gi * gg;
    const float hn = go * tanhf(cn);
    a.h_out[(size_t)row * H + j] = hn;
    a.c_out[(size_t)row * H + j] = cn;
    if (a.gates_out) {
        float* go_ = a.gates_out + (size_t)row * G + j;
        go_[0] = gi; go_[H] = gf; go_[2 * H] = gg; go_[3 * H] = go;
    }
    if (a.hdrop_out) {
        float hd = hn;
        const int drow = row - dc.row0;                // (DropCfg::row0)
        if (dc.mode && drow >= 0) hd = dc.keep((uint64_t)drow * H + j) ? hn * 2.0f : 0.f;
        a.hdrop_out[(size_t)row * H + j] = hd;
    }
}

// the gate-per-wave kernel wherever the hidden size allows 16-byte accesses
inline void launch_lstm_point(const LstmPointArgs& a, const DropCfg& dc, hipStream_t st) {
    const dim3 grid(cdiv(a.H, 256), a.rows);
    if (a.H % 4 != 0) hipLaunchKernelGGL(lstm_point_kernel, grid, dim3(256), 0, st, a, dc);
    else hipLaunchKernelGGL(lstm_point_gw_kernel, grid, dim3(256), 0, st, a, dc);
}

// ---------------------------------------------------------------------------------------------------------
// SoftAttention scores (:57-59):  score[row,r] = w_aff . drop(relu(enc_ctx[img,r,:] + dec_ctx[row,:])) + b_aff
// with dec_ctx[row,:] = sum_z slab[z,row,:] + b_dec (the dec_att GEMM's split-K partials are summed here).
// Grid (rows, parts): a workgroup builds dec_ctx[row] in LDS once and its 4 waves walk the regions of its part;
// each region row of enc_ctx (A floats) is read once with 16-byte loads and reduced by wave shuffles.
struct AttScoreArgs {
    const float* enc_ctx;        // [n_img, R, A] hoisted enc_att(feats) + bias
    const int32_t* img_of_row;   // null = identity
    const float* dec_slab; int nsplit;
    const float* b_dec;          // [A]
    const float* w_aff;          // [A] weight-normed
    const float* b_aff;          // [1]
    float* dec_ctx_out;          // [rows, A] or null (kept for backward)
    float* scores;               // [rows, R]
    int rows, R, A;
    const int* live;             // step_dead(live): return at entry
};
// Grid (rows, parts), 256 threads.  Part p owns regions p, p + parts, p + 2 parts, ...; a wave takes up to three of them
// at a time and issues all their loads (3 regions x 4 float4 per lane) before the first use: the kernel moves
// R*A*4 bytes per row once and is bound by how many bytes each CU keeps in flight, not by arithmetic.
__global__ __launch_bounds__(256) void att_scores_kernel(AttScoreArgs a, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float sdec[];   // A floats
    const int lflag = live_flag(a.live);
    const int row = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t MN = (size_t)a.rows * a.A;
    const int img = a.img_of_row ? a.img_of_row[row] : row;
    constexpr int NB = 3;
    // the projected features do not depend on dec_ctx: the loads of the wave's first group of regions (its only one at 36
    // regions, A <= 1024) are issued before the split-K slabs are summed, so both round trips overlap
    f32x4 xf[NB][4];
    if (part + nparts * wave < a.R) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int r = part + nparts * (wave + 4 * b);
            const float* e = a.enc_ctx + ((size_t)img * a.R + (r < a.R ? r : part + nparts * wave)) * a.A;
#pragma unroll
            for (int u = 0; u < 4; ++u) xf[b][u] = *reinterpret_cast<const f32x4*>(e + min(lane * 4 + 256 * u, a.A - 4));
        }
    }
    if (flag_dead(lflag)) return;                 // behind the first loads, in front of the first write (icz_common.h)
    for (int c = tid * 4; c < a.A; c += 1024) {
        const size_t off = (size_t)row * a.A + c;
        f32x4 s = *reinterpret_cast<const f32x4*>(a.dec_slab + off);
        for (int z = 1; z < a.nsplit; ++z) s += *reinterpret_cast<const f32x4*>(a.dec_slab + (size_t)z * MN + off);
        s += *reinterpret_cast<const f32x4*>(a.b_dec + c);
        *reinterpret_cast<f32x4*>(sdec + c) = s;
        if (a.dec_ctx_out && part == 0) *reinterpret_cast<f32x4*>(a.dec_ctx_out + off) = s;
    }
    __syncthreads();
    const float baff = a.b_aff[0];
    const int drow = row - dc.row0;                    // (DropCfg::row0: evaluation-mode rows of a merged chain in front)
    if (drow < 0) dc.mode = 0;
    const float sc = dc.mode ? 2.0f : 1.0f;
    const bool shared_bits = dc.mode == 2 && (a.A & 255) == 0;       // whole 256-column strips: every lane runs every c0 iteration
    for (int i0 = wave; part + nparts * i0 < a.R; i0 += 4 * NB) {
        int rr[NB];
        bool ok[NB];
        const float* e[NB];
        float acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int r = part + nparts * (i0 + 4 * b);
            ok[b] = r < a.R;
            rr[b] = ok[b] ? r : part + nparts * i0;
            e[b] = a.enc_ctx + ((size_t)img * a.R + rr[b]) * a.A;
            acc[b] = 0.f;
        }
        for (int c0 = lane * 4; c0 < a.A; c0 += 1024) {
            f32x4 x[NB][4];
            if (i0 == wave && c0 == lane * 4) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[b][u] = xf[b][u];
            } else {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int u = 0; u < 4; ++u) x[b][u] = *reinterpret_cast<const f32x4*>(e[b] + min(c0 + 256 * u, a.A - 4));
            }
            // Philox keep-bits: a call covers 128 consecutive columns = one half-wave of one u; the 8 blocks of each of the NB
            // regions are computed once (lane 8 b + block) and handed round by shuffles instead of 64 lanes calling it 4 NB times
            uint32_t kq[NB][4];
            if (shared_bits) {
                const int pb = lane >> 3, pg = lane & 7;
                const int pr = pb == 0 ? rr[0] : (pb == 1 ? rr[1] : rr[NB - 1]);
                const uint64_t g = (((uint64_t)drow * a.R + pr) * a.A + (c0 - lane * 4) + 128 * pg) >> 7;
                const uint64_t seed = *dc.seed_p;
                uint4_ ctr = {(uint32_t)g, (uint32_t)(g >> 32), dc.step, dc.stream};
                const uint4_ pw = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int src = 8 * b + 2 * u + (lane >> 5);
                        const uint32_t w0 = __shfl(pw.x, src), w1 = __shfl(pw.y, src), w2 = __shfl(pw.z, src), w3 = __shfl(pw.w, src);
                        const int wi = (lane >> 3) & 3;
                        const uint32_t ww = wi == 0 ? w0 : (wi == 1 ? w1 : (wi == 2 ? w2 : w3));
                        kq[b][u] = (ww >> ((4 * lane) & 31)) & 0xFu;
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 256 * u;
                if (c < a.A) {
                    const f32x4 d = *reinterpret_cast<const f32x4*>(sdec + c);
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.w_aff + c);
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        const uint32_t k = shared_bits ? kq[b][u] : (dc.mode ? dc.keep4(((uint64_t)drow * a.R + rr[b]) * a.A + c) : 0xFu);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float zv = fmaxf(x[b][u][j] + d[j], 0.f);
                            zv = ((k >> j) & 1u) ? zv * sc : 0.f;
                            acc[b] += zv * w[j];
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float v = wave_sum(acc[b]);
            if (lane == 0 && ok[b]) a.scores[(size_t)row * a.R + rr[b]] = v + baff;
        }
    }
}

constexpr int ATT_CTX_MAX_G = 8;
// The same for beam search, where the G = beam rows of an image sit next to each other (row = img * G + g) and share its
// projected features: one workgroup per (image, part) builds the G dec_ctx rows in LDS and walks its regions once, every
// region row of enc_ctx (A floats) is fetched once instead of G times (at 640 rows the per-row kernel re-reads 47 MB per
// step through the caches).  Evaluation mode only (beam search has no dropout); same arithmetic and summation order per
// row as att_scores_kernel.  Grid (n_img, parts).
__global__ __launch_bounds__(256) void att_scores_group_kernel(AttScoreArgs a, int G) {
    extern __shared__ __attribute__((aligned(16))) float sdec[];   // [G][A]
    const int img = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t MN = (size_t)a.rows * a.A;
    for (int g = 0; g < G; ++g) {
        const int row = img * G + g;
        for (int c = tid * 4; c < a.A; c += 1024) {
            const size_t off = (size_t)row * a.A + c;
            f32x4 s = *reinterpret_cast<const f32x4*>(a.dec_slab + off);
            for (int z = 1; z < a.nsplit; ++z) s += *reinterpret_cast<const f32x4*>(a.dec_slab + (size_t)z * MN + off);
            s += *reinterpret_cast<const f32x4*>(a.b_dec + c);
            *reinterpret_cast<f32x4*>(sdec + (size_t)g * a.A + c) = s;
            if (a.dec_ctx_out && part == 0) *reinterpret_cast<f32x4*>(a.dec_ctx_out + off) = s;
        }
    }
    __syncthreads();
    const float baff = a.b_aff[0];
    for (int i0 = wave; part + nparts * i0 < a.R; i0 += 4) {
        const int r = part + nparts * i0;
        const float* e = a.enc_ctx + ((size_t)img * a.R + r) * a.A;
        float acc[ATT_CTX_MAX_G];
#pragma unroll
        for (int g = 0; g < ATT_CTX_MAX_G; ++g) acc[g] = 0.f;
        for (int c0 = lane * 4; c0 < a.A; c0 += 1024) {
            f32x4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const f32x4*>(e + min(c0 + 256 * u, a.A - 4));
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + 256 * u;
                if (c < a.A) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(a.w_aff + c);
#pragma unroll
                    for (int g = 0; g < ATT_CTX_MAX_G; ++g)
                        if (g < G) {
                            const f32x4 d = *reinterpret_cast<const f32x4*>(sdec + (size_t)g * a.A + c);
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[g] += fmaxf(x[u][j] + d[j], 0.f) * w[j];
                        }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < ATT_CTX_MAX_G; ++g)
            if (g < G) {
                const float v = wave_sum(acc[g]);
                if (lane == 0) a.scores[(size_t)(img * G + g) * a.R + r] = v + baff;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// softmax over regions + attention-weighted feature sum (:60-61):
//   alpha = softmax_r(score[row,:]);  ctx[row, d] = sum_r alpha[r] * feats[img, r, d]
// Grid (rows, D/512), 256 threads: thread (half, cg) sums one half of the regions for 4 consecutive d (16-byte loads,
// 9 in flight per thread); the two halves meet in LDS.  Every wave recomputes the R <= 64 softmax (lane r holds score r).
__global__ __launch_bounds__(256) void att_ctx_kernel(const float* __restrict__ feats, const int32_t* __restrict__ img_of_row,
                                                      const float* __restrict__ scores, float* __restrict__ alpha_out,
                                                      float* __restrict__ alpha_out2, int alpha2_stride,
                                                      float* __restrict__ ctx, int R, int D, const int* __restrict__ live) {
    __shared__ float sal[64];
    __shared__ __attribute__((aligned(16))) float spart[128 * 4];
    const int lflag = live_flag(live);
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int cg = tid & 127, half = tid >> 7;
    const int d = blockIdx.y * 512 + cg * 4;
    const bool valid = d < D;
    const int img = img_of_row ? img_of_row[row] : row;
    const int Rh = (R + 1) / 2;
    const int r_lo = half ? Rh : 0, r_hi = half ? R : Rh;
    const float* f = feats + (size_t)img * R * D + (valid ? d : 0);
    // the feature loads do not depend on the weights: the first two rounds (all of them at 36 regions) are issued before the
    // scores are read, so the softmax runs while they are in flight instead of in front of them
    constexpr int RB = 9;
    f32x4 x0[RB], x1[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) x0[u] = *reinterpret_cast<const f32x4*>(f + (size_t)min(r_lo + u, r_hi - 1) * D);
#pragma unroll
    for (int u = 0; u < RB; ++u) x1[u] = *reinterpret_cast<const f32x4*>(f + (size_t)min(r_lo + RB + u, r_hi - 1) * D);
    if (flag_dead(lflag)) return;                 // behind the first loads, in front of the first write (icz_common.h)
    const float scv = lane < R ? scores[(size_t)row * R + lane] : -INFINITY;
    const float mx = wave_max(scv);
    const float ex = lane < R ? expf(scv - mx) : 0.f;
    const float sum = wave_sum(ex);
    const float al = ex / sum;
    if (tid < 64) {
        sal[tid] = al;
        if (blockIdx.y == 0 && tid < R) {
            alpha_out[(size_t)row * R + tid] = al;
            if (alpha_out2) alpha_out2[(size_t)row * alpha2_stride + tid] = al;
        }
    }
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
#pragma unroll
        for (int u = 0; u < RB; ++u)
            if (r_lo + u < r_hi) acc += x0[u] * sal[r_lo + u];
#pragma unroll
        for (int u = 0; u < RB; ++u)
            if (r_lo + RB + u < r_hi) acc += x1[u] * sal[r_lo + RB + u];
        for (int r0 = r_lo + 2 * RB; r0 < r_hi; r0 += RB) {
            f32x4 x[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) x[u] = *reinterpret_cast<const f32x4*>(f + (size_t)min(r0 + u, r_hi - 1) * D);
#pragma unroll
            for (int u = 0; u < RB; ++u)
                if (r0 + u < r_hi) acc += x[u] * sal[r0 + u];
        }
    }
    if (half) *reinterpret_cast<f32x4*>(spart + cg * 4) = acc;
    __syncthreads();
    if (!half && valid) *reinterpret_cast<f32x4*>(ctx + (size_t)row * D + d) = acc + *reinterpret_cast<const f32x4*>(spart + cg * 4);
}

// The same for beam search, where the G = beam rows of an image sit next to each other (row = img * G + b) and share its
// features: one workgroup does the G rows of one image for its 512 columns, so every feature vector is fetched once
// instead of G times (at 640 rows the per-row kernel moves 189 MB through the caches per step).  Same arithmetic and
// summation order per row as att_ctx_kernel.  Grid (n_img, D/512).
__global__ __launch_bounds__(256) void att_ctx_group_kernel(const float* __restrict__ feats, const float* __restrict__ scores,
                                                            float* __restrict__ alpha_out, float* __restrict__ ctx, int R, int D, int G) {
    __shared__ float sal[ATT_CTX_MAX_G][64];
    __shared__ __attribute__((aligned(16))) float spart[ATT_CTX_MAX_G][128 * 4];
    const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int g = wave; g < G; g += 4) {                 // wave w: softmax of rows w, w + 4
        const int row = img * G + g;
        const float scv = lane < R ? scores[(size_t)row * R + lane] : -INFINITY;
        const float mx = wave_max(scv);
        const float ex = lane < R ? expf(scv - mx) : 0.f;
        const float sum = wave_sum(ex);
        const float al = ex / sum;
        sal[g][lane] = al;
        if (blockIdx.y == 0 && lane < R) alpha_out[(size_t)row * R + lane] = al;
    }
    __syncthreads();
    const int cg = tid & 127, half = tid >> 7;
    const int d = blockIdx.y * 512 + cg * 4;
    const bool valid = d < D;
    const int Rh = (R + 1) / 2;
    const int r_lo = half ? Rh : 0, r_hi = half ? R : Rh;
    const float* f = feats + (size_t)img * R * D + (valid ? d : 0);
    f32x4 acc[ATT_CTX_MAX_G];
#pragma unroll
    for (int g = 0; g < ATT_CTX_MAX_G; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (valid) {
        constexpr int RB = 9;
        for (int r0 = r_lo; r0 < r_hi; r0 += RB) {
            f32x4 x[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) x[u] = *reinterpret_cast<const f32x4*>(f + (size_t)min(r0 + u, r_hi - 1) * D);
#pragma unroll
            for (int g = 0; g < ATT_CTX_MAX_G; ++g)
                if (g < G) {
#pragma unroll
                    for (int u = 0; u < RB; ++u)
                        if (r0 + u < r_hi) acc[g] += x[u] * sal[g][r0 + u];
                }
        }
    }
    if (half) {
#pragma unroll
        for (int g = 0; g < ATT_CTX_MAX_G; ++g)
            if (g < G) *reinterpret_cast<f32x4*>(spart[g] + cg * 4) = acc[g];
    }
    __syncthreads();
    if (!half && valid) {
#pragma unroll
        for (int g = 0; g < ATT_CTX_MAX_G; ++g)
            if (g < G) *reinterpret_cast<f32x4*>(ctx + (size_t)(img * G + g) * D + d) = acc[g] + *reinterpret_cast<const f32x4*>(spart[g] + cg * 4);
    }
}

__device__ __forceinline__ f32x4 sum_slabs4(const float* p, int ns, size_t slab_stride, size_t off) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (!p) return s;
    s = *reinterpret_cast<const f32x4*>(p + off);
    for (int z = 1; z < ns; ++z) s += *reinterpret_cast<const f32x4*>(p + (size_t)z * slab_stride + off);
    return s;
}

// ---------------------------------------------------------------------------------------------------------
// greedy epilogue (:183): id = argmax_v logits[row, v] (first maximum wins, as torch.max does), in two stages:
//   argmax_part_kernel  grid (rows, P): block-wide (value, index) of its slice of the vocabulary -> part[row, p]
//   embed_argmax_kernel grid (E/1024, rows): reduces the P partials (tiny), records the id, and gathers the next
//                       step's embedding row (Embedding -> ReLU; eval mode, no dropout) in the same launch.
__device__ __forceinline__ void argmax_combine(float& best, int& bi, float ob, int oi) {
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
}
// ns > 1: `logits` holds the ns split-K slabs of the predict GEMM (slab z at + z * slab_stride, no bias yet): the logit is their
// sum in slab order + bias[v] (the resident-activation GEMM of gemm_resident_x3.hip leaves four slabs at 33 - 64 rows).
__global__ __launch_bounds__(256) void argmax_part_kernel(const float* __restrict__ logits, int V, int ldl, int P,
                                                          float* __restrict__ part_val, int* __restrict__ part_idx,
                                                          int ns = 1, size_t slab_stride = 0, const float* __restrict__ bias = nullptr) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.x, p = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = ((V + P - 1) / P + 3) & ~3;
    const int v0 = p * per, v1 = min(V, v0 + per);
    const float* l = logits + (size_t)row * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (ns > 1) {
        // 16-byte loads where the row, the slab stride and the slice start allow it (v0 is a multiple of 4 by construction)
        const bool vec = ((ldl | (int)(slab_stride & 3)) & 3) == 0 && (((uintptr_t)logits | (uintptr_t)bias) & 15) == 0;
        const int v1v = vec ? v0 + ((v1 - v0) & ~3) : v0;
        for (int v = v0 + tid * 4; v < v1v; v += 1024) {
            f32x4 x = *reinterpret_cast<const f32x4*>(l + v);
            for (int z = 1; z < ns; ++z) x += *reinterpret_cast<const f32x4*>(l + (size_t)z * slab_stride + v);
            x += *reinterpret_cast<const f32x4*>(bias + v);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (x[j] > best) { best = x[j]; bi = v + j; }
        }
        for (int v = v1v + tid; v < v1; v += 256) {
            float x = l[v];
            for (int z = 1; z < ns; ++z) x += l[(size_t)z * slab_stride + v];
            x += bias[v];
            if (x > best || (x == best && v < bi)) { best = x; bi = v; }
        }
    } else {
        for (int v = v0 + tid; v < v1; v += 256) {
            const float x = l[v];
            if (x > best) { best = x; bi = v; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w) argmax_combine(best, bi, sv[w], si[w]);
        part_val[row * P + p] = best;
        part_idx[row * P + p] = bi;
    }
}
__global__ __launch_bounds__(256) void embed_argmax_kernel(const float* __restrict__ part_val, const int* __restrict__ part_idx, int P,
                                                           const float* __restrict__ table, int E, float* __restrict__ emb,
                                                           int64_t* __restrict__ it_next, int64_t* __restrict__ ids_out,
                                                           int ids_stride, int t, int relu = 1) {
    const int row = blockIdx.y;
    float best = part_val[row * P];
    int bi = part_idx[row * P];
    for (int p = 1; p < P; ++p) argmax_combine(best, bi, part_val[row * P + p], part_idx[row * P + p]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        it_next[row] = bi;
        if (ids_out) ids_out[(size_t)row * ids_stride + t] = bi;
    }
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    f32x4 x = *reinterpret_cast<const f32x4*>(table + (size_t)bi * E + e);
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = relu ? fmaxf(x[j], 0.f) : x[j];
    *reinterpret_cast<f32x4*>(emb + (size_t)row * E + e) = x;
}

// Greedy token choice of a decoder step in ONE launch (round 3), one 1024-thread workgroup per row like the multinomial kernel:
// logits = sum of the predict GEMM's split-K slabs + bias (never written back: greedy decoding keeps no logits), argmax with ties to
// the lowest index (torch.max, :183), the id recorded, and the next step's input embedding (Embedding -> ReLU, eval mode) gathered
// by the same workgroup.  Replaces argmax_part_kernel + embed_argmax_kernel (5.6 + 4.8 us) at 33 - 64 rows; 64 workgroups leave
// the other three quarters of the chip to the sampled chain that runs beside the greedy one in an SCST step.
__global__ __launch_bounds__(1024) void greedy_select_kernel(const float* __restrict__ logits, int V, int ldl, int ns, size_t slab_stride,
                                                            const float* __restrict__ bias, const float* __restrict__ table, int E,
                                                            float* __restrict__ emb, int64_t* __restrict__ it_next,
                                                            int64_t* __restrict__ ids_out, int ids_stride, int t, int relu = 1,
                                                            uint8_t* __restrict__ unfinished = nullptr, int* __restrict__ n_unfinished = nullptr) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // SCST baseline only (n_unfinished != null): once EVERY row has emitted <end> nothing downstream reads the greedy ids any more
    // (get_self_critical_reward cuts each row at its <end>, Utils.py:354) -- the steps behind that point return at entry, ids = 0
    if (n_unfinished && t > 0 && n_unfinished[t - 1] == 0) {
        if (tid == 0 && ids_out) ids_out[(size_t)row * ids_stride + t] = 0;
        return;
    }
    const float* l = logits + (size_t)row * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    const bool vec = ((ldl | (int)(slab_stride & 3)) & 3) == 0 && (((uintptr_t)logits | (uintptr_t)bias) & 15) == 0;
    const int Vv = vec ? (V & ~3) : 0;
    for (int v = tid * 4; v < Vv; v += 4096) {
        f32x4 x = *reinterpret_cast<const f32x4*>(l + v);
        for (int z = 1; z < ns; ++z) x += *reinterpret_cast<const f32x4*>(l + (size_t)z * slab_stride + v);
        if (bias) x += *reinterpret_cast<const f32x4*>(bias + v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x[j] > best) { best = x[j]; bi = v + j; }
    }
    for (int v = Vv + tid; v < V; v += 1024) {
        float x = l[v];
        for (int z = 1; z < ns; ++z) x += l[(size_t)z * slab_stride + v];
        if (bias) x += bias[v];
        if (x > best || (x == best && v < bi)) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, bi, __shfl_xor(best, o, 64), __shfl_xor(bi, o, 64));
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    best = sv[0]; bi = si[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) argmax_combine(best, bi, sv[w], si[w]);
    if ((unsigned)bi >= (unsigned)V) bi = 0;        // a row of NaN / -inf logits (diverged training) never updates bi: <pad>, not a wild read
    if (tid == 0) {
        it_next[row] = bi;
        if (ids_out) ids_out[(size_t)row * ids_stride + t] = bi;
        if (n_unfinished) {
            const bool unf = (t == 0 || unfinished[row] != 0) && bi != 2;
            unfinished[row] = unf ? 1 : 0;
            if (unf) atomicAdd(&n_unfinished[t], 1);
        }
    }
    for (int e = tid * 4; e < E; e += 4096) {
        f32x4 x = *reinterpret_cast<const f32x4*>(table + (size_t)bi * E + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = relu ? fmaxf(x[j], 0.f) : x[j];      // NIC embeds without the ReLU (NIC_Model.py:112)
        *reinterpret_cast<f32x4*>(emb + (size_t)row * E + e) = x;
    }
}

__global__ void set_scalars_kernel(uint64_t* seed_p, uint64_t seed, float* f_p, float f) {
    if (seed_p) *seed_p = seed;
    if (f_p) *f_p = f;
}

// dst[c][r] = src[r * ld + c] for up to four matrices in one launch (rows, cols multiples of 64; dst row stride = rows): the
// transposed LSTM weight copies behind the per-step dgrad GEMMs of BPTT, refreshed once per optimiser step.
struct TransposeJob { const float* src; int ld, rows, cols; float* dst; int block0; };
struct TransposeTable { TransposeJob j[4]; int count; };
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeTable tab) {
    __shared__ float tile[64][65];
    int ji = 0;
    for (int k = 1; k < tab.count; ++k) if ((int)blockIdx.x >= tab.j[k].block0) ji = k;
    const TransposeJob& jb = tab.j[ji];
    const int b = blockIdx.x - jb.block0, ntc = jb.cols / 64;
    const int r0 = (b / ntc) * 64, c0 = (b % ntc) * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = ty + 16 * i;
        const f32x4 v = *reinterpret_cast<const f32x4*>(jb.src + (size_t)(r0 + r) * jb.ld + c0 + 4 * tx);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[r][4 * tx + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = ty + 16 * i;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = tile[4 * tx + e][c];
        *reinterpret_cast<f32x4*>(jb.dst + (size_t)(c0 + c) * jb.rows + r0 + 4 * tx) = v;
    }
}

// zero up to 8 float buffers of n floats each (n % 4 == 0) in one launch (recurrent-state resets)
struct ZeroList { float* p[8]; int count; };
__global__ __launch_bounds__(256) void zero_bufs_kernel(ZeroList z, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < z.count; ++k) *reinterpret_cast<f32x4*>(z.p[k] + i) = zero;
}

// attention maps of a stored forward pass, [T, B, NH, R] -> [B, T, R] averaged over the NH heads (BUTD: NH = 1)
__global__ void saved_alphas_kernel(const float* __restrict__ src, int T, int B, int NH, int R, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * T * R) return;
    const int r = i % R, t = (i / R) % T, b = i / (R * T);
    const float* p = src + ((size_t)(t * B + b) * NH) * R + r;
    float s = 0.f;
    for (int h = 0; h < NH; ++h) s += p[(size_t)h * R];
    out[i] = s / (float)NH;
}

// start of a greedy chain: first input token <sta>; the per-step counters of unfinished rows (SCST baseline, may be null) = 0
__global__ void greedy_init_kernel(int64_t* it, int rows, int* n_unfinished, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < rows) it[i] = 1;
    if (n_unfinished && i < T) n_unfinished[i] = 0;
}
__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace
}  // namespace icz

// =========================================================================================================
// Training-mode kernels: multinomial epilogue, REINFORCE / XE loss gradients, BPTT pointwise parts.
// =========================================================================================================
namespace icz {
namespace {   // internal linkage: this header is included by several translation units

// block-wide helpers (256 threads)
__device__ __forceinline__ float block_max_256(float v, float* sm) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum_256(float v, float* sm) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    __syncthreads();
    return r;
}

// ---------------------------------------------------------------------------------------------------------
// sample_rl epilogue (:221-233): logp = log_softmax(logits); draw ~ multinomial(exp(logp)) by inverse CDF
// (smallest i with cumsum(p)[i] > u * sum(p), float64 -- the contract shared with the oracle); store logp[draw];
// unfinished &= (draw != <end>); it = draw * unfinished.  A row block = one workgroup; each thread owns a
// contiguous slice of the vocabulary so the global "first index above target" is the minimum over threads.
// Steps after every row has finished are left zero, as the reference's early break does (:233).
struct SampleSelArgs {
    const float* logits; int V; int ldl;
    const float* uniforms;        // [rows] for this step or null (Philox)
    const uint64_t* seed_p; int t; int T;
    uint8_t* unfinished;          // [rows] in/out
    int* n_unfinished;            // [T] counters (zeroed before the rollout)
    int64_t* seq_out; float* logp_out;   // [rows, T]
    int64_t* it_next;             // [rows]
    int32_t* draw_out;            // [rows] raw draw (for backward)
    float* lse_out;               // [rows] max + log(sum exp) (for backward)
    // optional fused epilogue: the next step's input embedding emb_next[row,:] = drop(relu(table[it_next[row],:])) (the
    // embed_kernel of step t + 1, :77-81) with that step's dropout configuration
    const float* emb_table; float* emb_next; int E; DropCfg emb_drop;
    // ns > 1: `logits` holds the ns split-K slabs of the predict GEMM (slab z at + z * slab_stride, no bias yet); the kernel
    // sums them in slab order, adds bias[v] and leaves the finished row in logits_store (the saved logits of backward)
    int ns; size_t slab_stride; const float* bias; float* logits_store;
    int* live_rows;               // optional: (number of steps the reference ran so far) x rows -- the (t, b) rows the batched GEMMs of
                                  // the backward pass need to read (GemmArgs::rows_live, embed_grad_kernel)
    // merged chain (row0 > 0, sample_select_kernel<true>): rows < row0 are the greedy baseline's
    int row0; int64_t* ids_out; uint8_t* g_unfinished; int* n_any;
};
constexpr int SEL_THREADS = 1024;       // 16 waves per row: the row (40 KB) sits in LDS
__device__ __forceinline__ float block_max_n(float v, float* sm, int nw) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int w = 1; w < nw; ++w) r = fmaxf(r, sm[w]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_sum_n(float v, float* sm, int nw) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sm[0];
    for (int w = 1; w < nw; ++w) r += sm[w];
    __syncthreads();
    return r;
}
// Inverse-CDF draw over the probabilities in LDS (the sampler contract of torch.multinomial's stand-in, oracle/butd.py
// inverse_cdf_draw): smallest v with cumsum(p)[v] > u * sum(p), sums in float64.  Every thread of the SEL_THREADS-wide
// workgroup calls it; the result (clamped to V - 1) is valid in thread 0 after the call.
__device__ __forceinline__ int block_inverse_cdf(const float* srow, int V, float u, double* smd, int* smi) {
    constexpr int NW = SEL_THREADS / 64;
    const int tid = threadIdx.x;
    // contiguous slice per thread -> the global "first index above target" is the minimum over threads
    const int per = (V + SEL_THREADS - 1) / SEL_THREADS;
    const int v0 = min(V, tid * per), v1 = min(V, v0 + per);
    double loc = 0.0;
    for (int v = v0; v < v1; ++v) loc += (double)srow[v];
    // exclusive prefix over the slice sums: wave-level inclusive scan (shuffles) + the wave totals
    double inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(inc, o, 64);
        if ((tid & 63) >= o) inc += up;
    }
    if ((tid & 63) == 63) smd[tid >> 6] = inc;
    __syncthreads();
    double wave_off = 0.0, total = 0.0;
    for (int w = 0; w < NW; ++w) {
        if (w < (tid >> 6)) wave_off += smd[w];
        total += smd[w];
    }
    const double prefix = wave_off + inc - loc;
    const double target = (double)u * total;
    int cand = 0x7fffffff;
    {
        double run = prefix;
        for (int v = v0; v < v1; ++v) {
            run += (double)srow[v];
            if (run > target) { cand = v; break; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if ((tid & 63) == 0) smi[tid >> 6] = cand;
    __syncthreads();
    int d = 0;
    if (tid == 0) {          // only thread 0 reads smi[]: the caller may reuse it right away
        d = smi[0];
        for (int w = 1; w < NW; ++w) d = min(d, smi[w]);
        if (d > V - 1) d = V - 1;
    }
    return d;
}

// Scheduled sampling, DecoderRNN.forward (BUTD_Model.py:120-130): from time step 2 on, row b of the XE forward feeds a draw
// from softmax(logits of step t-1) instead of its caption token when gate < ss_prob.  One workgroup per active row; a row
// whose gate is not below ss_prob leaves at once (its token stays the caption's).
struct SsSelArgs {
    const float* logits_prev;  // [rows, ldl] logits of step t - 1
    int ldl, V, t;
    float ss_prob;
    const float* gate;         // [B] explicit uniforms for this step, or nullptr -> Philox (RNG_SS_GATE)
    const float* draw;         // [B] explicit uniforms for the inverse-CDF draw, or nullptr -> Philox (RNG_SS_DRAW)
    const uint64_t* seed_p;
    int64_t* tok;              // [B] tokens fed at step t (in: caption tokens; out: mixed)
};
__global__ __launch_bounds__(SEL_THREADS) void ss_select_kernel(SsSelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float srow[];
    __shared__ float smf[16];
    __shared__ double smd[16];
    __shared__ int smi[16];
    constexpr int NW = SEL_THREADS / 64;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float g = a.gate ? a.gate[row] : rng_uniform(*a.seed_p, (uint32_t)a.t, (uint64_t)row, RNG_SS_GATE);
    if (!(g < a.ss_prob)) return;
    const float* l = a.logits_prev + (size_t)row * a.ldl;
    float mx = -INFINITY;
    for (int v = tid; v < a.V; v += SEL_THREADS) { const float x = l[v]; srow[v] = x; mx = fmaxf(mx, x); }
    mx = block_max_n(mx, smf, NW);
    float se = 0.f;
    for (int v = tid; v < a.V; v += SEL_THREADS) { const float e = expf(srow[v] - mx); srow[v] = e; se += e; }
    se = block_sum_n(se, smf, NW);
    for (int v = tid; v < a.V; v += SEL_THREADS) srow[v] = srow[v] / se;                  // torch.softmax
    __syncthreads();
    const float u = a.draw ? a.draw[row] : rng_uniform(*a.seed_p, (uint32_t)a.t, (uint64_t)row, RNG_SS_DRAW);
    const int d = block_inverse_cdf(srow, a.V, u, smd, smi);
    if (tid == 0) a.tok[row] = d;
}
// tokens of XE step t (t >= 2) for the `rows` active rows; gate / draw: explicit [T, B] arrays or nullptr (Philox)
inline int ss_select_launch(hipStream_t st, int rows, const float* logits_prev, int ldl, int V, int t, int B, float ss_prob,
                            const float* gate, const float* draw, const uint64_t* seed_p, int64_t* tok_t) {
    static bool lds_set = false;
    if (!lds_set) {
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ss_select_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        lds_set = true;
    }
    ICZ_REQUIRE(sizeof(float) * (size_t)V <= 159 * 1024, "scheduled sampling: a row of %d logits does not fit in LDS", V);
    SsSelArgs sa = {};
    sa.logits_prev = logits_prev; sa.ldl = ldl; sa.V = V; sa.t = t; sa.ss_prob = ss_prob;
    sa.gate = gate ? gate + (size_t)t * B : nullptr;
    sa.draw = draw ? draw + (size_t)t * B : nullptr;
    sa.seed_p = seed_p; sa.tok = tok_t;
    hipLaunchKernelGGL(ss_select_kernel, dim3(rows), dim3(SEL_THREADS), sizeof(float) * V, st, sa);
    return ICZ_OK;
}
// One workgroup of 16 waves per row.  Passes: (1) finished logits x = sum of the predict GEMM's split-K slabs + bias -> LDS (and
// the saved-logits slot of backward), row maximum M; (2) every thread takes a contiguous slice of the row: float32 sum of
// exp(x - M) -> lse, then the float64 sum of p_v = expf((x_v - M) - lse) over its slice and a block scan -> total and the slice's
// prefix; (3) the thread whose slice crosses u * total walks it again (p recomputed: the same float expression) and reports the
// first index above the target.  Round 2's kernel made two more passes over the row (p written back to LDS, strided re-reads).
// MEASURED (round 3, same box): a two-launch form over the whole chip (rows x 16 slice workgroups for the sums, then one small
// workgroup per row for the draw: 7.9 + 8.1 us against 15.4 us for round 2's kernel) made the SCST rollouts SLOWER, 2.83 -> 3.09 ms:
// the sampled chain runs beside the greedy chain, and a launch that fills every CU stalls the other chain's kernels, while this
// one leaves three quarters of the chip to them.
// MG = true: the merged chain of a small SCST batch (Butd::merged_chain): rows [0, row0) are the GREEDY baseline's rows (evaluation
// mode: argmax with ties to the lowest index, BUTD_Model.py:183, next embedding without dropout, ids to ids_out), rows >= row0 the
// sampled rollout's (row - row0 of every per-row array of the sampled batch).  Two counters: n_unfinished[t] = sampled rows still
// unfinished (the reference's break, :233: behind it seq / logp are zeros and BPTT has nothing to do), n_any[t] = those + the
// greedy rows that have not emitted <end> (0 = no kernel of the remaining steps runs).
template <bool MG>
__global__ __launch_bounds__(SEL_THREADS) void sample_select_kernel(SampleSelArgs a) {
    extern __shared__ __attribute__((aligned(16))) float srow[];     // V floats: the finished logits of the row
    __shared__ float smf[16];
    __shared__ double smd[16];
    __shared__ int smi[16];
    constexpr int NW = SEL_THREADS / 64;
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sr = MG ? row - a.row0 : row;                 // row of the sampled batch; < 0: a greedy row of the merged chain
    const bool greedy_row = MG && sr < 0;
    const float sc = a.emb_drop.mode ? 2.0f : 1.0f;
    const int* const gate = MG ? a.n_any : a.n_unfinished;
    const bool all_dead = a.t > 0 && gate[a.t - 1] == 0;    // the kernels of this step returned at entry (step_dead)
    if (all_dead || (MG && !greedy_row && a.t > 0 && a.n_unfinished[a.t - 1] == 0)) {
        // every (sampled) row has finished: zeros, as the reference's early break leaves them (:233)
        if (tid == 0) {
            if (greedy_row) a.ids_out[(size_t)row * a.T + a.t] = 0;
            else { a.seq_out[(size_t)sr * a.T + a.t] = 0; a.logp_out[(size_t)sr * a.T + a.t] = 0.f; }
            a.it_next[row] = 0;
            a.draw_out[row] = -1;
            a.lse_out[row] = 0.f;
        }
        if (MG && !all_dead && a.emb_next)                  // the greedy half goes on: this row keeps running on <pad> (finite, never read)
            for (int e = tid * 4; e < a.E; e += 4 * SEL_THREADS) {
                f32x4 x = *reinterpret_cast<const f32x4*>(a.emb_table + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = fmaxf(x[j], 0.f);
                *reinterpret_cast<f32x4*>(a.emb_next + (size_t)row * a.E + e) = x;
            }
        return;
    }
    if (a.live_rows && sr == 0 && tid == 0) *a.live_rows = (a.t + 1) * (int)gridDim.x;
    const float* l = a.logits + (size_t)row * a.ldl;
    const float u = greedy_row ? 0.f : (a.uniforms ? a.uniforms[sr] : rng_uniform(*a.seed_p, (uint32_t)a.t, (uint64_t)sr));
    const bool was_unf = greedy_row ? (a.t == 0 || a.g_unfinished[row] != 0) : a.unfinished[sr] != 0;   // loaded early: the tail below is a chain of dependent accesses
    // pass 1: one coalesced pass over HBM / L2; every later pass runs out of LDS
    float mx = -INFINITY;
    if (a.ns > 1) {
        float* ls = a.logits_store + (size_t)row * a.ldl;
        const bool vec = ((a.ldl | (int)(a.slab_stride & 3)) & 3) == 0 &&
                         (((uintptr_t)a.logits | (uintptr_t)a.bias | (uintptr_t)a.logits_store) & 15) == 0;
        const int Vv = vec ? (a.V & ~3) : 0;
        for (int v = tid * 4; v < Vv; v += 4 * SEL_THREADS) {
            f32x4 x = *reinterpret_cast<const f32x4*>(l + v);
            for (int z = 1; z < a.ns; ++z) x += *reinterpret_cast<const f32x4*>(l + (size_t)z * a.slab_stride + v);
            x += *reinterpret_cast<const f32x4*>(a.bias + v);
            *reinterpret_cast<f32x4*>(ls + v) = x;
            *reinterpret_cast<f32x4*>(srow + v) = x;
            mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
        }
        for (int v = Vv + tid; v < a.V; v += SEL_THREADS) {
            float x = l[v];
            for (int z = 1; z < a.ns; ++z) x += l[(size_t)z * a.slab_stride + v];
            x += a.bias[v];
            ls[v] = x; srow[v] = x; mx = fmaxf(mx, x);
        }
    } else {
        for (int v = tid; v < a.V; v += SEL_THREADS) { const float x = l[v]; srow[v] = x; mx = fmaxf(mx, x); }
    }
    mx = block_max_n(mx, smf, NW);
    if (greedy_row) {
        // argmax, ties to the lowest index: the first element of the row that equals its maximum
        const int per = (a.V + SEL_THREADS - 1) / SEL_THREADS;
        const int v0 = min(a.V, tid * per), v1 = min(a.V, v0 + per);
        int cand = 0x7fffffff;
        for (int v = v0; v < v1; ++v)
            if (srow[v] == mx) { cand = v; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        if (lane == 0) smi[wave] = cand;
        __syncthreads();
        int bi = smi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) bi = min(bi, smi[w]);
        if ((unsigned)bi >= (unsigned)a.V) bi = 0;      // a row of NaN logits (diverged training): <pad>, not a wild read
        if (tid == 0) {
            const bool unf = was_unf && bi != 2;
            a.g_unfinished[row] = unf ? 1 : 0;
            a.ids_out[(size_t)row * a.T + a.t] = bi;
            a.it_next[row] = bi;
            a.draw_out[row] = -1;                       // no gradient flows into an evaluation-mode row (reinforce_dlogits_kernel)
            a.lse_out[row] = 0.f;
            if (unf) atomicAdd(&a.n_any[a.t], 1);
        }
        if (a.emb_next)
            for (int e = tid * 4; e < a.E; e += 4 * SEL_THREADS) {
                f32x4 x = *reinterpret_cast<const f32x4*>(a.emb_table + (size_t)bi * a.E + e);
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = fmaxf(x[j], 0.f);
                *reinterpret_cast<f32x4*>(a.emb_next + (size_t)row * a.E + e) = x;
            }
        return;
    }
    // pass 2: contiguous slice per thread -> the first index above the target is the minimum over threads.
    // p_v = exp(log_softmax(x)_v) = expf((x_v - M) - lse) with lse = logf(sum expf(x - M)) in float32 -- the reference's own
    // float expression (:221-223), so that the float64 CDF below agrees with the oracle's to the last bit in all but a few draws in 10^5
    const int per = (a.V + SEL_THREADS - 1) / SEL_THREADS;
    const int v0 = min(a.V, tid * per), v1 = min(a.V, v0 + per);
    float se = 0.f;
    for (int v = v0; v < v1; ++v) se += expf(srow[v] - mx);
    se = block_sum_n(se, smf, NW);
    const float lse = logf(se);
    double loc = 0.0;
    for (int v = v0; v < v1; ++v) loc += (double)expf((srow[v] - mx) - lse);
    double inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double up = __shfl_up(inc, o, 64);
        if (lane >= o) inc += up;
    }
    if (lane == 63) smd[wave] = inc;
    __syncthreads();
    double wave_off = 0.0, total = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const double t = smd[w];
        if (w < wave) wave_off += t;
        total += t;
    }
    const double target = (double)u * total;
    // pass 3: only the slice that crosses the target is walked again
    int cand = 0x7fffffff;
    {
        // [lower, upper) of this thread's slice: lower IS the previous lane's upper (the scan's own value, shuffled -- not
        // inc - loc, which can differ from it in the last place and leave a one-ulp gap that no thread claims: the draw then fell
        // through to V - 1); across waves the lower bound of lane 0 is the same sum of wave totals as the previous wave's upper
        const double prev = __shfl_up(inc, 1, 64);
        double run = wave_off + (lane ? prev : 0.0);
        if (run <= target && wave_off + inc > target) {
            cand = v1 - 1;                               // the slice's last element carries the scan's own cumulative value
            for (int v = v0; v < v1 - 1; ++v) {
                run += (double)expf((srow[v] - mx) - lse);
                if (run > target) { cand = v; break; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
    if (lane == 0) smi[wave] = cand;
    __syncthreads();
    int tok = 0;
    {
        int d = smi[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) d = min(d, smi[w]);
        if (d > a.V - 1) d = a.V - 1;                // rounding at u -> 1: the last token
        const bool unf = was_unf && (d != 2);
        tok = unf ? d : 0;
        if (tid == 0) {
            a.unfinished[sr] = unf ? 1 : 0;
            a.seq_out[(size_t)sr * a.T + a.t] = tok;
            a.logp_out[(size_t)sr * a.T + a.t] = (srow[d] - mx) - lse;
            a.it_next[row] = tok;
            a.draw_out[row] = d;
            a.lse_out[row] = mx + lse;
            if (unf) {
                atomicAdd(&a.n_unfinished[a.t], 1);
                if (MG) atomicAdd(&a.n_any[a.t], 1);
            }
        }
    }
    if (a.emb_next)
        for (int e = tid * 4; e < a.E; e += 4 * SEL_THREADS) {
            f32x4 x = *reinterpret_cast<const f32x4*>(a.emb_table + (size_t)tok * a.E + e);
            const uint32_t k = a.emb_drop.mode ? a.emb_drop.keep4((uint64_t)sr * a.E + e) : 0xFu;
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = ((k >> j) & 1u) ? fmaxf(x[j], 0.f) * sc : 0.f;
            *reinterpret_cast<f32x4*>(a.emb_next + (size_t)row * a.E + e) = x;
        }
}
inline void launch_sample_select(hipStream_t st, int rows, const SampleSelArgs& a) {
    if (a.row0 > 0) hipLaunchKernelGGL(sample_select_kernel<true>, dim3(rows), dim3(SEL_THREADS), sizeof(float) * a.V, st, a);
    else hipLaunchKernelGGL(sample_select_kernel<false>, dim3(rows), dim3(SEL_THREADS), sizeof(float) * a.V, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// RewardCriterion (Utils.py:295-317): mask[b,0] = 1, mask[b,t] = (seq[b,t-1] > 0);
//   loss = -sum(logp * reward * mask) / sum(mask).   One workgroup; also emits coef[b,t] = -reward*mask/denom
// (= d loss / d logp) for the gradient kernel.  denom = mask_sum_global if > 0 else the local mask sum.
__global__ __launch_bounds__(256) void reinforce_loss_kernel(const float* __restrict__ logp, const int64_t* __restrict__ seq,
                                                             const float* __restrict__ reward, int B, int T,
                                                             const float* __restrict__ mask_sum_global_p, float* __restrict__ coef,
                                                             float* __restrict__ loss_out, float* __restrict__ mask_sum_out) {
    __shared__ float smf[4];
    const int tid = threadIdx.x;
    float ms = 0.f, ls = 0.f;
    for (int i = tid; i < B * T; i += 256) {
        const int t = i % T;
        const float m = (t == 0) ? 1.f : (seq[i - 1] > 0 ? 1.f : 0.f);
        ms += m;
        ls += -logp[i] * reward[i] * m;
    }
    ms = block_sum_256(ms, smf);
    ls = block_sum_256(ls, smf);
    const float msg = mask_sum_global_p ? mask_sum_global_p[0] : 0.f;
    const float denom = msg > 0.f ? msg : ms;
    for (int i = tid; i < B * T; i += 256) {
        const int t = i % T;
        const float m = (t == 0) ? 1.f : (seq[i - 1] > 0 ? 1.f : 0.f);
        coef[i] = -reward[i] * m / denom;
    }
    if (tid == 0) {
        if (loss_out) loss_out[0] = ls / denom;
        if (mask_sum_out) mask_sum_out[0] = ms;
    }
}

// d loss / d logits for the whole rollout (row = t*B + b):
//   dlogits[row,v] = coef[b,t] * (1[v == draw] - softmax(logits)[v])         (log_softmax + gather backward)
// Written in place over the saved logits; pad columns [V, ldl) are zeroed.
__global__ __launch_bounds__(256) void reinforce_dlogits_kernel(float* __restrict__ logits, int V, int ldl,
                                                                const int32_t* __restrict__ draw, const float* __restrict__ lse,
                                                                const float* __restrict__ coef, int B, int T, int Bs = 0, int row0 = 0) {
    // Bs > 0: the slots hold Bs rows per step, of which rows [row0, row0 + B) are the sampled rollout's (merged chain); the rows in
    // front are evaluation-mode rows (draw = -1): zero gradient
    const int row = blockIdx.y;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= ldl) return;
    const int stride = Bs > 0 ? Bs : B;
    const int t = row / stride, b = row % stride - row0;
    float* l = logits + (size_t)row * ldl;
    const float c = b >= 0 ? coef[(size_t)b * T + t] : 0.f;
    const int d = draw[row];
    float g = 0.f;
    if (v < V && d >= 0 && c != 0.f) g = c * ((v == d ? 1.f : 0.f) - expf(l[v] - lse[row]));
    l[v] = g;
}

// ---------------------------------------------------------------------------------------------------------
// LabelSmoothingLoss (Utils.py:268-286) on one time step of the XE forward (rows = rows active at t):
//   loss_row = sum_v true_v (log true_v - logp_v),  true = 1-s at the target, s/(V-1) elsewhere
//   dlogits  = (softmax - true) / N_tokens          (KLDiv(log_softmax) backward)
// One workgroup per row; per-row loss goes to loss_rows (summed later in fixed order).
// All time steps in one launch: grid (B, T), block (b, t) = row t * B + b of the saved logits, active while b < rows.n[t] (the
// captions are sorted by decreasing length, Engine.py:178); target = captions[b, t + 1].  One launch per step left the chip
// to at most B workgroups seventeen times per XE step.
constexpr int XE_MAX_T = 128;
struct XeRows { int n[XE_MAX_T]; };
__global__ __launch_bounds__(256) void xe_loss_dlogits_kernel(float* __restrict__ logits, int V, int ldl,
                                                              const int64_t* __restrict__ captions, int L, int B, XeRows rows,
                                                              float smoothing, float inv_n_host, const float* __restrict__ n_dev,
                                                              float* __restrict__ loss_rows) {
    __shared__ float smf[4];
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    if (b >= rows.n[t]) {          // inactive (t, b): its gradient row must be zero for the batched GEMMs of the backward pass (the batched
        float* z = logits + (size_t)(t * B + b) * ldl;      // vocabulary projection of the forward pass fills every row)
        for (int v = tid; v < ldl; v += 256) z[v] = 0.f;
        return;
    }
    // data-parallel: the all-reduced token count, a device scalar (0 = never handed over: the local count, as the SCST path does)
    const float inv_n = (n_dev && n_dev[0] > 0.f) ? 1.0f / n_dev[0] : inv_n_host;
    const int row = t * B + b;
    float* l = logits + (size_t)row * ldl;
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 256) mx = fmaxf(mx, l[v]);
    mx = block_max_256(mx, smf);
    float se = 0.f;
    for (int v = tid; v < V; v += 256) se += expf(l[v] - mx);
    se = block_sum_256(se, smf);
    const float lse = mx + logf(se);
    const int tg = (int)captions[(size_t)b * L + t + 1];
    const float conf = 1.f - smoothing, low = smoothing / (float)(V - 1);
    const float lconf = conf > 0.f ? logf(conf) : 0.f, llow = low > 0.f ? logf(low) : 0.f;
    float ls = 0.f;
    for (int v = tid; v < ldl; v += 256) {
        float g = 0.f;
        if (v < V) {
            const float lp = l[v] - lse;
            const float tr = (v == tg) ? conf : low;
            if (tr > 0.f) ls += tr * (((v == tg) ? lconf : llow) - lp);
            g = (expf(lp) - tr) * inv_n;
        }
        l[v] = g;
    }
    ls = block_sum_256(ls, smf);
    if (tid == 0) loss_rows[row] = ls;
}

// sum of n floats in fixed order by one workgroup, scaled
__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ x, int n, float scale_host, const float* __restrict__ n_dev,
                                                        float* __restrict__ out) {
    __shared__ float smf[4];
    const float scale = (n_dev && n_dev[0] > 0.f) ? 1.0f / n_dev[0] : scale_host;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = block_sum_256(s, smf);
    if (threadIdx.x == 0) out[0] = s * scale;
}

// ---------------------------------------------------------------------------------------------------------
// LSTMCell backward, pointwise part.  dh = sum of up to three slab sets (+ optional dropped-output term),
// dc = dc_in (optional).  Emits d(pre-activation gates) [rows,4H] and dc_prev.
struct LstmBwdArgs {
    const float* dh_a; int ns_a;      // slabs [ns][rows][lda_a] read at column offset 0..H
    int lda_a;
    const float* dh_b; int ns_b; int lda_b;
    const float* dh_c; int ns_c; int lda_c;
    const float* dhdrop;              // [rows,H] gradient w.r.t. the dropped copy of h (predict input) or null
    const float* dc_in;               // [rows,H] or null
    const float* gates;               // [rows,4H] activated i,f,g,o
    const float* c_prev;              // [rows,H] or null (= zeros)
    const float* c_cur;               // [rows,H]
    float* dgates;                    // [rows,4H]
    float* dc_prev;                   // [rows,H]
    int rows, H;
    int rows_a, rows_b, rows_c;       // row counts of the slab sets (slab stride = rows_x * lda_x)
    int dc_in_rows;                   // valid rows of dc_in
    const int* live;                  // step_dead(live): this step never ran -> its dgates rows are ZERO (the weight-gradient GEMMs read
                                      // every (t, b) row) and nothing else is touched
    const int* carry_live;            // step_dead(carry_live): the step behind this one (t + 1) never ran -> no dh_a, no dc_in
};
// grid (H/256, rows): one hidden unit per thread (see lstm_point_kernel)
__global__ __launch_bounds__(256) void lstm_bwd_point_kernel(LstmBwdArgs a, DropCfg dc) {
    const int row = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= a.H) return;
    const int H = a.H, G = 4 * H;
    // loads that do not depend on the step's liveness first: the two flags are tested behind them (live_flag / flag_dead, icz_common.h)
    const int lflag = live_flag(a.live), cflag = live_flag(a.carry_live);
    const float* g = a.gates + (size_t)row * G + j;
    const float gi = g[0], gf = g[H], gg = g[2 * H], go = g[3 * H];
    const float cc = a.c_cur[(size_t)row * H + j];
    const float cp = a.c_prev ? a.c_prev[(size_t)row * H + j] : 0.f;
    float dhd = 0.f;
    if (a.dhdrop) dhd = a.dhdrop[(size_t)row * H + j];
    if (flag_dead(lflag)) {
        float* o = a.dgates + (size_t)row * G + j;
        o[0] = 0.f; o[H] = 0.f; o[2 * H] = 0.f; o[3 * H] = 0.f;
        return;
    }
    const bool carry = cflag != 0;
    // a slab set only holds rows_x rows (XE: the batch shrinks with t); rows beyond contribute zero
    float dh = 0.f;
    if (a.dh_a && carry && row < a.rows_a) dh += sum_slabs1(a.dh_a, a.ns_a, (size_t)a.rows_a * a.lda_a, (size_t)row * a.lda_a + j);
    if (a.dh_b && row < a.rows_b) dh += sum_slabs1(a.dh_b, a.ns_b, (size_t)a.rows_b * a.lda_b, (size_t)row * a.lda_b + j);
    if (a.dh_c && row < a.rows_c) dh += sum_slabs1(a.dh_c, a.ns_c, (size_t)a.rows_c * a.lda_c, (size_t)row * a.lda_c + j);
    if (a.dhdrop) {
        float d = dhd;
        if (dc.mode) d = dc.keep((uint64_t)row * H + j) ? d * 2.0f : 0.f;
        dh += d;
    }
    const float dcin = (a.dc_in && carry && row < a.dc_in_rows) ? a.dc_in[(size_t)row * H + j] : 0.f;
    const float tc = tanhf(cc);
    const float dcv = dcin + dh * go * (1.f - tc * tc);
    float* o = a.dgates + (size_t)row * G + j;
    o[0] = dcv * gg * gi * (1.f - gi);
    o[H] = dcv * cp * gf * (1.f - gf);
    o[2 * H] = dcv * gi * (1.f - gg * gg);
    o[3 * H] = dh * tc * go * (1.f - go);
    a.dc_prev[(size_t)row * H + j] = dcv * gf;
}

// ---------------------------------------------------------------------------------------------------------
// SoftAttention backward in three full-chip kernels (each moves its bytes once, with many loads in flight per lane):
//   per step   att_bwd_dalpha_kernel   dalpha[row,r] = dctx[row,:] . feats[row,r,:]
//   per step   att_bwd_ddec_kernel     ds = softmax'(alpha, dalpha);  ddec[row,a] = sum_r on(row,r,a) ds_r w_aff[a] scale
//   once       att_bwd_denc_kernel     denc[row,r,a] = sum_t on_t ds_t w_aff scale;  dwaff partials   (after the time loop)
// with zpre = enc_ctx[row,r,a] + dec_ctx_t[row,a] and on = zpre > 0 && keep-bit (relu + dropout).  enc_ctx is shared by all
// time steps, so its gradient is a sum over t: accumulating it step by step would read-modify-write R*A floats per row
// and step; instead the steps only record ds_t (R floats per row) and the sum over t is formed once, in registers.

// dctx = sum of slabs [ns][rows][ldc] columns 0..D (the LM-LSTM dgrad GEMM output).  Grid (rows, parts = ceil(D / 512)), 256 threads:
// a workgroup sums the slabs of ITS 512 columns only (with 16 slabs per step the slab sum is as many bytes as the features: every
// part summing the whole row, as before round 3, quadrupled it) and emits partial dot products dalpha_part[row][part][r] over
// those columns for all regions; att_bwd_ddec_kernel adds the parts in order.  Wave w takes regions w, w + 4, ...: two 16-byte
// feature loads per lane and region, all of a wave's loads independent.
constexpr int DALPHA_COLS = 512;
__global__ __launch_bounds__(256) void att_bwd_dalpha_kernel(const float* __restrict__ dctx, int ns, int ldc, int rows,
                                                             const float* __restrict__ feats, int R, int D,
                                                             float* __restrict__ dalpha_part, const int* __restrict__ live) {
    __shared__ __attribute__((aligned(16))) float sd[DALPHA_COLS];
    const int lflag = live_flag(live);
    const int row = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c0 = part * DALPHA_COLS;
    const size_t ss = (size_t)rows * ldc;
    constexpr int NR = 5;                         // regions per wave and pass (R <= 64: at most 16 per wave)
    const int ca = c0 + 4 * lane, cb = ca + 256;
    const bool va = ca < D, vb = cb < D;
    const float* frow = feats + (size_t)row * R * D;
    f32x4 xa[NR], xb[NR];
    auto load_pass = [&](int r0) {                // clamped: a pass past the last region re-reads it and is not stored
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = min(r0 + 4 * u, R - 1);
            xa[u] = *reinterpret_cast<const f32x4*>(frow + (size_t)r * D + (va ? ca : 0));
            xb[u] = *reinterpret_cast<const f32x4*>(frow + (size_t)r * D + (vb ? cb : 0));
        }
    };
    load_pass(wave);                              // in flight behind the slab sum
    if (flag_dead(lflag)) return;                 // behind the first loads, in front of the first write (icz_common.h)
    if (tid < DALPHA_COLS / 4) {
        const int c = c0 + 4 * tid;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c < D) v = sum_slabs4(dctx, ns, ss, (size_t)row * ldc + c);
        *reinterpret_cast<f32x4*>(sd + 4 * tid) = v;
    }
    __syncthreads();
    const f32x4 ga = *reinterpret_cast<const f32x4*>(sd + 4 * lane), gb = *reinterpret_cast<const f32x4*>(sd + 256 + 4 * lane);
    for (int r0 = wave; r0 < R; r0 += 4 * NR) {
        if (r0 != wave) load_pass(r0);
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            float acc = 0.f;
            if (va) acc += xa[u][0] * ga[0] + xa[u][1] * ga[1] + xa[u][2] * ga[2] + xa[u][3] * ga[3];
            if (vb) acc += xb[u][0] * gb[0] + xb[u][1] * gb[1] + xb[u][2] * gb[2] + xb[u][3] * gb[3];
            acc = wave_sum(acc);
            const int r = r0 + 4 * u;
            if (lane == 0 && r < R) dalpha_part[((size_t)row * nparts + part) * R + r] = acc;
        }
    }
}

// Grid (rows, A/256), 256 threads: thread (wq = wave, cg) sums regions wq, wq+4, ... for 4 consecutive attention columns;
// the four waves meet in LDS.  ds_out (block y = 0) keeps ds for att_bwd_denc_kernel.
struct AttBwdDdecArgs {
    const float* enc_ctx; const float* dec_ctx; const float* w_aff; const float* alpha; const float* dalpha;   // dalpha: [rows][nparts][R] partials
    float* ddec; float* ds_out;
    int R, A, nparts;
    const int* live;              // step_dead(live): ddec and ds rows of this step are ZERO (read by the GEMMs / kernels over all steps)
};
__global__ __launch_bounds__(256) void att_bwd_ddec_kernel(AttBwdDdecArgs a, DropCfg dc) {
    __shared__ float sds[64];
    __shared__ __attribute__((aligned(16))) float spart[3 * 64 * 4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wq = tid >> 6;
    const int lflag = live_flag(a.live);
    const float al = lane < a.R ? a.alpha[(size_t)row * a.R + lane] : 0.f;
    const float da0 = lane < a.R ? a.dalpha[(size_t)row * a.nparts * a.R + lane] : 0.f;
    if (flag_dead(lflag)) {                       // behind the first loads (icz_common.h); a dead step's d dec / ds rows are zeros
        if (blockIdx.y == 0 && tid < a.R) a.ds_out[(size_t)row * a.R + tid] = 0.f;
        const int c = blockIdx.y * 256 + tid;
        if (c < a.A) a.ddec[(size_t)row * a.A + c] = 0.f;
        return;
    }
    float da = da0;
    if (lane < a.R)
        for (int p = 1; p < a.nparts; ++p) da += a.dalpha[((size_t)row * a.nparts + p) * a.R + lane];
    const float dot = wave_sum(al * da);
    const float ds_l = al * (da - dot);
    if (tid < 64) {
        sds[tid] = ds_l;
        if (blockIdx.y == 0 && tid < a.R) a.ds_out[(size_t)row * a.R + tid] = ds_l;
    }
    __syncthreads();
    const int c = blockIdx.y * 256 + lane * 4;
    const bool valid = c < a.A;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
        const f32x4 d = *reinterpret_cast<const f32x4*>(a.dec_ctx + (size_t)row * a.A + c);
        const f32x4 w = *reinterpret_cast<const f32x4*>(a.w_aff + c);
        const float sc = dc.mode ? 2.0f : 1.0f;
        constexpr int RB = 9;
        const bool shared_bits = dc.mode == 2 && (a.A & 255) == 0;      // every lane of the wave is valid and takes every iteration
        for (int r0 = wq; r0 < a.R; r0 += 4 * RB) {
            f32x4 x[RB];
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = min(r0 + 4 * u, a.R - 1);
                x[u] = *reinterpret_cast<const f32x4*>(a.enc_ctx + ((size_t)row * a.R + r) * a.A + c);
            }
            // Philox keep-bits, shared as in att_scores_kernel: lane 2 u + half computes the block of region u's half-strip
            uint32_t kq[RB];
            if (shared_bits) {
                const int pu = min(lane >> 1, RB - 1), ph = lane & 1;
                const int pr = min(r0 + 4 * pu, a.R - 1);
                const uint64_t g = (((uint64_t)row * a.R + pr) * a.A + blockIdx.y * 256 + 128 * ph) >> 7;
                const uint64_t seed = *dc.seed_p;
                uint4_ ctr = {(uint32_t)g, (uint32_t)(g >> 32), dc.step, dc.stream};
                const uint4_ pw = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
                for (int u = 0; u < RB; ++u) {
                    const int src = 2 * u + (lane >> 5);
                    const uint32_t w0 = __shfl(pw.x, src), w1 = __shfl(pw.y, src), w2 = __shfl(pw.z, src), w3 = __shfl(pw.w, src);
                    const int wi = (lane >> 3) & 3;
                    const uint32_t ww = wi == 0 ? w0 : (wi == 1 ? w1 : (wi == 2 ? w2 : w3));
                    kq[u] = (ww >> ((4 * lane) & 31)) & 0xFu;
                }
            }
#pragma unroll
            for (int u = 0; u < RB; ++u) {
                const int r = r0 + 4 * u;
                if (r < a.R) {
                    const float ds = sds[r];
                    const uint32_t k = shared_bits ? kq[u] : (dc.mode ? dc.keep4(((uint64_t)row * a.R + r) * a.A + c) : 0xFu);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool on = (x[u][j] + d[j] > 0.f) && ((k >> j) & 1u);
                        acc[j] += on ? ds * w[j] * sc : 0.f;
                    }
                }
            }
        }
    }
    if (wq > 0) *reinterpret_cast<f32x4*>(spart + ((wq - 1) * 64 + lane) * 4) = acc;
    __syncthreads();
    if (wq == 0 && valid) {
#pragma unroll
        for (int w = 0; w < 3; ++w) acc += *reinterpret_cast<const f32x4*>(spart + (w * 64 + lane) * 4);
        *reinterpret_cast<f32x4*>(a.ddec + (size_t)row * a.A + c) = acc;
    }
}

// Grid (B, parts), 256 threads.  Part p owns regions p, p + parts, ...; a thread owns 4 consecutive attention columns,
// keeps dec_ctx_t of up to TT time steps in registers and walks its regions: one read of enc_ctx, one write of denc.
//   denc[row,r,a]          = sum_t on_t ds_t[row,r] w_aff[a] scale
//   dwaff_part[row,p,a]    = sum_{r in p} sum_t on_t ds_t[row,r] zpre scale        (column-summed over rows x parts afterwards)
struct AttBwdDencArgs {
    const float* enc_ctx; const float* dec_all; const float* ds_all; const float* w_aff;
    float* denc; float* dwaff_part;
    int B, R, A, T;
    int mode; const uint8_t* mask; size_t mask_step; const uint64_t* seed_p; uint32_t stream;
    int Bs;          // rows per time step in dec_all / ds_all (0 = B; a merged chain stores 2 B, the pointers then start at its second half)
};
template <int TT>
__global__ __launch_bounds__(256) void att_bwd_denc_kernel(AttBwdDencArgs a) {
    static_assert(TT <= 32, "one half-wave computes the TT Philox words of its 128-column group");
    extern __shared__ __attribute__((aligned(16))) float sds_t[];     // [T][R] ds of this row
    // Philox keep-bits: one call covers 128 consecutive elements = the 32 lanes of a half-wave (4 columns each), so instead
    // of every lane calling it for every time step, lane j of a half-wave calls it for step t0 + j and the words are handed
    // round through LDS: TT calls per half-wave and region instead of 32 * TT.
    __shared__ uint32_t sbits[4][2][TT][4];
    const int row = blockIdx.x, part = blockIdx.y, nparts = gridDim.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, half = lane >> 5, hl = lane & 31;
    const int Bst = a.Bs > 0 ? a.Bs : a.B;
    for (int i = tid; i < a.T * a.R; i += 256) {
        const int t = i / a.R, r = i % a.R;
        sds_t[i] = a.ds_all[((size_t)t * Bst + row) * a.R + r];
    }
    __syncthreads();
    const float sc = a.mode ? 2.0f : 1.0f;
    const bool shared_bits = a.mode == 2 && (a.A & 127) == 0;
    for (int c0 = 0; c0 < a.A; c0 += 1024) {          // uniform over the workgroup (barriers inside)
        const int c = c0 + tid * 4;
        const bool cv = c < a.A;
        const f32x4 w = cv ? *reinterpret_cast<const f32x4*>(a.w_aff + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 dw = {0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < a.T; t0 += TT) {
            f32x4 d[TT];
#pragma unroll
            for (int j = 0; j < TT; ++j) {
                const int t = min(t0 + j, a.T - 1);
                d[j] = cv ? *reinterpret_cast<const f32x4*>(a.dec_all + ((size_t)t * Bst + row) * a.A + c) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            for (int r = part; r < a.R; r += nparts) {
                const size_t eoff = ((size_t)row * a.R + r) * a.A + c;
                if (shared_bits) {
                    __syncthreads();            // the previous region's words have been read
                    if (cv && hl < TT && t0 + hl < a.T) {
                        const uint64_t g = (eoff - (size_t)(4 * hl)) >> 7;          // the half-wave's group: its first lane's element
                        const uint64_t seed = *a.seed_p;
                        uint4_ ctr = {(uint32_t)g, (uint32_t)(g >> 32), (uint32_t)(t0 + hl), a.stream};
                        const uint4_ rr = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
                        uint32_t* o = sbits[wave][half][hl];
                        o[0] = rr.x; o[1] = rr.y; o[2] = rr.z; o[3] = rr.w;
                    }
                    __syncthreads();
                }
                if (!cv) continue;
                const f32x4 x = *reinterpret_cast<const f32x4*>(a.enc_ctx + eoff);
                f32x4 de = {0.f, 0.f, 0.f, 0.f};
                if (t0 > 0) de = *reinterpret_cast<const f32x4*>(a.denc + eoff);
#pragma unroll
                for (int j = 0; j < TT; ++j) {
                    const int t = t0 + j;
                    if (t < a.T) {
                        const float ds = sds_t[t * a.R + r];
                        uint32_t k = 0xFu;
                        if (a.mode == 1) {
                            const uint32_t m = *reinterpret_cast<const uint32_t*>(a.mask + (size_t)t * a.mask_step + eoff);
                            k = ((m & 0xFFu) ? 1u : 0u) | ((m & 0xFF00u) ? 2u : 0u) | ((m & 0xFF0000u) ? 4u : 0u) | ((m & 0xFF000000u) ? 8u : 0u);
                        } else if (shared_bits) {
                            k = (sbits[wave][half][j][(eoff >> 5) & 3] >> (eoff & 31)) & 0xFu;
                        } else if (a.mode == 2) {
                            k = (rng_group_bits(*a.seed_p, a.stream, (uint32_t)t, eoff) >> (eoff & 31)) & 0xFu;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float zp = x[q] + d[j][q];
                            const bool on = (zp > 0.f) && ((k >> q) & 1u);
                            de[q] += on ? ds * w[q] * sc : 0.f;
                            dw[q] += on ? zp * sc * ds : 0.f;
                        }
                    }
                }
                *reinterpret_cast<f32x4*>(a.denc + eoff) = de;
            }
        }
        if (cv) *reinterpret_cast<f32x4*>(a.dwaff_part + ((size_t)row * nparts + part) * a.A + c) = dw;
    }
}

// ---------------------------------------------------------------------------------------------------------
// out[n] = sum_k X[k, n] (column sums over K rows; bias gradients) in one launch, fixed order -> reproducible.
// Grid (N/32), 256 threads = 32 columns x 8 row groups: thread (c, g) adds rows g, g+8, ... of its column (eight independent
// loads in flight), the eight groups meet in LDS and are added in group order.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int K, int N, int ldx, float* __restrict__ out) {
    __shared__ float part[8][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + c;
    float s = 0.f;
    if (n < N) {
        const float* x = X + n;
        int k = g;
        for (; k + 56 < K; k += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(k + 8 * u) * ldx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < K; k += 8) s += x[(size_t)k * ldx];
    }
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && n < N) {
        float t = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += part[q][c];
        out[n] = t;
    }
}

// Several column sums in ONE launch (the bias gradients at the end of BPTT): blocks laid out job after job; out2 (optional)
// receives a copy (b_ih and b_hh have the same gradient); a job with K = 0 writes zeros.
struct ColsumJob { const float* X; int K, N, ldx; float* out; float* out2; int block0; };
struct ColsumTable { ColsumJob j[8]; int count; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(ColsumTable tab) {
    __shared__ float part[8][33];
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < tab.count; ++i)
        if ((int)blockIdx.x >= tab.j[i].block0) k = i;
    const ColsumJob& jb = tab.j[k];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int n = ((int)blockIdx.x - jb.block0) * 32 + c;
    float s = 0.f;
    if (n < jb.N) {
        const float* x = jb.X + n;
        int kk = g;
        for (; kk + 56 < jb.K; kk += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(size_t)(kk + 8 * u) * jb.ldx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; kk < jb.K; kk += 8) s += x[(size_t)kk * jb.ldx];
    }
    part[g][c] = s;
    __syncthreads();
    if (g == 0 && n < jb.N) {
        float t = part[0][c];
#pragma unroll
        for (int q = 1; q < 8; ++q) t += part[q][c];
        jb.out[n] = t;
        if (jb.out2) jb.out2[n] = t;
    }
}

// out[b, n] = sum_t X[t, b, n]   (time sum of the TD gate gradients for the hoisted mean-feature weights)
__global__ __launch_bounds__(256) void timesum_kernel(const float* __restrict__ X, int T, size_t BN, float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= BN) return;
    f32x4 s = *reinterpret_cast<const f32x4*>(X + i);
    for (int t = 1; t < T; ++t) s += *reinterpret_cast<const f32x4*>(X + (size_t)t * BN + i);
    *reinterpret_cast<f32x4*>(out + i) = s;
}

// ---------------------------------------------------------------------------------------------------------
// Embedding gradient (Embedding -> ReLU -> Dropout backward):
//   dE[v,:] = sum over (t,b) with tok[t,b] == v of demb[t,b,:] * (emb[t,b,:] > 0 ? scale : 0)
// No atomics: a workgroup takes 8 vocabulary rows.  The (t,b) token list is loaded into LDS once per workgroup and each wave
// finds the occurrences of two of the rows by ballot, in list order.  Then wave w adds the rows' occurrences for the columns
// 256 w .. 256 w + 255 (+ 1024 k), in list order, eight occurrences' loads in flight: the kernel lasts as long as its most
// frequent token (<sta> once per caption; in XE batches the most frequent word), so its chain of dependent loads is what to
// keep short.  Rows without occurrences (almost all of them) are written as zeros.
constexpr int EG_ROWS = 8;
__global__ __launch_bounds__(256) void embed_grad_kernel(const int64_t* __restrict__ tok, int n_tok,
                                                         const float* __restrict__ demb, int ns, size_t slab_stride,
                                                         const float* __restrict__ emb, float scale, int E,
                                                         float* __restrict__ dE, int V, int relu, const int* __restrict__ rows_live) {
    extern __shared__ int eg_sm[];     // tokens [n_tok], then the occurrence lists of the eight rows [8][n_tok]
    __shared__ int snh[EG_ROWS];
    int* stok = eg_sm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rows_live) n_tok = min(n_tok, *rows_live);       // the (t, b) prefix of the steps a sampled rollout really ran (d emb rows behind it are stale)
    for (int i = tid; i < n_tok; i += 256) stok[i] = (int)tok[i];
    __syncthreads();
    for (int rr = 0; rr < EG_ROWS / 4; ++rr) {
        const int row8 = wave + 4 * rr, v = blockIdx.x * EG_ROWS + row8;
        int* hits = eg_sm + n_tok + row8 * n_tok;
        int nh = 0;
        for (int i0 = 0; i0 < n_tok; i0 += 64) {
            const int i = i0 + lane;
            const bool m = i < n_tok && stok[i] == v;
            const unsigned long long bal = __ballot(m);
            if (m) hits[nh + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            nh += __popcll(bal);
        }
        if (lane == 0) snh[row8] = nh;
    }
    __syncthreads();
    for (int row8 = 0; row8 < EG_ROWS; ++row8) {
        const int v = blockIdx.x * EG_ROWS + row8;
        if (v >= V) break;
        const int nh = snh[row8];
        const int* hits = eg_sm + n_tok + row8 * n_tok;
        for (int e = 256 * wave + lane * 4; e < E; e += 1024) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            for (int h0 = 0; h0 < nh; h0 += 8) {
                f32x4 g[8], x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const size_t off = (size_t)hits[min(h0 + u, nh - 1)] * E + e;
                    g[u] = sum_slabs4(demb, ns, slab_stride, off);
                    if (relu) x[u] = *reinterpret_cast<const f32x4*>(emb + off);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (h0 + u < nh) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) s[j] += (!relu || x[u][j] > 0.f) ? g[u][j] * scale : 0.f;
                    }
            }
            *reinterpret_cast<f32x4*>(dE + (size_t)v * E + e) = s;
        }
    }
}
// host side: grid, LDS size (above the 64 KB default the kernel needs the explicit opt-in)
inline hipError_t embed_grad_launch(hipStream_t st, const int64_t* tok, int n_tok, const float* demb, int ns, size_t slab_stride,
                                    const float* emb, float scale, int E, float* dE, int V, int relu, const int* rows_live = nullptr) {
    const size_t lds = sizeof(int) * (1 + EG_ROWS) * (size_t)n_tok;
    if (lds > 160 * 1024 - 1024) return hipErrorInvalidValue;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(embed_grad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(embed_grad_kernel, dim3((V + EG_ROWS - 1) / EG_ROWS), dim3(256), lds, st, tok, n_tok, demb, ns, slab_stride, emb, scale, E, dE,
                       V, relu, rows_live);
    return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------
// weight_norm backward (one wave per row):  w = g v / ||v||
//   dg[r] = dw[r,:] . v[r,:] / ||v||;   dv[r,:] = (g/||v||) (dw[r,:] - (dg/||v||) v[r,:])
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dw, int lddw, const float* __restrict__ v,
                                                              const float* __restrict__ g, const float* __restrict__ norm,
                                                              float* __restrict__ dv, float* __restrict__ dg, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* vr = v + (size_t)row * cols;
    const float* dr = dw + (size_t)row * lddw;
    float dot = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
        dot += x[0] * d[0] + x[1] * d[1] + x[2] * d[2] + x[3] * d[3];
    }
    dot = wave_sum(dot);
    const float nrm = norm[row];
    const float dgv = dot / nrm;
    const float s = g[row] / nrm;
    float* o = dv + (size_t)row * cols;
    for (int c = lane * 4; c < cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
        *reinterpret_cast<f32x4*>(o + c) = (d - x * (dgv / nrm)) * s;
    }
    if (lane == 0) dg[row] = dgv;
}

// The attention block's three weight-normed layers in ONE launch (blocks job after job, 4 rows per block).
struct WeightNormBwdJob { const float* dw; int lddw; const float* v; const float* g; const float* norm; float* dv; float* dg; int rows, cols, block0; };
struct WeightNormBwdTable { WeightNormBwdJob j[4]; int count; };
__global__ __launch_bounds__(256) void weight_norm_bwd_multi_kernel(WeightNormBwdTable tab) {
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < tab.count; ++i)
        if ((int)blockIdx.x >= tab.j[i].block0) k = i;
    const WeightNormBwdJob& jb = tab.j[k];
    const int lane = threadIdx.x & 63;
    const int row = ((int)blockIdx.x - jb.block0) * 4 + (threadIdx.x >> 6);
    if (row >= jb.rows) return;
    const float* vr = jb.v + (size_t)row * jb.cols;
    const float* dr = jb.dw + (size_t)row * jb.lddw;
    float dot = 0.f;
    for (int c = lane * 4; c < jb.cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
        dot += x[0] * d[0] + x[1] * d[1] + x[2] * d[2] + x[3] * d[3];
    }
    dot = wave_sum(dot);
    const float nrm = jb.norm[row];
    const float dgv = dot / nrm;
    const float s = jb.g[row] / nrm;
    float* o = jb.dv + (size_t)row * jb.cols;
    for (int c = lane * 4; c < jb.cols; c += 256) {
        f32x4 x = *reinterpret_cast<const f32x4*>(vr + c);
        f32x4 d = *reinterpret_cast<const f32x4*>(dr + c);
        *reinterpret_cast<f32x4*>(o + c) = (d - x * (dgv / nrm)) * s;
    }
    if (lane == 0) jb.dg[row] = dgv;
}

// ---------------------------------------------------------------------------------------------------------
// clip_gradient + Adam over a table of tensors in ONE launch (21 parameter tensors per step otherwise cost 21 launches).
struct AdamTensor { float* p; const float* g; float* m; float* v; size_t n; size_t block0; };
constexpr int ADAM_MAX_TENSORS = 32;
struct AdamTable { AdamTensor t[ADAM_MAX_TENSORS]; int count; };
__global__ __launch_bounds__(256) void adam_clamp_multi_kernel(AdamTable tab, float lr, float clip, float bc1, float sqrt_bc2) {
    // find the tensor of this block (blocks are laid out tensor after tensor)
    int k = 0;
#pragma unroll 1
    for (int i = 1; i < tab.count; ++i)
        if (blockIdx.x >= tab.t[i].block0) k = i;
    const AdamTensor T = tab.t[k];
    const size_t i = ((size_t)blockIdx.x - T.block0) * 1024 + threadIdx.x * 4;
    if (i >= T.n) return;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    if (i + 4 <= T.n && (((uintptr_t)(T.p + i) | (uintptr_t)(T.g + i) | (uintptr_t)(T.m + i) | (uintptr_t)(T.v + i)) & 15) == 0) {
        f32x4 g = *reinterpret_cast<const f32x4*>(T.g + i), m = *reinterpret_cast<f32x4*>(T.m + i);
        f32x4 v = *reinterpret_cast<f32x4*>(T.v + i), p = *reinterpret_cast<f32x4*>(T.p + i);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gv = fminf(fmaxf(g[e], -clip), clip);
            m[e] = m[e] * b1 + gv * (1.f - b1);
            v[e] = v[e] * b2 + (gv * gv) * (1.f - b2);
            p[e] = p[e] - (lr / bc1) * (m[e] / (sqrtf(v[e]) / sqrt_bc2 + eps));
        }
        *reinterpret_cast<f32x4*>(T.m + i) = m;
        *reinterpret_cast<f32x4*>(T.v + i) = v;
        *reinterpret_cast<f32x4*>(T.p + i) = p;
    } else {
        for (size_t e = i; e < T.n && e < i + 4; ++e) {
            const float gv = fminf(fmaxf(T.g[e], -clip), clip);
            const float mn = T.m[e] * b1 + gv * (1.f - b1);
            const float vn = T.v[e] * b2 + (gv * gv) * (1.f - b2);
            T.m[e] = mn; T.v[e] = vn;
            T.p[e] = T.p[e] - (lr / bc1) * (mn / (sqrtf(vn) / sqrt_bc2 + eps));
        }
    }
}

// clip_gradient (Utils.py:241-250) + Adam (Utils.py:219-220) fused, elementwise.
__global__ __launch_bounds__(256) void adam_clamp_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, size_t n, float lr, float clip, float bc1,
                                                         float sqrt_bc2) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    float gv = fminf(fmaxf(g[i], -clip), clip);
    const float mn = m[i] * b1 + gv * (1.f - b1);
    const float vn = v[i] * b2 + (gv * gv) * (1.f - b2);
    m[i] = mn;
    v[i] = vn;
    const float denom = sqrtf(vn) / sqrt_bc2 + eps;
    p[i] = p[i] - (lr / bc1) * (mn / denom);
}

}  // namespace
}  // namespace icz
