// Device kernels of the BUTD decoder step (forward).  All HBM-bound byte movers: coalesced 16-byte accesses,
// wave-shuffle reductions, no atomics (bitwise reproducible).  Reference line numbers: Models/BUTD_Model.py.
#pragma once
#include "icz_common.h"
#include "rng.h"

namespace icz {

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
                                                    float* __restrict__ emb, int rows, int E, DropCfg dc) {
    const int row = blockIdx.y;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    f32x4 x = *reinterpret_cast<const f32x4*>(table + (size_t)it[row] * E + e);
    uint32_t k = dc.mode ? dc.keep4((uint64_t)row * E + e) : 0xFu;
    const float sc = dc.mode ? 2.0f : 1.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = ((k >> j) & 1u) ? fmaxf(x[j], 0.f) * sc : 0.f;
    *reinterpret_cast<f32x4*>(emb + (size_t)row * E + e) = x;
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
};
__global__ __launch_bounds__(256) void lstm_point_kernel(LstmPointArgs a, DropCfg dc) {
    const int row = blockIdx.y;
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= a.H) return;
    const int H = a.H, G = 4 * H;
    const size_t MN = (size_t)a.rows * G;
    f32x4 gt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const size_t off = (size_t)row * G + q * H + j;
        f32x4 s = *reinterpret_cast<const f32x4*>(a.slab + off);
        for (int z = 1; z < a.nsplit; ++z) s += *reinterpret_cast<const f32x4*>(a.slab + (size_t)z * MN + off);
        if (a.pre) {
            const int pr = a.pre_row ? a.pre_row[row] : row;
            s += *reinterpret_cast<const f32x4*>(a.pre + (size_t)pr * G + q * H + j);
        }
        s += *reinterpret_cast<const f32x4*>(a.b_ih + q * H + j);
        s += *reinterpret_cast<const f32x4*>(a.b_hh + q * H + j);
        gt[q] = s;
    }
    f32x4 cp = *reinterpret_cast<const f32x4*>(a.c_prev + (size_t)row * H + j);
    f32x4 hn, cn, gi, gf, gg, go;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gi[e] = sigmoidf_(gt[0][e]);
        gf[e] = sigmoidf_(gt[1][e]);
        gg[e] = tanhf(gt[2][e]);
        go[e] = sigmoidf_(gt[3][e]);
        cn[e] = gf[e] * cp[e] + gi[e] * gg[e];
        hn[e] = go[e] * tanhf(cn[e]);
    }
    *reinterpret_cast<f32x4*>(a.h_out + (size_t)row * H + j) = hn;
    *reinterpret_cast<f32x4*>(a.c_out + (size_t)row * H + j) = cn;
    if (a.gates_out) {
        float* go_ = a.gates_out + (size_t)row * G + j;
        *reinterpret_cast<f32x4*>(go_) = gi;
        *reinterpret_cast<f32x4*>(go_ + H) = gf;
        *reinterpret_cast<f32x4*>(go_ + 2 * H) = gg;
        *reinterpret_cast<f32x4*>(go_ + 3 * H) = go;
    }
    if (a.hdrop_out) {
        f32x4 hd = hn;
        if (dc.mode) {
            uint32_t k = dc.keep4((uint64_t)row * H + j);
#pragma unroll
            for (int e = 0; e < 4; ++e) hd[e] = ((k >> e) & 1u) ? hn[e] * 2.0f : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.hdrop_out + (size_t)row * H + j) = hd;
    }
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
};
__global__ __launch_bounds__(256) void att_scores_kernel(AttScoreArgs a, DropCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float sdec[];   // A floats
    const int row = blockIdx.x, part = blockIdx.y, nparts = gridDim.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t MN = (size_t)a.rows * a.A;
    for (int c = tid * 4; c < a.A; c += 1024) {
        const size_t off = (size_t)row * a.A + c;
        f32x4 s = *reinterpret_cast<const f32x4*>(a.dec_slab + off);
        for (int z = 1; z < a.nsplit; ++z) s += *reinterpret_cast<const f32x4*>(a.dec_slab + (size_t)z * MN + off);
        s += *reinterpret_cast<const f32x4*>(a.b_dec + c);
        *reinterpret_cast<f32x4*>(sdec + c) = s;
        if (a.dec_ctx_out && part == 0) *reinterpret_cast<f32x4*>(a.dec_ctx_out + off) = s;
    }
    __syncthreads();
    const int img = a.img_of_row ? a.img_of_row[row] : row;
    const int per = (a.R + nparts - 1) / nparts;
    const int r_end = min(a.R, (part + 1) * per);
    const float baff = a.b_aff[0];
    for (int r = part * per + wave; r < r_end; r += 4) {
        const float* e = a.enc_ctx + ((size_t)img * a.R + r) * a.A;
        float acc = 0.f;
        for (int c = lane * 4; c < a.A; c += 256) {
            f32x4 x = *reinterpret_cast<const f32x4*>(e + c);
            f32x4 d = *reinterpret_cast<const f32x4*>(sdec + c);
            f32x4 w = *reinterpret_cast<const f32x4*>(a.w_aff + c);
            uint32_t k = dc.mode ? dc.keep4(((uint64_t)row * a.R + r) * a.A + c) : 0xFu;
            const float sc = dc.mode ? 2.0f : 1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float zv = fmaxf(x[j] + d[j], 0.f);
                zv = ((k >> j) & 1u) ? zv * sc : 0.f;
                acc += zv * w[j];
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) a.scores[(size_t)row * a.R + r] = acc + baff;
    }
}

// ---------------------------------------------------------------------------------------------------------
// softmax over regions + attention-weighted feature sum (:60-61):
//   alpha = softmax_r(score[row,:]);  ctx[row, d] = sum_r alpha[r] * feats[img, r, d]
// Grid (rows, D/1024): each thread owns 4 consecutive d; a wave's loads are 1 KiB contiguous per region.
// The R <= 64 scores are reduced with wave shuffles (lane r holds score r) by every wave (redundantly, R is tiny).
__global__ __launch_bounds__(256) void att_ctx_kernel(const float* __restrict__ feats, const int32_t* __restrict__ img_of_row,
                                                      const float* __restrict__ scores, float* __restrict__ alpha_out,
                                                      float* __restrict__ alpha_out2, int alpha2_stride,
                                                      float* __restrict__ ctx, int R, int D) {
    const int row = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const float sc = lane < R ? scores[(size_t)row * R + lane] : -INFINITY;
    const float mx = wave_max(sc);
    const float ex = lane < R ? expf(sc - mx) : 0.f;
    const float sum = wave_sum(ex);
    const float al = ex / sum;
    if (blockIdx.y == 0 && threadIdx.x < R) {
        alpha_out[(size_t)row * R + threadIdx.x] = al;
        if (alpha_out2) alpha_out2[(size_t)row * alpha2_stride + threadIdx.x] = al;
    }
    const int d = (blockIdx.y * 256 + threadIdx.x) * 4;
    const bool valid = d < D;            // no early return: every lane must stay active for the shuffles
    const int img = img_of_row ? img_of_row[row] : row;
    const float* f = feats + (size_t)img * R * D + (valid ? d : 0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r) {
        const float w = __shfl(al, r, 64);
        if (valid) acc += *reinterpret_cast<const f32x4*>(f + (size_t)r * D) * w;
    }
    if (valid) *reinterpret_cast<f32x4*>(ctx + (size_t)row * D + d) = acc;
}

// ---------------------------------------------------------------------------------------------------------
// greedy epilogue (:183): id = argmax_v logits[row, v] (first maximum wins, as torch.max does)
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int V, int64_t* __restrict__ it_next,
                                                     int64_t* __restrict__ ids_out, int ids_stride, int t) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* l = logits + (size_t)row * V;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        float x = l[v];
        if (x > best) { best = x; bi = v; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        it_next[row] = bi;
        if (ids_out) ids_out[(size_t)row * ids_stride + t] = bi;
    }
}

__global__ void fill_i64_kernel(int64_t* p, int64_t v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

}  // namespace icz
