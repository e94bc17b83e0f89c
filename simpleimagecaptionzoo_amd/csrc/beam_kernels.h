// Per-image beam expand / prune kernels shared by the decoders (BUTD: butd_beam.hip, NIC: nic.hip).
#pragma once
#include "butd_kernels.h"

namespace icz {
namespace {

constexpr int BEAM_MAX_K = 8;

struct BeamArgs {
    const float* logits; int V; int ldl; int k; int step; int L;   // L = max_steps + 1 (sequence capacity)
    int* n_act;             // [n_img]   active beams (the reference's shrinking k)
    float* run;             // [n_img,k] running scores of the active beams
    const int32_t* seqs_in; int32_t* seqs_out;     // [n_img,k,L]
    int32_t* src_row;       // [n_img*k] decoder row whose state feeds this row next step
    int64_t* it_next;       // [n_img*k]
    float* best_score; int* best_len; int32_t* best_seq; int* has_complete;   // best finished hypothesis per image
    int* n_live;            // [1] number of images that still have active beams after this step
};

// Beam expand / prune of one step (:271-300) in two launches:
//   beam_rowtopk_kernel  grid (n_img * k): one workgroup per decoder row -> its log-softmax normaliser and its own best
//                        n_act candidates (value = run + log_softmax, index v); every row of every image in parallel
//   beam_merge_kernel    grid (n_img), one wave: top-n_act of the <= k*k row candidates (ties -> lower flat index r*V+v,
//                        the order of a top-k over the flattened [k, V] scores), retire finished beams, compact the rest
// The global top-n_act are contained in the union of the per-row top-n_act, so the result equals a search over all k*V.
__global__ __launch_bounds__(256) void beam_rowtopk_kernel(const float* __restrict__ logits, int V, int ldl, int k, int step,
                                                           const int* __restrict__ n_act, const float* __restrict__ run,
                                                           float* __restrict__ cand_val, int* __restrict__ cand_idx, int compact) {
    __shared__ float smf[4];
    __shared__ float s_val[4];
    __shared__ int s_idx[4], s_who[4];
    const int row = blockIdx.x, img = row / k, r = row % k, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = n_act[img];
    const int nr = (step == 1) ? 1 : na;          // step 1 scores row 0 only (:273-274)
    if (r >= nr) return;
    const float* l = logits + (size_t)(compact ? img : row) * ldl;      // compact (step 1 only): one decoder row per image
    // Every sweep fetches the thread's strided slice (40 logits at V = 10102) in batches of U independent loads: one memory
    // latency per batch instead of one per element.  (All 40 in registers across the three sweeps: the unrolled kernel
    // outgrows the instruction cache; the row staged in LDS for the second and third sweep: no faster, 36 -> 37 us.)
    constexpr int U = 8;
    auto slice = [&](int v0, float (&x)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) { const int v = v0 + 256 * u; x[u] = v < V ? l[v] : -INFINITY; }
    };
    float mx = -INFINITY;
    for (int v0 = tid; v0 < V; v0 += 256 * U) {
        float x[U];
        slice(v0, x);
#pragma unroll
        for (int u = 0; u < U; ++u) mx = fmaxf(mx, x[u]);
    }
    mx = block_max_256(mx, smf);
    float se = 0.f;
    for (int v0 = tid; v0 < V; v0 += 256 * U) {
        float x[U];
        slice(v0, x);
#pragma unroll
        for (int u = 0; u < U; ++u) if (v0 + 256 * u < V) se += expf(x[u] - mx);
    }
    se = block_sum_256(se, smf);
    const float ls = logf(se);
    const float rs = (step == 1) ? 0.f : run[row];
    // thread-local best `na` of its strided slice (sorted: value descending, index ascending on ties)
    float tv[BEAM_MAX_K];
    int ti[BEAM_MAX_K];
#pragma unroll
    for (int j = 0; j < BEAM_MAX_K; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    // The list is sorted, so an element enters it only if it beats the last kept entry (position na - 1): nearly all of a
    // thread's ~40 elements fail that one test and skip the insertion chain.
    float worst = -INFINITY;
    int worst_i = 0x7fffffff;
    for (int v0 = tid; v0 < V; v0 += 256 * U) {
        float x[U];
        slice(v0, x);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (v0 + 256 * u >= V) continue;
            float val = rs + ((x[u] - mx) - ls);
            int idx = v0 + 256 * u;
            if (!(val > worst || (val == worst && idx < worst_i))) continue;
#pragma unroll
            for (int j = 0; j < BEAM_MAX_K; ++j) {          // branch-free insertion: swap down the list
                const bool take = (j < na) & ((val > tv[j]) | ((val == tv[j]) & (idx < ti[j])));
                const float ov = tv[j]; const int oi = ti[j];
                tv[j] = take ? val : ov; ti[j] = take ? idx : oi;
                val = take ? ov : val; idx = take ? oi : idx;
            }
#pragma unroll
            for (int j = 0; j < BEAM_MAX_K; ++j) {
                worst = (j == na - 1) ? tv[j] : worst;
                worst_i = (j == na - 1) ? ti[j] : worst_i;
            }
        }
    }
    // na rounds: every thread offers the head of its list, the block takes the best and that thread pops it
    int head = 0;
    for (int j = 0; j < na; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < BEAM_MAX_K; ++q)
            if (q == head) { best = tv[q]; bi = ti[q]; }
        int who = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64), ow = __shfl_xor(who, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; who = ow; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; s_who[wave] = who; }
        __syncthreads();
        best = s_val[0]; bi = s_idx[0]; who = s_who[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; who = s_who[w]; }
        if (tid == who) ++head;
        if (tid == 0) { cand_val[row * BEAM_MAX_K + j] = best; cand_idx[row * BEAM_MAX_K + j] = bi; }
        __syncthreads();
    }
}

// The same result from a row held in registers (16-byte loads, one pass over memory) and a threshold instead of per-thread
// sorted lists.  The sorted insertion above is VALU-bound: a wave skips an element only when none of its 64 lanes inserts it,
// so nearly all of the 40 x 8-deep insertion chains execute (36 us per step at 640 rows).  Here every thread takes the maximum
// of its 4 NV4 candidate scores; tau = the n-th largest of the 256 thread maxima (n = active beams; equal maxima of different
// threads count separately).  At least n scores are >= tau, so every member of the row's top n is: the threads append their
// scores >= tau to an LDS list (a handful unless the row is full of ties) and one wave picks the top n of the list by
// (score descending, index ascending) -- the order of a top-k over the flattened scores.  A list that overflows (massive ties,
// e.g. constant logits) sends the workgroup through the insertion algorithm on the LDS-staged scores instead.
// Needs V <= 1024 NV4, 16-byte aligned rows (ldl % 4 == 0).  The log-sum-exp is summed in a different element order than in
// the kernel above (thread t holds elements 4 (t + 256 u) + j): scores may differ in the last bit.
constexpr int BEAM_CAND_CAP = 128;
template <int NV4>
__global__ __launch_bounds__(256) void beam_rowtopk_reg_kernel(const float* __restrict__ logits, int V, int ldl, int k, int step,
                                                               const int* __restrict__ n_act, const float* __restrict__ run,
                                                               float* __restrict__ cand_val, int* __restrict__ cand_idx, int compact) {
    __shared__ float smf[4];
    __shared__ float s_val[4];
    __shared__ int s_idx[4], s_who[4];
    __shared__ int s_cnt, s_taken;
    __shared__ float s_cv[BEAM_CAND_CAP];
    __shared__ int s_ci[BEAM_CAND_CAP];
    extern __shared__ __attribute__((aligned(16))) float s_row[];          // overflow path only: 1024 NV4 floats
    const int row = blockIdx.x, img = row / k, r = row % k, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int na = n_act[img];
    const int nr = (step == 1) ? 1 : na;          // step 1 scores row 0 only (:273-274)
    if (r >= nr) return;
    const float* l = logits + (size_t)(compact ? img : row) * ldl;      // compact (step 1 only): one decoder row per image
    f32x4 x[NV4];
#pragma unroll
    for (int u = 0; u < NV4; ++u) {
        const int v = 4 * (tid + 256 * u);
        x[u] = v < V ? *reinterpret_cast<const f32x4*>(l + v) : (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int j = 0; j < 4; ++j) if (v + j >= V) x[u][j] = -INFINITY;
    }
    if (tid == 0) { s_cnt = 0; s_taken = 0; }
    float mx = -INFINITY;
#pragma unroll
    for (int u = 0; u < NV4; ++u) mx = fmaxf(mx, fmaxf(fmaxf(x[u][0], x[u][1]), fmaxf(x[u][2], x[u][3])));
    mx = block_max_256(mx, smf);
    float se = 0.f;
#pragma unroll
    for (int u = 0; u < NV4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) se += expf(x[u][j] - mx);          // exp(-inf) = 0 for the slots beyond V
    se = block_sum_256(se, smf);
    const float ls = logf(se);
    const float rs = (step == 1) ? 0.f : run[row];
    float tmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < NV4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            x[u][j] = rs + ((x[u][j] - mx) - ls);                      // the candidate score (beyond V: -inf)
            tmax = fmaxf(tmax, x[u][j]);
        }
    // tau: rounds of block maximum over the thread maxima not yet counted, until na of them are
    float tau = -INFINITY;
    {
        float cur = tmax;
        for (int round = 0; round < BEAM_MAX_K; ++round) {
            const float m = block_max_256(cur, smf);
            const bool hit = cur == m && m > -INFINITY;
            const unsigned long long b = __ballot(hit);
            if (lane == 0 && b) atomicAdd(&s_taken, __popcll(b));
            if (hit) cur = -INFINITY;
            __syncthreads();
            tau = m;
            const int taken = s_taken;
            __syncthreads();
            if (taken >= na || m == -INFINITY) break;
        }
    }
#pragma unroll
    for (int u = 0; u < NV4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x[u][j] >= tau && x[u][j] > -INFINITY) {
                const int pos = atomicAdd(&s_cnt, 1);
                if (pos < BEAM_CAND_CAP) { s_cv[pos] = x[u][j]; s_ci[pos] = 4 * (tid + 256 * u) + j; }
            }
    __syncthreads();
    const int cnt = s_cnt;
    if (cnt <= BEAM_CAND_CAP) {
        if (wave != 0) return;
        float v0 = lane < cnt ? s_cv[lane] : -INFINITY, v1 = lane + 64 < cnt ? s_cv[lane + 64] : -INFINITY;
        int i0 = lane < cnt ? s_ci[lane] : 0x7fffffff, i1 = lane + 64 < cnt ? s_ci[lane + 64] : 0x7fffffff;
        for (int j = 0; j < na; ++j) {
            float best = v0;
            int bi = i0;
            if (v1 > best || (v1 == best && i1 < bi)) { best = v1; bi = i1; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (i0 == bi) { v0 = -INFINITY; i0 = 0x7fffffff; }      // taken (indices are unique)
            if (i1 == bi) { v1 = -INFINITY; i1 = 0x7fffffff; }
            if (lane == 0) { cand_val[row * BEAM_MAX_K + j] = best; cand_idx[row * BEAM_MAX_K + j] = bi; }
        }
        return;
    }
    // ---- overflow: the insertion algorithm of beam_rowtopk_kernel on the scores, staged in LDS in index order
#pragma unroll
    for (int u = 0; u < NV4; ++u) *reinterpret_cast<f32x4*>(s_row + 4 * (tid + 256 * u)) = x[u];
    __syncthreads();
    float tv[BEAM_MAX_K];
    int ti[BEAM_MAX_K];
#pragma unroll
    for (int j = 0; j < BEAM_MAX_K; ++j) { tv[j] = -INFINITY; ti[j] = 0x7fffffff; }
    float worst = -INFINITY;
    int worst_i = 0x7fffffff;
    for (int v = tid; v < V; v += 256) {
        float val = s_row[v];
        int idx = v;
        if (!(val > worst || (val == worst && idx < worst_i))) continue;
#pragma unroll
        for (int j = 0; j < BEAM_MAX_K; ++j) {
            const bool take = (j < na) & ((val > tv[j]) | ((val == tv[j]) & (idx < ti[j])));
            const float ov = tv[j]; const int oi = ti[j];
            tv[j] = take ? val : ov; ti[j] = take ? idx : oi;
            val = take ? ov : val; idx = take ? oi : idx;
        }
#pragma unroll
        for (int j = 0; j < BEAM_MAX_K; ++j) {
            worst = (j == na - 1) ? tv[j] : worst;
            worst_i = (j == na - 1) ? ti[j] : worst_i;
        }
    }
    int head = 0;
    for (int j = 0; j < na; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < BEAM_MAX_K; ++q)
            if (q == head) { best = tv[q]; bi = ti[q]; }
        int who = tid;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64), ow = __shfl_xor(who, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; who = ow; }
        }
        if (lane == 0) { s_val[wave] = best; s_idx[wave] = bi; s_who[wave] = who; }
        __syncthreads();
        best = s_val[0]; bi = s_idx[0]; who = s_who[0];
#pragma unroll
        for (int w = 1; w < 4; ++w)
            if (s_val[w] > best || (s_val[w] == best && s_idx[w] < bi)) { best = s_val[w]; bi = s_idx[w]; who = s_who[w]; }
        if (tid == who) ++head;
        if (tid == 0) { cand_val[row * BEAM_MAX_K + j] = best; cand_idx[row * BEAM_MAX_K + j] = bi; }
        __syncthreads();
    }
}

// per-row candidates of one beam step: the register kernel where the vocabulary fits it, else the sweep kernel
inline void launch_beam_rowtopk(hipStream_t st, int rows, const float* logits, int V, int ldl, int k, int step, const int* n_act,
                                const float* run, float* cand_val, int* cand_idx, int compact = 0) {
    const bool ok = ldl % 4 == 0 && ((uintptr_t)logits & 15) == 0;
    if (ok && V <= 1024 * 3) hipLaunchKernelGGL((beam_rowtopk_reg_kernel<3>), dim3(rows), dim3(256), sizeof(float) * 1024 * 3, st, logits, V, ldl, k, step, n_act, run, cand_val, cand_idx, compact);
    else if (ok && V <= 1024 * 10) hipLaunchKernelGGL((beam_rowtopk_reg_kernel<10>), dim3(rows), dim3(256), sizeof(float) * 1024 * 10, st, logits, V, ldl, k, step, n_act, run, cand_val, cand_idx, compact);
    else hipLaunchKernelGGL(beam_rowtopk_kernel, dim3(rows), dim3(256), 0, st, logits, V, ldl, k, step, n_act, run, cand_val, cand_idx, compact);
}

__global__ __launch_bounds__(64) void beam_merge_kernel(BeamArgs a, const float* __restrict__ cand_val, const int* __restrict__ cand_idx) {
    __shared__ float pick_val[BEAM_MAX_K];
    __shared__ int pick_idx[BEAM_MAX_K];
    __shared__ int new_src[BEAM_MAX_K], new_tok[BEAM_MAX_K], s_newn;
    __shared__ float new_run[BEAM_MAX_K];
    const int img = blockIdx.x, lane = threadIdx.x;
    const int k = a.k, V = a.V;
    const int na = a.n_act[img];
    const int row0 = img * k;
    if (na == 0) {
        for (int j = lane; j < k; j += 64) { a.src_row[row0 + j] = row0 + j; a.it_next[row0 + j] = 0; }
        return;
    }
    const int nr = (a.step == 1) ? 1 : na;
    // lane c <-> candidate (row r = c / BEAM_MAX_K, rank j = c % BEAM_MAX_K)
    const int cr = lane / BEAM_MAX_K, cj = lane % BEAM_MAX_K;
    float val = -INFINITY;
    int idx = 0x7fffffff;
    if (cr < nr && cj < na) {
        val = cand_val[(row0 + cr) * BEAM_MAX_K + cj];
        idx = cr * V + cand_idx[(row0 + cr) * BEAM_MAX_K + cj];
    }
    for (int j = 0; j < na; ++j) {
        float best = val;
        int bi = idx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (idx == bi) { val = -INFINITY; idx = 0x7fffffff; }      // taken (flat indices are unique)
        if (lane == 0) { pick_val[j] = best; pick_idx[j] = bi; }
    }
    __syncthreads();
    // retire finished beams, compact the rest (:279-300)
    if (lane == 0) {
        int nn = 0;
        for (int j = 0; j < na; ++j) {
            const int src = pick_idx[j] / V, tok = pick_idx[j] % V;
            if (tok == 2) {
                if (!a.has_complete[img] || pick_val[j] > a.best_score[img]) {
                    a.has_complete[img] = 1;
                    a.best_score[img] = pick_val[j];
                    a.best_len[img] = a.step + 1;
                    int32_t* bs = a.best_seq + (size_t)img * a.L;
                    const int32_t* ss = a.seqs_in + (size_t)(row0 + src) * a.L;
                    for (int i = 0; i < a.step; ++i) bs[i] = ss[i];
                    bs[a.step] = 2;
                }
            } else {
                new_src[nn] = src; new_tok[nn] = tok; new_run[nn] = pick_val[j];
                ++nn;
            }
        }
        s_newn = nn;
        a.n_act[img] = nn;
        if (nn > 0) atomicAdd(a.n_live, 1);
    }
    __syncthreads();
    const int nn = s_newn;
    for (int j = 0; j < k; ++j) {
        if (j < nn) {
            const int32_t* ss = a.seqs_in + (size_t)(row0 + new_src[j]) * a.L;
            int32_t* so = a.seqs_out + (size_t)(row0 + j) * a.L;
            for (int i = lane; i < a.step; i += 64) so[i] = ss[i];
            if (lane == 0) {
                so[a.step] = new_tok[j];
                a.run[row0 + j] = new_run[j];
                a.src_row[row0 + j] = row0 + new_src[j];
                a.it_next[row0 + j] = new_tok[j];
            }
        } else if (lane == 0) {
            a.src_row[row0 + j] = row0 + j;
            a.it_next[row0 + j] = 0;
        }
    }
}

// state re-gather: out[row,:] = in[src_row[row],:] for the four state tensors
__global__ __launch_bounds__(256) void beam_gather_kernel(const int32_t* __restrict__ src_row, int H,
                                                          const float* __restrict__ a0, const float* __restrict__ a1,
                                                          const float* __restrict__ a2, const float* __restrict__ a3,
                                                          float* __restrict__ o0, float* __restrict__ o1,
                                                          float* __restrict__ o2, float* __restrict__ o3, int src_div) {
    const int row = blockIdx.y;
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= H) return;
    // src_div = k after a compact first step (the state of image img sits in row img, src_row says img k + 0), else 1
    const size_t s = (size_t)(src_row[row] / src_div) * H + j, d = (size_t)row * H + j;
    *reinterpret_cast<f32x4*>(o0 + d) = *reinterpret_cast<const f32x4*>(a0 + s);
    *reinterpret_cast<f32x4*>(o1 + d) = *reinterpret_cast<const f32x4*>(a1 + s);
    *reinterpret_cast<f32x4*>(o2 + d) = *reinterpret_cast<const f32x4*>(a2 + s);
    *reinterpret_cast<f32x4*>(o3 + d) = *reinterpret_cast<const f32x4*>(a3 + s);
}

// out[row,:] = in[img_of_row[row],:]   (features.expand(k, ...) of the reference's beam search)
__global__ __launch_bounds__(256) void beam_expand_rows_kernel(const float* __restrict__ in, const int32_t* __restrict__ img_of_row, int E,
                                                               float* __restrict__ out) {
    const int row = blockIdx.y;
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= E) return;
    *reinterpret_cast<f32x4*>(out + (size_t)row * E + e) = *reinterpret_cast<const f32x4*>(in + (size_t)img_of_row[row] * E + e);
}

// final selection (:302-313): best finished hypothesis if any, else the best-scoring live beam
__global__ void beam_finalize_kernel(int k, int L, int steps_done, const int* __restrict__ n_act, const float* __restrict__ run,
                                     const int32_t* __restrict__ seqs, const int* __restrict__ has_complete,
                                     const int* __restrict__ best_len, const int32_t* __restrict__ best_seq,
                                     float* __restrict__ out, int32_t* __restrict__ lens) {
    const int img = blockIdx.x;
    const int32_t* src;
    int len;
    if (has_complete[img]) {
        src = best_seq + (size_t)img * L;
        len = best_len[img];
    } else {
        int bi = 0;
        float bv = -INFINITY;
        for (int j = 0; j < n_act[img]; ++j)
            if (run[img * k + j] > bv) { bv = run[img * k + j]; bi = j; }
        src = seqs + (size_t)(img * k + bi) * L;
        len = steps_done + 1;
    }
    for (int i = threadIdx.x; i < L; i += blockDim.x) out[(size_t)img * L + i] = i < len ? (float)src[i] : 0.f;
    if (threadIdx.x == 0) lens[img] = len;
}

__global__ void beam_init_kernel(int n_img, int k, int L, int* n_act, int32_t* seqs, int32_t* img_of_row, int64_t* it,
                                 int* has_complete, float* best_score) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_img * k) return;
    seqs[(size_t)row * L] = 1;      // <sta>
    img_of_row[row] = row / k;
    it[row] = 1;
    if (row % k == 0) {
        const int img = row / k;
        n_act[img] = k;
        has_complete[img] = 0;
        best_score[img] = -INFINITY;
    }
}

}  // namespace
}  // namespace icz
