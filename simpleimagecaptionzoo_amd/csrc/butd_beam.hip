// Batched beam search (DecoderRNN.beam_search_sample, Models/BUTD_Model.py:236-318) for many images at once.
// The reference decodes one image per call with a Python list comprehension over tensor elements every step
// (k host syncs per step, :282-283).  Here every image owns k consecutive decoder rows; after the shared decoder
// step a per-image workgroup does log-softmax + running score + top-k over (active beams x V), retires beams that
// emitted <end> (k shrinks exactly as in the reference, no length normalisation), and emits the row permutation
// that re-gathers the LSTM state.  No host synchronisation inside a step.
#include "butd_impl.h"

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

__global__ __launch_bounds__(256) void beam_step_kernel(BeamArgs a) {
    __shared__ float smf[4];
    __shared__ int smi[4];
    __shared__ float s_mx[BEAM_MAX_K], s_lse[BEAM_MAX_K];
    __shared__ float pick_val[BEAM_MAX_K];
    __shared__ int pick_idx[BEAM_MAX_K];
    __shared__ int new_src[BEAM_MAX_K], new_tok[BEAM_MAX_K], s_newn;
    __shared__ float new_run[BEAM_MAX_K];
    const int img = blockIdx.x, tid = threadIdx.x;
    const int k = a.k, V = a.V;
    const int na = a.n_act[img];
    const int row0 = img * k;
    if (na == 0) {
        for (int j = tid; j < k; j += 256) { a.src_row[row0 + j] = row0 + j; a.it_next[row0 + j] = 0; }
        return;
    }
    const int nr = (a.step == 1) ? 1 : na;    // step 1 scores row 0 only (:273-274)
    for (int r = 0; r < nr; ++r) {
        const float* l = a.logits + (size_t)(row0 + r) * a.ldl;
        float mx = -INFINITY;
        for (int v = tid; v < V; v += 256) mx = fmaxf(mx, l[v]);
        mx = block_max_256(mx, smf);
        float se = 0.f;
        for (int v = tid; v < V; v += 256) se += expf(l[v] - mx);
        se = block_sum_256(se, smf);
        if (tid == 0) { s_mx[r] = mx; s_lse[r] = logf(se); }
    }
    __syncthreads();
    // top-na of run[r] + log_softmax(logits[r])[v] over (r, v); ties -> lower flat index
    for (int j = 0; j < na; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int r = 0; r < nr; ++r) {
            const float* l = a.logits + (size_t)(row0 + r) * a.ldl;
            const float rs = (a.step == 1) ? 0.f : a.run[row0 + r];
            const float mx = s_mx[r], ls = s_lse[r];
            for (int v = tid; v < V; v += 256) {
                const int idx = r * V + v;
                bool taken = false;
                for (int q = 0; q < j; ++q) taken |= (pick_idx[q] == idx);
                if (taken) continue;
                const float val = rs + ((l[v] - mx) - ls);
                if (val > best || (val == best && idx < bi)) { best = val; bi = idx; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o, 64);
            int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((tid & 63) == 0) { smf[tid >> 6] = best; smi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (smf[w] > best || (smf[w] == best && smi[w] < bi)) { best = smf[w]; bi = smi[w]; }
            pick_val[j] = best;
            pick_idx[j] = bi;
        }
        __syncthreads();
    }
    // retire finished beams, compact the rest (:279-300)
    if (tid == 0) {
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
            for (int i = tid; i < a.step; i += 256) so[i] = ss[i];
            if (tid == 0) {
                so[a.step] = new_tok[j];
                a.run[row0 + j] = new_run[j];
                a.src_row[row0 + j] = row0 + new_src[j];
                a.it_next[row0 + j] = new_tok[j];
            }
        } else if (tid == 0) {
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
                                                          float* __restrict__ o2, float* __restrict__ o3) {
    const int row = blockIdx.y;
    const int j = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (j >= H) return;
    const size_t s = (size_t)src_row[row] * H + j, d = (size_t)row * H + j;
    *reinterpret_cast<f32x4*>(o0 + d) = *reinterpret_cast<const f32x4*>(a0 + s);
    *reinterpret_cast<f32x4*>(o1 + d) = *reinterpret_cast<const f32x4*>(a1 + s);
    *reinterpret_cast<f32x4*>(o2 + d) = *reinterpret_cast<const f32x4*>(a2 + s);
    *reinterpret_cast<f32x4*>(o3 + d) = *reinterpret_cast<const f32x4*>(a3 + s);
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

int Butd::beam_search(const float* feats, int n_img, int k, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seqs_out && lens_out, "butd beam: null argument");
    ICZ_REQUIRE(k >= 1 && k <= BEAM_MAX_K, "butd beam: beam size %d out of range 1..%d", k, BEAM_MAX_K);
    ICZ_REQUIRE(n_img > 0 && (long)n_img * k <= dims.max_rows, "butd beam: %d images x %d beams exceed row capacity %d", n_img, k, dims.max_rows);
    ICZ_REQUIRE(max_steps >= 1 && max_steps <= 256, "butd beam: max_steps out of range");
    const int rows = n_img * k, L = max_steps + 1, H = dims.H;
    if (bm.cap_rows < rows || bm.cap_L < L) {
        const size_t R_ = dims.max_rows, L_ = L > 51 ? L : 51;
        ICZ_TRY(alloc((void**)&bm.n_act, sizeof(int) * R_));
        ICZ_TRY(alloc((void**)&bm.run, sizeof(float) * R_));
        ICZ_TRY(alloc((void**)&bm.seqs[0], sizeof(int32_t) * R_ * L_));
        ICZ_TRY(alloc((void**)&bm.seqs[1], sizeof(int32_t) * R_ * L_));
        ICZ_TRY(alloc((void**)&bm.src_row, sizeof(int32_t) * R_));
        ICZ_TRY(alloc((void**)&bm.img_of_row, sizeof(int32_t) * R_));
        ICZ_TRY(alloc((void**)&bm.best_score, sizeof(float) * R_));
        ICZ_TRY(alloc((void**)&bm.best_len, sizeof(int) * R_));
        ICZ_TRY(alloc((void**)&bm.has_complete, sizeof(int) * R_));
        ICZ_TRY(alloc((void**)&bm.best_seq, sizeof(int32_t) * R_ * L_));
        ICZ_TRY(alloc((void**)&bm.n_live, sizeof(int) * 260));
        ICZ_CHECK_HIP(hipHostMalloc((void**)&bm.n_live_host, sizeof(int) * 4, 0));
        bm.cap_rows = (int)R_;
        bm.cap_L = (int)L_;
    }
    ICZ_TRY(prologue(feats, n_img, st));
    ICZ_TRY(zero_state(rows, 0, st));
    ICZ_CHECK_HIP(hipMemsetAsync(bm.n_live, 0, sizeof(int) * 260, st));
    ICZ_CHECK_HIP(hipMemsetAsync(bm.run, 0, sizeof(float) * rows, st));
    hipLaunchKernelGGL(beam_init_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, n_img, k, L, bm.n_act, bm.seqs[0], bm.img_of_row, it,
                       bm.has_complete, bm.best_score);
    int sb = 0, steps_done = 0;
    for (int step = 1; step <= max_steps; ++step) {
        StepIO s = {};
        s.rows = rows; s.feats = feats; s.img_of_row = bm.img_of_row; s.it = it;
        s.h1_in = h1[0]; s.c1_in = c1[0]; s.h2_in = h2[0]; s.c2_in = c2[0];
        s.h1_out = h1[1]; s.c1_out = c1[1]; s.h2_out = h2[1]; s.c2_out = c2[1];
        ICZ_TRY(this->step(s, st));
        BeamArgs a = {logits, dims.V, pad_vocab(dims.V), k, step, L, bm.n_act, bm.run, bm.seqs[sb], bm.seqs[sb ^ 1], bm.src_row, it,
                      bm.best_score, bm.best_len, bm.best_seq, bm.has_complete, bm.n_live + step};
        hipLaunchKernelGGL(beam_step_kernel, dim3(n_img), dim3(256), 0, st, a);
        hipLaunchKernelGGL(beam_gather_kernel, dim3(cdiv(H, 1024), rows), dim3(256), 0, st, bm.src_row, H, h1[1], c1[1], h2[1], c2[1],
                           h1[0], c1[0], h2[0], c2[0]);
        sb ^= 1;
        steps_done = step;
        // every few steps ask the device whether any image still has live beams (one 4-byte read-back)
        if (step >= 6 && (step % 3) == 0 && step < max_steps) {
            ICZ_CHECK_HIP(hipMemcpyAsync(bm.n_live_host, bm.n_live + step, sizeof(int), hipMemcpyDeviceToHost, st));
            ICZ_CHECK_HIP(hipStreamSynchronize(st));
            if (bm.n_live_host[0] == 0) break;
        }
    }
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(n_img), dim3(64), 0, st, k, L, steps_done, bm.n_act, bm.run, bm.seqs[sb],
                       bm.has_complete, bm.best_len, bm.best_seq, seqs_out, lens_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

using namespace icz;
extern "C" int icz_butd_beam_search(icz_butd_t* h, const float* feats, int32_t n_img, int32_t beam, int32_t max_steps,
                                    float* seqs_out, int32_t* lens_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->beam_search(feats, n_img, beam, max_steps, seqs_out, lens_out, (hipStream_t)stream);
}
