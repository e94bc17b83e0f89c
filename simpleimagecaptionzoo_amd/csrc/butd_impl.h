// Internal declarations of the BUTD decoder handle.
#pragma once
#include <vector>
#include <stdint.h>

#include "butd_kernels.h"
#include "gemm_f32.h"

namespace icz {

// Inputs / outputs of one decoder step.  Null outputs fall back to the handle's scratch buffers.
struct StepIO {
    int rows;
    const float* feats;              // [n_img, R, D]
    const int32_t* img_of_row;       // beam search: image of each decoder row; null = identity
    int rows_per_img;                // beam search: > 1 = the rows img * rows_per_img + b belong to image img (img_of_row agrees)
    const int64_t* it;               // [rows] input token ids
    bool emb_ready;                  // the embedding of `it` is already in the emb buffer (skip embed_kernel)
    const float *h1_in, *c1_in, *h2_in, *c2_in;
    float *h1_out, *c1_out, *h2_out, *c2_out;
    float* emb_out;                  // [rows,E]
    float* gates_td_out;             // [rows,4H] activated gates (backward)
    float* gates_lm_out;
    float* dec_ctx_out;              // [rows,A]
    float* alpha_out;                // [rows,R]
    float* alpha_out2; int alpha2_stride;   // second copy laid out [rows, T, R] (caller's alphas tensor)
    float* ctx_out;                  // [rows,D]
    float* h2drop_out;               // [rows,H]
    float* logits_out;               // [rows,V]
    int logits_ld;                   // row stride of logits_out (0 = V)
    int* pred_nsplit;                // non-null: the caller's consumer of the logits can sum split-K slabs (argmax_part_kernel,
                                     // sample_select_kernel): step() may leave the predict GEMM's slabs [ns][rows][Vp] in the
                                     // chain's workspace instead of finished logits and reports ns here (1 = logits_out is final)
    float* ws_alt;                   // split-K slab workspace / attention scores of this chain (null = the handle's):
    float* scores_alt;               // two decode chains running concurrently must not share them
    DropCfg drop_emb, drop_att, drop_out;
    bool skip_predict;               // teacher-forced XE forward: the vocabulary projection of ALL time steps is one GEMM after the loop
    const int* live;                 // sampled rollout: the count of unfinished rows after the previous step; 0 = every kernel of this step
                                     // returns at entry (step_dead, icz_common.h: the reference's break, BUTD_Model.py:233)
};

// Activations kept by a training-mode forward (slot t = time step, slot stride = B rows) and backward scratch.
struct TrainBuf {
    int B = 0, T = 0;
    int64_t* tok = nullptr;                                   // [(T+1), B] input token of each step
    float *emb = nullptr, *h1 = nullptr, *c1 = nullptr, *h2 = nullptr, *c2 = nullptr;   // states: [(T+1), B, H], slot 0 = zeros
    float *gtd = nullptr, *glm = nullptr, *dec = nullptr, *alpha = nullptr, *ctx = nullptr, *h2d = nullptr, *logit = nullptr;
    int32_t* draw = nullptr; float* lse = nullptr;
    uint8_t* unf = nullptr; int* nunf = nullptr; float *coef = nullptr, *loss_rows = nullptr;
    uint8_t* gunf = nullptr; int* gnunf = nullptr;            // the same for the greedy baseline of an SCST step (greedy_chain, scst = true)
    int* live_rows = nullptr;                                 // (steps the sampled rollout ran) x B: row limit of the backward pass's batched GEMMs
    int* nany = nullptr;                                      // merged chain: unfinished sampled rows + greedy rows that have not ended, per step
    int32_t* img2 = nullptr;                                  // merged chain: image of each decoder row (row r of 2 B -> image r mod B)
    float *dGtd = nullptr, *dGlm = nullptr, *dDec = nullptr, *dEmb = nullptr, *dH2d = nullptr, *dEnc = nullptr;
    float *dwaff = nullptr, *dalpha = nullptr, *dS = nullptr, *dGsum = nullptr;
    float *dc1[2] = {nullptr, nullptr}, *dc2[2] = {nullptr, nullptr};
    float* X[4] = {nullptr, nullptr, nullptr, nullptr}; size_t xfloats = 0;
    float *dWp = nullptr, *dWenc = nullptr, *dWdec = nullptr, *dWaff = nullptr, *scalars = nullptr;
    float* wslab = nullptr; size_t wslab_floats = 0;      // split-K slabs of the two attention weight gradients (gemm_tn_split, round 6)
    int* scalars_i = nullptr; int scalars_i_cap = 0;
};

struct BeamBuf {
    int cap_rows = 0, cap_L = 0;
    int* n_act = nullptr; float* run = nullptr; int32_t* seqs[2] = {nullptr, nullptr};
    int32_t *src_row = nullptr, *img_of_row = nullptr, *best_seq = nullptr;
    float* best_score = nullptr; int *best_len = nullptr, *has_complete = nullptr, *n_live = nullptr, *n_live_host = nullptr;
    float* feat_rows = nullptr;       // NIC: image embedding replicated per beam row
    float* cand_val = nullptr; int* cand_idx = nullptr;     // [rows, BEAM_MAX_K] per-row candidates of one step
};

struct Butd {
    static constexpr int TARGET_WGS = 512;   // ~2 workgroups per CU on 256 CUs
    static constexpr int ATT_PARTS = 4;
    static constexpr int STEP_WGS = 256;     // skinny decoder-step GEMMs: split-K for ~1 workgroup per CU
    static constexpr int ARGMAX_PARTS = 8;
    icz_butd_dims dims;
    icz_butd_params P;
    bool bound = false, fresh = false;
    std::vector<void*> allocs;
    std::vector<void*> tallocs;          // training buffers (TrainBuf): re-allocated when a batch needs more rows / steps
    bool alloc_train = false;            // alloc() target: tallocs instead of allocs

    // weight-normed weights (w = g v / ||v||) and the row norms ||v||
    float *w_enc = nullptr, *w_dec = nullptr, *w_aff = nullptr, *w_pred = nullptr;
    float *n_enc = nullptr, *n_dec = nullptr, *n_aff = nullptr, *n_pred = nullptr;
    // per-image hoisted tensors
    float *mean = nullptr, *premean = nullptr, *enc_ctx = nullptr;
    // recurrent state (double buffered) and per-step scratch
    float *h1[2], *c1[2], *h2[2], *c2[2];
    float *emb = nullptr, *ctx = nullptr, *scores = nullptr, *alpha = nullptr, *h2drop = nullptr, *logits = nullptr;
    int64_t* it = nullptr;
    float* amax_val = nullptr; int* amax_idx = nullptr;
    uint64_t* d_seed = nullptr;          // Philox seed of the current training-mode call (device resident)
    float ss_prob = 0.f;                 // scheduled sampling in xe_forward (icz_butd_set_scheduled_sampling)
    const float* ss_gate = nullptr; const float* ss_draw = nullptr;      // explicit [T, B] uniforms (device) or Philox
    float* d_msum_global = nullptr;      // data-parallel loss normaliser (0 = use the local one)
    float* ws = nullptr;
    size_t ws_floats = 0;
    // Transposed copies of the LSTM weights: the per-step dgrad products of BPTT, dx = dy W, run as NT products on W^T through the
    // resident-activation kernel (weights streamed once, split-precision MFMA) instead of the fp32 NN kernel.  Allocated with the
    // first training buffers where the sizes fit that kernel, refreshed once per optimiser step (refresh / first backward after it).
    float *wt_lm_ih = nullptr, *wt_lm_hh = nullptr, *wt_td_ih_h2 = nullptr, *wt_td_hh = nullptr;
    bool wt_fresh = false;
    bool wt_possible() const;
    int refresh_transposes(hipStream_t st);

    ~Butd();
    int alloc(void** p, size_t bytes);
    int init(const icz_butd_dims& d);
    int refresh(hipStream_t st);
    int gemm_nt(GemmArgs& g, int* nsplit_out, hipStream_t st);
    int prologue(const float* feats, int n_img, hipStream_t st);
    int step(const StepIO& s, hipStream_t st);
    int zero_state(int rows, int which, hipStream_t st);
    int greedy(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st);

    // hipGraph cache: a whole rollout / backward is ~300-700 launches of 2-30 us kernels; replaying a captured graph
    // removes the per-launch host cost and shrinks the inter-kernel gaps.  Keyed by every pointer / size baked into
    // the captured kernel arguments, so it only pays when the caller reuses its buffers (the Engine does).
    struct GraphEntry { std::vector<uintptr_t> key; hipGraphExec_t exec; uint64_t last_use; };
    std::vector<GraphEntry> graphs;
    bool use_graphs = false;
    bool concurrent = true;      // run independent chains on side streams (off: one stream, for per-kernel timing)
    hipStream_t cap_st = nullptr;
    uint64_t tick = 0;
    template <class F> int run_cached(const std::vector<uintptr_t>& key, hipStream_t st, F&& fn);
    void clear_graphs();                 // captured kernel arguments hold parameter / buffer addresses: drop them when those change
    int greedy_impl(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st);
    int sample_impl(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st);
    int greedy_chain(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st, bool scst = false);
    int sample_chain(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st, int row0 = 0, int64_t* ids_out = nullptr);
    int rollouts(const float* feats, int B, int T, const icz_rng* r, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st);
    int rollouts_impl(const float* feats, int B, int T, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st);
    hipStream_t side_st = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t low_st = nullptr;        // lowest-priority stream for work overlapped with the BPTT chain
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;   // the attention block's tail beside the LSTM weight gradients (bptt)
    icz_grad_ready_cb grad_cb = nullptr; void* grad_cb_user = nullptr;   // DP overlap hook (icz_butd_set_grad_callback)
    int sample_backward_impl(const float* reward, const icz_butd_params& G, float* loss_out, float* mask_sum_out, hipStream_t st,
                             int phases = 0xF, bool fire_cb = true);
    bool bptt_joined = false;            // the predict-gradient branch has been joined (bptt phases)
    int merge_small = 8;                 // option "merge_small" = n: SCST rollouts of <= n images (n <= 32) run as ONE chain of 2 B decoder rows
                                         // (sample_chain with row0 = B); 0 = never.  Default 8: measured round 5 (EXPERIMENTS.md), the
                                         // merged chain saves 0.2 ms of rollouts at every size but costs the backward pass 0.06 / 0.15 /
                                         // 0.37 ms at 8 / 16 / 32 images (its batched GEMMs then run over 2 B rows per step)
    int cur_rows = 0, cur_row0 = 0;      // rows per step slot of the stored forward pass and the offset of the rows the backward pass works on
                                         // (a merged chain stores 2 B rows per step, the sampled rollout's are rows B .. 2 B - 1)
    bool early_out = true;               // option "early_out": 0 = run the steps behind the reference's break as rounds 1 - 4 did (A/B)
    bool bptt_early_out = false;         // backward of a sampled rollout: steps behind the reference's break return at entry (bptt)
    bool small_nt = true;                // option "small_nt": BPTT steps of <= 32 rows take their dgrad products on the transposed weight copies (fp32 NT kernel)

    // beam search (butd_beam.hip)
    BeamBuf bm;
    int beam_search(const float* feats, int n_img, int k, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st);

    // training paths (butd_train.hip)
    TrainBuf tb;
    icz_rng rng = {};
    int mode = 0;                 // 0 none, 1 sample rollout stored, 2 XE forward stored
    int cur_B = 0, cur_T = 0, cur_L = 0, n_tokens = 0;
    bool cur_train = false;
    const float* cur_feats = nullptr;
    const int64_t* cur_seq = nullptr; const float* cur_logp = nullptr; const int64_t* cur_captions = nullptr;
    std::vector<int> rows_t;
    int ensure_train(int B, int T);
    int train_step(const float* feats, int rows, int Bs, int t, bool train, hipStream_t st, bool emb_ready = false, int* pred_nsplit = nullptr,
                   bool skip_predict = false, const int* live = nullptr, int row0 = 0);
    int sample(const float* feats, int B, int T, const icz_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st);
    int sample_mask_sum(float* out, hipStream_t st);
    int sample_backward(const float* reward, const icz_butd_params* G, float* loss_out, float* mask_sum_out,
                        float mask_sum_global, hipStream_t st);
    int xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_rng* r,
                   int train, float* packed_out, hipStream_t st);
    int sample_backward_dlogp(const float* dlogp, const icz_butd_params* G, hipStream_t st);
    int xe_backward_dlogits(const float* dpacked, const icz_butd_params* G, hipStream_t st);
    int upload_pack_index(hipStream_t st);
    int xe_backward(float smoothing, const icz_butd_params* G, float* loss_out, float n_tokens_global, hipStream_t st);
    int gemm_auto(GemmLayout layout, GemmArgs& g, float* slab, size_t slab_floats, int* ns_out, hipStream_t st);
    int wgrad(const float* dY, int ldy, int M, const float* X, int ldx, int N, int K, float* out, int ldo, hipStream_t st, const int* rows_live = nullptr);
    int bptt_prelude(hipStream_t st);
    int colsum(const float* X, int K, int N, int ldx, float* out, hipStream_t st);
    int bptt(const icz_butd_params& G, hipStream_t st, int phases = 0xF, bool fire_cb = true);
};

void gemm_set_capturing(bool on);
bool gemm_prof_on();                  // gemm_f32.hip: event timing active (graphs captured now contain event nodes)

// The same hipGraph cache as Butd::run_cached for the other decoders (AoA): capture on first use of a key, replay afterwards.
struct GraphCache {
    struct Entry { std::vector<uintptr_t> key; hipGraphExec_t exec; uint64_t last_use; };
    std::vector<Entry> graphs;
    hipStream_t cap_st = nullptr;
    uint64_t tick = 0;
    void clear() {
        for (auto& e : graphs) (void)hipGraphExecDestroy(e.exec);
        graphs.clear();
    }
    ~GraphCache() {
        clear();
        if (cap_st) (void)hipStreamDestroy(cap_st);
    }
    template <class F>
    int run(const std::vector<uintptr_t>& key_in, hipStream_t st, F&& fn) {
        ++tick;
        std::vector<uintptr_t> key = key_in;
        key.push_back((gemm_prof_on() ? 1 : 0) + 2 * (uintptr_t)(gemm_big_switch() + 2));      // the tile-configuration override changes the captured launches
        for (auto& e : graphs)
            if (e.key == key) {
                e.last_use = tick;
                ICZ_CHECK_HIP(hipGraphLaunch(e.exec, st));
                return ICZ_OK;
            }
        if (!cap_st) ICZ_CHECK_HIP(hipStreamCreateWithFlags(&cap_st, hipStreamNonBlocking));
        ICZ_CHECK_HIP(hipStreamBeginCapture(cap_st, hipStreamCaptureModeThreadLocal));
        gemm_set_capturing(true);
        const int status = fn(cap_st);
        gemm_set_capturing(false);
        hipGraph_t g = nullptr;
        hipError_t ce = hipStreamEndCapture(cap_st, &g);
        if (status != ICZ_OK) { if (g) (void)hipGraphDestroy(g); return status; }
        if (ce != hipSuccess || !g) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return ICZ_ERR_HIP; }
        hipGraphExec_t exec = nullptr;
        hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ie != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ie)); return ICZ_ERR_HIP; }
        if (graphs.size() >= 16) {      // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < graphs.size(); ++i) if (graphs[i].last_use < graphs[lru].last_use) lru = i;
            (void)hipGraphExecDestroy(graphs[lru].exec);
            graphs.erase(graphs.begin() + lru);
        }
        graphs.push_back({key, exec, tick});
        ICZ_CHECK_HIP(hipGraphLaunch(exec, st));
        return ICZ_OK;
    }
};

template <class F>
int Butd::run_cached(const std::vector<uintptr_t>& key_in, hipStream_t st, F&& fn) {
    if (!use_graphs) return fn(st);
    ++tick;
    std::vector<uintptr_t> key = key_in;
    key.push_back((concurrent ? 1 : 0) + (early_out ? 2 : 0) + 4 * merge_small + (small_nt ? 256 : 0));          // flags that change the captured launch sequence
    key.push_back((gemm_prof_on() ? 1 : 0) + 2 * (uintptr_t)(gemm_big_switch() + 2));      // the tile-configuration override changes the captured launches
    for (auto& e : graphs)
        if (e.key == key) {
            e.last_use = tick;
            ICZ_CHECK_HIP(hipGraphLaunch(e.exec, st));
            return ICZ_OK;
        }
    if (!cap_st) ICZ_CHECK_HIP(hipStreamCreateWithFlags(&cap_st, hipStreamNonBlocking));
    ICZ_CHECK_HIP(hipStreamBeginCapture(cap_st, hipStreamCaptureModeThreadLocal));
    gemm_set_capturing(true);
    const int status = fn(cap_st);
    gemm_set_capturing(false);
    hipGraph_t g = nullptr;
    hipError_t ce = hipStreamEndCapture(cap_st, &g);
    if (status != ICZ_OK) { if (g) (void)hipGraphDestroy(g); return status; }
    if (ce != hipSuccess || !g) { set_error("hipStreamEndCapture failed: %s", hipGetErrorString(ce)); return ICZ_ERR_HIP; }
    hipGraphExec_t exec = nullptr;
    hipError_t ie = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) { set_error("hipGraphInstantiate failed: %s", hipGetErrorString(ie)); return ICZ_ERR_HIP; }
    if (graphs.size() >= 24) {      // evict the least recently used entry
        size_t lru = 0;
        for (size_t i = 1; i < graphs.size(); ++i) if (graphs[i].last_use < graphs[lru].last_use) lru = i;
        (void)hipGraphExecDestroy(graphs[lru].exec);
        graphs.erase(graphs.begin() + lru);
    }
    graphs.push_back({key, exec, tick});
    ICZ_CHECK_HIP(hipGraphLaunch(exec, st));
    return ICZ_OK;
}

}  // namespace icz
