// NIC decoder (Models/NIC_Model.py:39-212): single LSTM over a plain embedding; the image embedding enters through one
// LSTM step from the zero state.  Same kernels as the BUTD decoder (GEMMs, LSTM pointwise, select / beam / loss
// kernels); only the launch sequences differ.
#include <math.h>

#include <vector>

#include "beam_kernels.h"
#include "butd_impl.h"

namespace icz {

struct Nic {
    static constexpr int STEP_WGS = 256, TARGET_WGS = 512, ARGMAX_PARTS = 8;
    icz_nic_dims dims;
    icz_nic_params P;
    bool bound = false, fresh = false;
    std::vector<void*> allocs;
    int Vp = 0;
    float *w_pred = nullptr, *n_pred = nullptr, *zeros = nullptr;
    float *h[2], *c[2];
    float *emb = nullptr, *hdrop = nullptr, *logits = nullptr, *ws = nullptr;
    size_t ws_floats = 0;
    int64_t* it = nullptr;
    float* amax_val = nullptr; int* amax_idx = nullptr;
    uint64_t* d_seed = nullptr; float* d_msum = nullptr;
    float ss_prob = 0.f; const float* ss_gate = nullptr; const float* ss_draw = nullptr;      // scheduled sampling in xe_forward
    // training buffers (slot stride = capacity rows): th/tc slot 0 = zeros, slot 1 = after the image step, slot t+2 =
    // after token step t; tg / dG slot 0 = image step, slot t+1 = token step t
    int tcap_B = 0, tcap_T = 0;          // capacity of the training buffers (grown on demand by ensure_train)
    std::vector<void*> tallocs; bool alloc_train = false;
    int64_t* tok = nullptr;
    float *th = nullptr, *tc = nullptr, *temb = nullptr, *tg = nullptr, *thd = nullptr, *tlogit = nullptr;
    float *dG = nullptr, *dHd = nullptr, *dEmb = nullptr, *dcb[2] = {nullptr, nullptr}, *X = nullptr, *dWp = nullptr;
    float *coef = nullptr, *lse = nullptr, *loss_rows = nullptr;
    int32_t* draw = nullptr; uint8_t* unf = nullptr; int* nunf = nullptr; int* pack_idx = nullptr;
    size_t xfloats = 0;
    BeamBuf bm;
    icz_rng rng = {};
    int mode = 0, cur_B = 0, cur_T = 0, cur_L = 0, n_tokens = 0;
    bool cur_train = false;
    const float* cur_feats = nullptr; const int64_t* cur_seq = nullptr; const float* cur_logp = nullptr;
    const int64_t* cur_captions = nullptr;
    std::vector<int> rows_t;

    ~Nic() {
        if (bm.n_live_host) (void)hipHostFree(bm.n_live_host);
        for (void* p : tallocs) (void)hipFree(p);
        for (void* p : allocs) (void)hipFree(p);
    }
    int alloc(void** p, size_t bytes) {
        ICZ_CHECK_HIP(hipMalloc(p, bytes ? bytes : 16));
        ICZ_CHECK_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
        (alloc_train ? tallocs : allocs).push_back(*p);
        return ICZ_OK;
    }
    int init(const icz_nic_dims& d);
    int refresh(hipStream_t st);
    int image_step(const float* feats, int rows, float* h_out, float* c_out, float* gates_out, hipStream_t st);
    int token_step(int rows, const int64_t* tokens, bool emb_ready, const float* h_in, const float* c_in, float* h_out, float* c_out,
                   float* emb_out, float* gates_out, float* hdrop_out, float* logits_out, DropCfg drop_out, hipStream_t st, int* pred_nsplit = nullptr,
                   const int* live = nullptr);
    bool early_out = true, bptt_early_out = false;      // rollout / BPTT steps behind sample_rl's break (NIC_Model.py:150) return at entry
    int greedy(const float* feats, int B, int T, int64_t* ids_out, hipStream_t st);
    int ensure_train(int B, int T);
    int sample(const float* feats, int B, int T, const icz_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st);
    int sample_backward(const float* reward, const icz_nic_params* G, float* dfeats, float* loss_out, float* msum_out, float msum_global, hipStream_t st);
    int xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_rng* r, int train,
                   float* packed_out, hipStream_t st);
    int xe_backward(float smoothing, const icz_nic_params* G, float* dfeats, float* loss_out, float n_tokens_global, hipStream_t st);
    int bptt(const icz_nic_params& G, float* dfeats, hipStream_t st);
    int colsum(const float* Xm, int K, int N, int ldx, float* out, hipStream_t st);
    int nn(const float* A, int lda, int M, int K, const float* Bm, int ldb, int N, float* out, int* ns, int target, hipStream_t st,
           const int* live = nullptr);
    int tn(const float* dY, int ldy, int M, const float* Xm, int ldx, int N, int K, float* out, int ldo, int accumulate, hipStream_t st);
    int beam_search(const float* feats, int n_img, int k, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st);
};

int Nic::init(const icz_nic_dims& d) {
    dims = d;
    ICZ_REQUIRE(d.E % 4 == 0 && d.H % 4 == 0 && d.V > 3 && d.max_rows > 0 && d.max_len > 0, "nic: bad dimensions");
    Vp = pad_vocab(d.V);
    const size_t rows = d.max_rows, H = d.H, E = d.E;
    ICZ_TRY(alloc((void**)&w_pred, sizeof(float) * Vp * H));
    ICZ_TRY(alloc((void**)&n_pred, sizeof(float) * d.V));
    ICZ_TRY(alloc((void**)&zeros, sizeof(float) * rows * H));
    for (int i = 0; i < 2; ++i) { ICZ_TRY(alloc((void**)&h[i], sizeof(float) * rows * H)); ICZ_TRY(alloc((void**)&c[i], sizeof(float) * rows * H)); }
    ICZ_TRY(alloc((void**)&emb, sizeof(float) * rows * E));
    ICZ_TRY(alloc((void**)&hdrop, sizeof(float) * rows * H));
    ICZ_TRY(alloc((void**)&logits, sizeof(float) * rows * Vp));
    ICZ_TRY(alloc((void**)&it, sizeof(int64_t) * rows));
    ICZ_TRY(alloc((void**)&amax_val, sizeof(float) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&amax_idx, sizeof(int) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&d_seed, 16));
    ICZ_TRY(alloc((void**)&d_msum, 16));
    const size_t nmax = 4 * H > (size_t)Vp ? 4 * H : (size_t)Vp;
    ws_floats = (size_t)TARGET_WGS * 4096 * 2 + rows * nmax;
    {   // the resident decoder-step GEMM (33..128 rows) leaves one slab per 256-deep k range of K = E + H (Butd::init's rule)
        const size_t kmax = (size_t)d.E + H, r128 = rows < 128 ? rows : 128;
        const size_t need = (kmax / 256 + 1) * r128 * 4 * H;
        if (need > ws_floats) ws_floats = need;
    }
    ICZ_TRY(alloc((void**)&ws, sizeof(float) * ws_floats));
    ICZ_CHECK_HIP(hipDeviceSynchronize());      // alloc() zero-fills on the NULL stream; callers use non-blocking streams (see ensure_train)
    return ICZ_OK;
}

int Nic::refresh(hipStream_t st) {
    ICZ_REQUIRE(bound, "nic: parameters not bound");
    hipLaunchKernelGGL(weight_norm_kernel, dim3(cdiv(dims.V, 4)), dim3(256), 0, st, P.predict_v, P.predict_g, w_pred, n_pred, dims.V, dims.H);
    ICZ_CHECK_HIP(hipGetLastError());
    fresh = true;
    return ICZ_OK;
}

// h, c = LSTMCell(features, (0, 0))   (NIC_Model.py:52-56)
int Nic::image_step(const float* feats, int rows, float* h_out, float* c_out, float* gates_out, hipStream_t st) {
    const int H = dims.H, E = dims.E;
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {feats, P.w_ih, E, E, E, nullptr};
    g.M = rows; g.N = 4 * H; g.out = ws; g.ldo = 4 * H;
    g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, STEP_WGS), ws_floats);
    ICZ_REQUIRE(gemm_slab_floats(g.M, g.N, g.nsplit) <= ws_floats, "nic: workspace too small");
    ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    LstmPointArgs a = {ws, g.nsplit, nullptr, nullptr, P.b_ih, P.b_hh, zeros, h_out, c_out, gates_out, nullptr, rows, H};
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    launch_lstm_point(a, off, st);
    return ICZ_OK;
}

// embed -> LSTMCell -> predict(dropout(h))   (NIC_Model.py:112-114)
// live: step_dead (icz_common.h) -- every kernel of a rollout step behind the reference's break returns at entry
int Nic::token_step(int rows, const int64_t* tokens, bool emb_ready, const float* h_in, const float* c_in, float* h_out, float* c_out,
                    float* emb_out, float* gates_out, float* hdrop_out, float* logits_out, DropCfg drop_out, hipStream_t st, int* pred_nsplit,
                    const int* live) {
    const int H = dims.H, E = dims.E, V = dims.V;
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    if (!emb_ready) hipLaunchKernelGGL(embed_kernel, dim3(cdiv(E, 1024), rows), dim3(256), 0, st, P.embed_weight, tokens, emb_out, rows, E, off, 0, live);
    GemmArgs g = {};
    g.nseg = 2;
    g.live = live;
    g.seg[0] = {emb_out, P.w_ih, E, E, E, nullptr};
    g.seg[1] = {h_in, P.w_hh, H, H, H, nullptr};
    g.M = rows; g.N = 4 * H; g.out = ws; g.ldo = 4 * H;
    g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, STEP_WGS), ws_floats);
    ICZ_REQUIRE(gemm_slab_floats(g.M, g.N, g.nsplit) <= ws_floats, "nic: workspace too small");
    ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    LstmPointArgs a = {ws, g.nsplit, nullptr, nullptr, P.b_ih, P.b_hh, c_in, h_out, c_out, gates_out, hdrop_out, rows, H, live};
    launch_lstm_point(a, drop_out, st);
    ICZ_TRY(gemm_predict(hdrop_out, H, w_pred, P.predict_b, rows, V, Vp, logits_out, Vp, ws, ws_floats, pred_nsplit, st, live));
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Nic::greedy(const float* feats, int B, int T, int64_t* ids_out, hipStream_t st) {
    ICZ_REQUIRE(feats && ids_out && B > 0 && B <= dims.max_rows && T > 0, "nic greedy: bad arguments");
    ICZ_REQUIRE(fresh, "nic: call icz_nic_refresh_weights after binding/updating parameters");
    ICZ_TRY(image_step(feats, B, h[0], c[0], nullptr, st));
    hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, it, (int64_t)1, B);
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        int pns = 1;
        ICZ_TRY(token_step(B, it, t > 0, h[cur], c[cur], h[cur ^ 1], c[cur ^ 1], emb, nullptr, hdrop, logits, off, st, &pns));
        if (pns > 1)
            hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(1024), 0, st, (const float*)ws, dims.V, Vp, pns, (size_t)B * Vp,
                               (const float*)P.predict_b, P.embed_weight, dims.E, emb, it, ids_out, T, t, 0);
        else {
            hipLaunchKernelGGL(argmax_part_kernel, dim3(B, ARGMAX_PARTS), dim3(256), 0, st, logits, dims.V, Vp, ARGMAX_PARTS, amax_val, amax_idx);
            hipLaunchKernelGGL(embed_argmax_kernel, dim3(cdiv(dims.E, 1024), B), dim3(256), 0, st, amax_val, amax_idx, ARGMAX_PARTS,
                               P.embed_weight, dims.E, emb, it, ids_out, T, t, 0);
        }
        cur ^= 1;
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// Training buffers sized by what the batches ask for (the reference never truncates captions, Datasets.py:47-51: the step
// count of an XE batch is only known when it arrives); growing re-allocates them all.
int Nic::ensure_train(int Bq, int Tq) {
    if (Bq <= tcap_B && Tq <= tcap_T) return ICZ_OK;
    ICZ_REQUIRE(Tq <= XE_MAX_T, "nic: %d steps exceed the limit of %d", Tq, XE_MAX_T);
    if (tcap_B > Bq) Bq = tcap_B;
    if (tcap_T > Tq) Tq = tcap_T;
    if (dims.max_len > Tq) Tq = dims.max_len;
    if (!tallocs.empty()) {
        ICZ_CHECK_HIP(hipDeviceSynchronize());
        for (void* p : tallocs) (void)hipFree(p);
        tallocs.clear();
        tcap_B = tcap_T = 0; mode = 0;
    }
    struct Scope { bool& f; Scope(bool& x) : f(x) { f = true; } ~Scope() { f = false; } } scope(alloc_train);
    const size_t B = Bq, T = Tq, H = dims.H, E = dims.E;
    const size_t TB = T * B;
    ICZ_TRY(alloc((void**)&tok, sizeof(int64_t) * (TB + B)));
    ICZ_TRY(alloc((void**)&th, sizeof(float) * (TB + 2 * B) * H));
    ICZ_TRY(alloc((void**)&tc, sizeof(float) * (TB + 2 * B) * H));
    ICZ_TRY(alloc((void**)&temb, sizeof(float) * TB * E));
    ICZ_TRY(alloc((void**)&tg, sizeof(float) * (TB + B) * 4 * H));
    ICZ_TRY(alloc((void**)&thd, sizeof(float) * TB * H));
    ICZ_TRY(alloc((void**)&tlogit, sizeof(float) * TB * Vp));
    ICZ_TRY(alloc((void**)&dG, sizeof(float) * (TB + B) * 4 * H));
    ICZ_TRY(alloc((void**)&dHd, sizeof(float) * TB * H));
    ICZ_TRY(alloc((void**)&dEmb, sizeof(float) * TB * E));
    ICZ_TRY(alloc((void**)&dcb[0], sizeof(float) * B * H));
    ICZ_TRY(alloc((void**)&dcb[1], sizeof(float) * B * H));
    xfloats = (size_t)TARGET_WGS * 4096 * 2 + B * H;
    ICZ_TRY(alloc((void**)&X, sizeof(float) * xfloats));
    ICZ_TRY(alloc((void**)&dWp, sizeof(float) * (size_t)Vp * H));
    ICZ_TRY(alloc((void**)&coef, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&lse, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&loss_rows, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&draw, sizeof(int32_t) * TB));
    ICZ_TRY(alloc((void**)&unf, B));
    ICZ_TRY(alloc((void**)&nunf, sizeof(int) * T));
    ICZ_TRY(alloc((void**)&pack_idx, sizeof(int) * 2 * T));
    // The hipMemset calls above run on the NULL stream; callers enqueue on NON-BLOCKING streams (torch's), which are not ordered behind
    // it: without this, a kernel of the first call after a (re)allocation could run BEFORE the zero-fill of its buffer and then be
    // wiped by it (round 5: sample_init_kernel's unfinished flags, seen as an all-zero rollout in 1 of 3 five-rank runs).
    ICZ_CHECK_HIP(hipDeviceSynchronize());
    tcap_B = Bq; tcap_T = Tq;
    return ICZ_OK;
}

static DropCfg nic_drop(const uint64_t* seed_p, bool train, const uint8_t* base, size_t per_step, int t) {
    DropCfg d = {0, nullptr, seed_p, RNG_OUT, (uint32_t)t};
    if (!train) return d;
    if (base) { d.mode = 1; d.mask = base + per_step * t; }
    else d.mode = 2;
    return d;
}

int Nic::sample(const float* feats, int B, int T, const icz_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seq_out && logp_out && r && B > 0 && B <= dims.max_rows && T > 0, "nic sample: bad arguments");
    ICZ_REQUIRE(fresh, "nic: call icz_nic_refresh_weights after binding/updating parameters");
    ICZ_TRY(ensure_train(B, T));
    rng = *r;
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 1; cur_B = B; cur_T = T; cur_train = true; cur_feats = feats; cur_seq = seq_out; cur_logp = logp_out;
    rows_t.assign(T, B);
    const size_t H = dims.H, E = dims.E, sH = (size_t)B * H;
    ICZ_CHECK_HIP(hipMemsetAsync(th, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tc, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(unf, 1, B, st));
    ICZ_CHECK_HIP(hipMemsetAsync(nunf, 0, sizeof(int) * T, st));
    hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, tok, (int64_t)1, B);
    ICZ_TRY(image_step(feats, B, th + sH, tc + sH, tg, st));
    for (int t = 0; t < T; ++t) {
        const size_t slot = (size_t)t * B;
        int pns = 1;
        ICZ_TRY(token_step(B, tok + slot, false, th + (slot + B) * H, tc + (slot + B) * H, th + (slot + 2 * B) * H, tc + (slot + 2 * B) * H,
                           temb + slot * E, tg + (slot + B) * 4 * H, thd + slot * H, tlogit + slot * Vp,
                           nic_drop(d_seed, true, rng.out_mask, sH, t), st, &pns, (t > 0 && early_out) ? nunf + (t - 1) : nullptr));
        SampleSelArgs a = {};
        a.logits = tlogit + slot * Vp; a.V = dims.V; a.ldl = Vp;
        if (pns > 1) { a.logits = ws; a.ns = pns; a.slab_stride = (size_t)B * Vp; a.bias = P.predict_b; a.logits_store = tlogit + slot * Vp; }
        a.uniforms = rng.uniforms ? rng.uniforms + slot : nullptr;
        a.seed_p = d_seed; a.t = t; a.T = T;
        a.unfinished = unf; a.n_unfinished = nunf; a.seq_out = seq_out; a.logp_out = logp_out;
        a.it_next = tok + slot + B; a.draw_out = draw + slot; a.lse_out = lse + slot;
        launch_sample_select(st, B, a);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Nic::sample_backward(const float* reward, const icz_nic_params* G, float* dfeats, float* loss_out, float* msum_out, float msum_global,
                         hipStream_t st) {
    ICZ_REQUIRE(mode == 1, "nic: no rollout stored (call icz_nic_sample first)");
    ICZ_REQUIRE(reward && G, "nic sample_backward: null argument");
    const int B = cur_B, T = cur_T;
    if (msum_global >= 0.f)      // < 0: keep the device value handed over by icz_nic_set_norm_global
        hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, (uint64_t*)nullptr, (uint64_t)0, d_msum, msum_global);
    hipLaunchKernelGGL(reinforce_loss_kernel, dim3(1), dim3(256), 0, st, cur_logp, cur_seq, reward, B, T, (const float*)d_msum, coef, loss_out, msum_out);
    hipLaunchKernelGGL(reinforce_dlogits_kernel, dim3(cdiv(Vp, 256), T * B), dim3(256), 0, st, tlogit, dims.V, Vp, draw, lse, coef, B, T);
    mode = 0;
    bptt_early_out = true;
    return bptt(*G, dfeats, st);
}

__global__ void nic_captions_to_tok_kernel(const int64_t* __restrict__ cap, int B, int L, int T, int64_t* __restrict__ tok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * B) return;
    tok[i] = cap[(size_t)(i % B) * L + i / B];
}
__global__ void nic_gather_packed_kernel(const float* __restrict__ logit, int V, int ldl, int B, const int* __restrict__ row_off,
                                         const int* __restrict__ rows_t, float* __restrict__ out) {
    const int tb_ = blockIdx.y, t = tb_ / B, b = tb_ % B;
    if (b >= rows_t[t]) return;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < V) out[(size_t)(row_off[t] + b) * V + v] = logit[(size_t)tb_ * ldl + v];
}

int Nic::xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_rng* r, int train,
                    float* packed_out, hipStream_t st) {
    ICZ_REQUIRE(feats && captions && lengths && B > 0 && B <= dims.max_rows && L > 1, "nic xe_forward: bad arguments");
    ICZ_REQUIRE(fresh, "nic: call icz_nic_refresh_weights after binding/updating parameters");
    ICZ_REQUIRE(!train || r, "nic xe_forward: training mode needs an icz_rng");
    int T = 0;
    for (int b = 0; b < B; ++b) {
        ICZ_REQUIRE(lengths[b] >= 1 && lengths[b] <= L - 1, "nic xe_forward: length %d out of range 1..%d", lengths[b], L - 1);
        ICZ_REQUIRE(b == 0 || lengths[b] <= lengths[b - 1], "nic xe_forward: lengths must be sorted in decreasing order");
        if (lengths[b] > T) T = lengths[b];
    }
    ICZ_TRY(ensure_train(B, T));
    if (r) rng = *r; else rng = {};
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 2; cur_B = B; cur_T = T; cur_L = L; cur_train = train != 0; cur_feats = feats; cur_captions = captions;
    rows_t.assign(T, 0);
    n_tokens = 0;
    for (int t = 0; t < T; ++t) {
        int cnt = 0;
        for (int b = 0; b < B; ++b) cnt += lengths[b] > t;
        rows_t[t] = cnt;
        n_tokens += cnt;
    }
    const size_t H = dims.H, E = dims.E, sH = (size_t)B * H;
    ICZ_CHECK_HIP(hipMemsetAsync(th, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tc, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tlogit, 0, sizeof(float) * (size_t)T * B * Vp, st));
    hipLaunchKernelGGL(nic_captions_to_tok_kernel, dim3(cdiv(T * B, 256)), dim3(256), 0, st, captions, B, L, T, tok);
    ICZ_TRY(image_step(feats, B, th + sH, tc + sH, tg, st));
    for (int t = 0; t < T; ++t) {
        const size_t slot = (size_t)t * B;
        if (t >= 2 && ss_prob > 0.f)          // NIC_Model.py:77-89
            ICZ_TRY(ss_select_launch(st, rows_t[t], tlogit + (slot - B) * Vp, (int)Vp, dims.V, t, B, ss_prob, ss_gate, ss_draw, d_seed, tok + slot));
        ICZ_TRY(token_step(rows_t[t], tok + slot, false, th + (slot + B) * H, tc + (slot + B) * H, th + (slot + 2 * B) * H,
                           tc + (slot + 2 * B) * H, temb + slot * E, tg + (slot + B) * 4 * H, thd + slot * H, tlogit + slot * Vp,
                           nic_drop(d_seed, train != 0, rng.out_mask, sH, t), st));
    }
    if (packed_out) {
        std::vector<int> hostv(2 * T);
        int acc = 0;
        for (int t = 0; t < T; ++t) { hostv[t] = acc; hostv[T + t] = rows_t[t]; acc += rows_t[t]; }
        ICZ_CHECK_HIP(hipMemcpyAsync(pack_idx, hostv.data(), sizeof(int) * 2 * T, hipMemcpyHostToDevice, st));
        ICZ_CHECK_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(nic_gather_packed_kernel, dim3(cdiv(dims.V, 256), T * B), dim3(256), 0, st, tlogit, dims.V, Vp, B, pack_idx,
                           pack_idx + T, packed_out);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Nic::xe_backward(float smoothing, const icz_nic_params* G, float* dfeats, float* loss_out, float n_tokens_global, hipStream_t st) {
    ICZ_REQUIRE(mode == 2, "nic: no XE forward stored (call icz_nic_xe_forward first)");
    ICZ_REQUIRE(G, "nic xe_backward: null grads");
    const int B = cur_B, T = cur_T;
    const float n = n_tokens_global > 0.f ? n_tokens_global : (float)n_tokens;
    const float* n_dev = n_tokens_global < 0.f ? d_msum : nullptr;      // < 0: the device scalar handed over by *_set_*_global
    ICZ_CHECK_HIP(hipMemsetAsync(loss_rows, 0, sizeof(float) * T * B, st));
    {
        ICZ_REQUIRE(T <= XE_MAX_T, "xe_backward: %d steps exceed %d", T, XE_MAX_T);
        XeRows xr = {};
        for (int t = 0; t < T; ++t) xr.n[t] = rows_t[t];
        hipLaunchKernelGGL(xe_loss_dlogits_kernel, dim3(B, T), dim3(256), 0, st, tlogit, dims.V, Vp, cur_captions, cur_L, B, xr, smoothing, 1.0f / n, n_dev,
                           loss_rows);
    }
    if (loss_out) hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, loss_rows, T * B, 1.0f / n, n_dev, loss_out);
    mode = 0;
    bptt_early_out = false;
    return bptt(*G, dfeats, st);
}

int Nic::colsum(const float* Xm, int K, int N, int ldx, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 32)), dim3(256), 0, st, Xm, K, N, ldx, out);
    return ICZ_OK;
}

// out (dense [M,N]) = A[M,K] . B[K,N]; split-K slabs go to `ws`/`X` and are reduced into `out` unless ns is requested
int Nic::nn(const float* A, int lda, int M, int K, const float* Bm, int ldb, int N, float* out, int* ns_out, int target, hipStream_t st,
            const int* live) {
    GemmArgs g = {};
    g.nseg = 1;
    g.live = live;
    g.seg[0] = {A, Bm, lda, ldb, K, nullptr};
    g.M = M; g.N = N; g.out = out; g.ldo = N;
    g.nsplit = M <= 64 ? gemm_pick_split(g, target, GEMM_NN) : gemm_pick_split_balanced(g, GEMM_NN, ns_out ? xfloats : ws_floats);
    float* slab = ns_out ? out : ws;
    if (g.nsplit > 1) {
        ICZ_REQUIRE(gemm_slab_floats(M, N, g.nsplit) <= (ns_out ? xfloats : ws_floats), "nic: slab buffer too small");
        g.out = slab;
    }
    ICZ_TRY(gemm_f32(GEMM_NN, g, st));
    if (ns_out) { *ns_out = g.nsplit; return ICZ_OK; }
    if (g.nsplit > 1) {
        const size_t MN = (size_t)M * N;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, ws, g.nsplit, MN, N, (const float*)nullptr, out);
    }
    return ICZ_OK;
}

int Nic::tn(const float* dY, int ldy, int M, const float* Xm, int ldx, int N, int K, float* out, int ldo, int accumulate, hipStream_t st) {
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {dY, Xm, ldy, ldx, K, nullptr};
    g.M = M; g.N = N; g.out = out; g.ldo = ldo; g.nsplit = 1; g.accumulate = accumulate;
    return gemm_f32(GEMM_TN, g, st);
}

int Nic::bptt(const icz_nic_params& G, float* dfeats, hipStream_t st) {
    const int B = cur_B, T = cur_T, H = dims.H, E = dims.E, V = dims.V;
    const int TB = T * B;
    const size_t sH = (size_t)B * H;
    // predict layer over all time steps
    ICZ_TRY(nn(tlogit, Vp, TB, Vp, w_pred, H, H, dHd, nullptr, TARGET_WGS, st));
    ICZ_TRY(tn(tlogit, Vp, Vp, thd, H, H, TB, dWp, H, 0, st));
    ICZ_TRY(colsum(tlogit, TB, V, Vp, G.predict_b, st));
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(cdiv(V, 4)), dim3(256), 0, st, dWp, H, P.predict_v, P.predict_g, n_pred, G.predict_v,
                       G.predict_g, V, H);
    const bool ragged = rows_t[T - 1] < B;
    if (ragged) ICZ_CHECK_HIP(hipMemsetAsync(dG, 0, sizeof(float) * (size_t)(TB + B) * 4 * H, st));
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    int cur = 0, nsx = 1, bnext = 0;
    for (int t = T - 1; t >= -1; --t) {       // t = -1: the image step
        const int bt = t >= 0 ? rows_t[t] : B;
        const size_t slot = (size_t)(t + 1) * B;      // gate / dG slot; state-after slot = slot + B, state-before slot = slot
        LstmBwdArgs a = {};
        a.dh_a = bnext ? X : nullptr; a.ns_a = nsx; a.lda_a = H; a.rows_a = bnext;
        a.dhdrop = t >= 0 ? dHd + (size_t)t * B * H : nullptr;
        a.dc_in = bnext ? dcb[cur] : nullptr; a.dc_in_rows = bnext;
        a.gates = tg + slot * 4 * H;
        a.c_prev = tc + slot * H; a.c_cur = tc + (slot + B) * H;
        a.dgates = dG + slot * 4 * H; a.dc_prev = dcb[cur ^ 1];
        a.rows = bt; a.H = H;
        // backward of a sampled rollout: a step behind the reference's break (NIC_Model.py:150) never ran -- zero d gates, no carry
        const bool eo = bptt_early_out && early_out;
        const int* const live = (eo && t > 0) ? nunf + (t - 1) : nullptr;
        a.live = live; a.carry_live = (eo && t >= 0 && t + 1 < T) ? nunf + t : nullptr;
        DropCfg d = t >= 0 ? nic_drop(d_seed, cur_train, rng.out_mask, sH, t) : off;
        hipLaunchKernelGGL(lstm_bwd_point_kernel, dim3(cdiv(H, 256), bt), dim3(256), 0, st, a, d);
        if (t >= 0) ICZ_TRY(nn(dG + slot * 4 * H, 4 * H, bt, 4 * H, P.w_hh, H, H, X, &nsx, STEP_WGS, st, live));   // d h_{t-1}
        bnext = bt;
        cur ^= 1;
    }
    // embedding gradient, image-feature gradient
    ICZ_TRY(nn(dG + (size_t)B * 4 * H, 4 * H, TB, 4 * H, P.w_ih, E, E, dEmb, nullptr, TARGET_WGS, st));
    ICZ_CHECK_HIP(embed_grad_launch(st, tok, TB, dEmb, 1, (size_t)0, temb, 1.0f, E, G.embed_weight, V, 0));
    if (dfeats) ICZ_TRY(nn(dG, 4 * H, B, 4 * H, P.w_ih, E, E, dfeats, nullptr, TARGET_WGS, st));
    // weight gradients
    ICZ_TRY(tn(dG + (size_t)B * 4 * H, 4 * H, 4 * H, temb, E, E, TB, G.w_ih, E, 0, st));
    ICZ_TRY(tn(dG, 4 * H, 4 * H, cur_feats, E, E, B, G.w_ih, E, 1, st));
    ICZ_TRY(tn(dG, 4 * H, 4 * H, th, H, H, TB + B, G.w_hh, H, 0, st));
    ICZ_TRY(colsum(dG, TB + B, 4 * H, 4 * H, G.b_ih, st));
    ICZ_CHECK_HIP(hipMemcpyAsync(G.b_hh, G.b_ih, sizeof(float) * 4 * H, hipMemcpyDeviceToDevice, st));
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Nic::beam_search(const float* feats, int n_img, int k, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seqs_out && lens_out, "nic beam: null argument");
    ICZ_REQUIRE(k >= 1 && k <= BEAM_MAX_K, "nic beam: beam size %d out of range 1..%d", k, BEAM_MAX_K);
    ICZ_REQUIRE(n_img > 0 && (long)n_img * k <= dims.max_rows, "nic beam: %d images x %d beams exceed row capacity %d", n_img, k, dims.max_rows);
    ICZ_REQUIRE(max_steps >= 1 && max_steps <= 256, "nic beam: max_steps out of range");
    ICZ_REQUIRE(fresh, "nic: call icz_nic_refresh_weights after binding/updating parameters");
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
        ICZ_TRY(alloc((void**)&bm.cand_val, sizeof(float) * R_ * BEAM_MAX_K));
        ICZ_TRY(alloc((void**)&bm.cand_idx, sizeof(int) * R_ * BEAM_MAX_K));
        ICZ_TRY(alloc((void**)&bm.feat_rows, sizeof(float) * R_ * dims.E));
        ICZ_CHECK_HIP(hipHostMalloc((void**)&bm.n_live_host, sizeof(int) * 4, 0));
        ICZ_CHECK_HIP(hipDeviceSynchronize());      // alloc() zero-fills on the NULL stream (see ensure_train)
        bm.cap_rows = (int)R_;
        bm.cap_L = (int)L_;
    }
    ICZ_CHECK_HIP(hipMemsetAsync(bm.n_live, 0, sizeof(int) * 260, st));
    ICZ_CHECK_HIP(hipMemsetAsync(bm.run, 0, sizeof(float) * rows, st));
    hipLaunchKernelGGL(beam_init_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, n_img, k, L, bm.n_act, bm.seqs[0], bm.img_of_row, it,
                       bm.has_complete, bm.best_score);
    // every beam row starts from the image step of its image (features.expand(k, ...), NIC_Model.py:164)
    hipLaunchKernelGGL(beam_expand_rows_kernel, dim3(cdiv(dims.E, 1024), rows), dim3(256), 0, st, feats, bm.img_of_row, dims.E, bm.feat_rows);
    ICZ_TRY(image_step(bm.feat_rows, rows, h[0], c[0], nullptr, st));
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    int sb = 0, steps_done = 0;
    for (int step = 1; step <= max_steps; ++step) {
        ICZ_TRY(token_step(rows, it, false, h[0], c[0], h[1], c[1], emb, nullptr, hdrop, logits, off, st));
        BeamArgs a = {logits, dims.V, Vp, k, step, L, bm.n_act, bm.run, bm.seqs[sb], bm.seqs[sb ^ 1], bm.src_row, it,
                      bm.best_score, bm.best_len, bm.best_seq, bm.has_complete, bm.n_live + step};
        launch_beam_rowtopk(st, rows, a.logits, a.V, a.ldl, a.k, a.step, (const int*)bm.n_act, (const float*)bm.run, bm.cand_val, bm.cand_idx);
        hipLaunchKernelGGL(beam_merge_kernel, dim3(n_img), dim3(64), 0, st, a, (const float*)bm.cand_val, (const int*)bm.cand_idx);
        hipLaunchKernelGGL(beam_gather_kernel, dim3(cdiv(H, 1024), rows), dim3(256), 0, st, bm.src_row, H, h[1], c[1], h[1], c[1], h[0], c[0], h[0], c[0], 1);
        sb ^= 1;
        steps_done = step;
        if (step >= 6 && (step % 3) == 0 && step < max_steps) {
            ICZ_CHECK_HIP(hipMemcpyAsync(bm.n_live_host, bm.n_live + step, sizeof(int), hipMemcpyDeviceToHost, st));
            ICZ_CHECK_HIP(hipStreamSynchronize(st));
            if (bm.n_live_host[0] == 0) break;
        }
    }
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(n_img), dim3(64), 0, st, k, L, steps_done, bm.n_act, bm.run, bm.seqs[sb], bm.has_complete,
                       bm.best_len, bm.best_seq, seqs_out, lens_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

// ================================================================================================
using namespace icz;
extern "C" {

int icz_nic_create(const icz_nic_dims* dims, icz_nic_t** out) {
    ICZ_REQUIRE(dims && out, "icz_nic_create: null argument");
    Nic* n = new Nic();
    int s = n->init(*dims);
    if (s != ICZ_OK) { delete n; return s; }
    *out = reinterpret_cast<icz_nic_t*>(n);
    return ICZ_OK;
}
int icz_nic_destroy(icz_nic_t* h) { delete reinterpret_cast<Nic*>(h); return ICZ_OK; }
int icz_nic_bind_params(icz_nic_t* h, const icz_nic_params* p) {
    ICZ_REQUIRE(h && p, "icz_nic_bind_params: null argument");
    const float* const* q = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(icz_nic_params) / sizeof(float*); ++i) {
        ICZ_REQUIRE(q[i] != nullptr, "icz_nic_bind_params: parameter pointer %zu is null", i);
        ICZ_REQUIRE(((uintptr_t)q[i] & 15) == 0, "icz_nic_bind_params: parameter %zu not 16-byte aligned", i);
    }
    Nic* n = reinterpret_cast<Nic*>(h);
    n->P = *p; n->bound = true; n->fresh = false;
    return ICZ_OK;
}
int icz_nic_refresh_weights(icz_nic_t* h, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->refresh((hipStream_t)stream);
}
int icz_nic_greedy(icz_nic_t* h, const float* features, int32_t B, int32_t max_len, int64_t* ids_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->greedy(features, B, max_len, ids_out, (hipStream_t)stream);
}
int icz_nic_sample(icz_nic_t* h, const float* features, int32_t B, int32_t max_len, const icz_rng* rng, int64_t* seq_out,
                   float* logprobs_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->sample(features, B, max_len, rng, seq_out, logprobs_out, (hipStream_t)stream);
}
int icz_nic_sample_backward(icz_nic_t* h, const float* reward, const icz_nic_params* grads, float* dfeatures_out, float* loss_out,
                            float* mask_sum_out, float mask_sum_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->sample_backward(reward, grads, dfeatures_out, loss_out, mask_sum_out, mask_sum_global, (hipStream_t)stream);
}
int icz_nic_set_scheduled_sampling(icz_nic_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms) {
    ICZ_REQUIRE(h, "null handle");
    ICZ_REQUIRE(ss_prob >= 0.f && ss_prob <= 1.f, "icz_nic_set_scheduled_sampling: ss_prob %g outside [0, 1]", (double)ss_prob);
    Nic* n = reinterpret_cast<Nic*>(h);
    n->ss_prob = ss_prob; n->ss_gate = gate_uniforms; n->ss_draw = draw_uniforms;
    return ICZ_OK;
}
int icz_nic_xe_forward(icz_nic_t* h, const float* features, const int64_t* captions, int32_t B, int32_t L, const int32_t* lengths_host,
                       const icz_rng* rng, int32_t train, float* packed_logits_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->xe_forward(features, captions, B, L, lengths_host, rng, train, packed_logits_out, (hipStream_t)stream);
}
int icz_nic_set_option(icz_nic_t* h, const char* name, int32_t value) {
    ICZ_REQUIRE(h && name, "icz_nic_set_option: null argument");
    if (strcmp(name, "early_out") == 0) { reinterpret_cast<Nic*>(h)->early_out = value != 0; return ICZ_OK; }
    set_error("icz_nic_set_option: unknown option '%s'", name);
    return ICZ_ERR_INVALID;
}
int icz_nic_set_norm_global(icz_nic_t* h, const float* norm_dev, void* stream) {
    ICZ_REQUIRE(h && norm_dev, "icz_nic_set_norm_global: null argument");
    ICZ_CHECK_HIP(hipMemcpyAsync(reinterpret_cast<Nic*>(h)->d_msum, norm_dev, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ICZ_OK;
}
int icz_nic_xe_backward(icz_nic_t* h, float smoothing, const icz_nic_params* grads, float* dfeatures_out, float* loss_out,
                        float n_tokens_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->xe_backward(smoothing, grads, dfeatures_out, loss_out, n_tokens_global, (hipStream_t)stream);
}
int icz_nic_beam_search(icz_nic_t* h, const float* features, int32_t n_img, int32_t beam, int32_t max_steps, float* seqs_out,
                        int32_t* lens_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Nic*>(h)->beam_search(features, n_img, beam, max_steps, seqs_out, lens_out, (hipStream_t)stream);
}

}  // extern "C"
