// BUTD decoder: handle, launch sequences and C ABI (include/icz.h).  gfx950 only.
#include <stdarg.h>

#include <vector>

#include "butd_impl.h"

namespace icz {

static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ------------------------------------------------------------------------------------------------
int Butd::alloc(void** p, size_t bytes) {
    ICZ_CHECK_HIP(hipMalloc(p, bytes ? bytes : 16));
    (alloc_train ? tallocs : allocs).push_back(*p);
    return ICZ_OK;
}

void Butd::clear_graphs() {
    for (auto& e : graphs) (void)hipGraphExecDestroy(e.exec);
    graphs.clear();
}

int Butd::init(const icz_butd_dims& d) {
    dims = d;
    ICZ_REQUIRE(d.R > 0 && d.R <= 64, "butd: R=%d must be in 1..64 (36 boxes / 49 grid cells)", d.R);
    ICZ_REQUIRE(d.D % 4 == 0 && d.H % 4 == 0 && d.E % 4 == 0 && d.A % 4 == 0, "butd: D,H,E,A must be multiples of 4");
    ICZ_REQUIRE(d.V > 3 && d.max_rows > 0 && d.max_len > 0, "butd: bad V/max_rows/max_len");
    const size_t rows = d.max_rows, H = d.H, D = d.D, E = d.E, A = d.A, V = d.V, R = d.R;
    ICZ_TRY(alloc((void**)&w_enc, sizeof(float) * A * D));
    ICZ_TRY(alloc((void**)&w_dec, sizeof(float) * A * H));
    ICZ_TRY(alloc((void**)&w_aff, sizeof(float) * A));
    const size_t Vp = pad_vocab(d.V);           // padded rows stay zero: the dgrad GEMM reads K = Vp rows
    ICZ_TRY(alloc((void**)&w_pred, sizeof(float) * Vp * H));
    ICZ_CHECK_HIP(hipMemset(w_pred, 0, sizeof(float) * Vp * H));
    ICZ_TRY(alloc((void**)&n_enc, sizeof(float) * A));
    ICZ_TRY(alloc((void**)&n_dec, sizeof(float) * A));
    ICZ_TRY(alloc((void**)&n_aff, sizeof(float) * 4));
    ICZ_TRY(alloc((void**)&n_pred, sizeof(float) * V));
    ICZ_TRY(alloc((void**)&mean, sizeof(float) * rows * D));
    ICZ_TRY(alloc((void**)&premean, sizeof(float) * rows * 4 * H));
    ICZ_TRY(alloc((void**)&enc_ctx, sizeof(float) * rows * R * A));
    for (int i = 0; i < 2; ++i) {
        ICZ_TRY(alloc((void**)&h1[i], sizeof(float) * rows * H));
        ICZ_TRY(alloc((void**)&c1[i], sizeof(float) * rows * H));
        ICZ_TRY(alloc((void**)&h2[i], sizeof(float) * rows * H));
        ICZ_TRY(alloc((void**)&c2[i], sizeof(float) * rows * H));
    }
    ICZ_TRY(alloc((void**)&emb, sizeof(float) * rows * E));
    ICZ_TRY(alloc((void**)&ctx, sizeof(float) * rows * D));
    ICZ_TRY(alloc((void**)&scores, sizeof(float) * rows * R));
    ICZ_TRY(alloc((void**)&alpha, sizeof(float) * rows * R));
    ICZ_TRY(alloc((void**)&h2drop, sizeof(float) * rows * H));
    ICZ_TRY(alloc((void**)&logits, sizeof(float) * rows * Vp));      // rows padded to Vp: 16-byte aligned rows
    ICZ_TRY(alloc((void**)&amax_val, sizeof(float) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&amax_idx, sizeof(int) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&it, sizeof(int64_t) * rows));
    ICZ_TRY(alloc((void**)&d_seed, 16));
    ICZ_TRY(alloc((void**)&d_msum_global, 16));
    ICZ_CHECK_HIP(hipMemset(d_seed, 0, 16));
    ICZ_CHECK_HIP(hipMemset(d_msum_global, 0, 16));
    size_t nmax = 4 * H;
    if (A > nmax) nmax = A;
    if (V > nmax) nmax = V;
    ws_floats = (size_t)TARGET_WGS * 4096 * 2 + rows * nmax;
    {   // the resident decoder-step GEMMs leave one slab per 256-deep k range: (2H + max(E, D)) / 256 slabs of rows x 4H at up to 128 rows
        const size_t kmax = 2 * H + (E > D ? E : D), r128 = rows < 128 ? rows : 128;
        const size_t need = (kmax / 256 + 1) * r128 * 4 * H;
        if (need > ws_floats) ws_floats = need;
    }
    ICZ_TRY(alloc((void**)&ws, sizeof(float) * ws_floats));
    // defaults of the options "early_out" / "merge_small" from the environment (A/B runs of whole programs; icz_butd_set_option overrides)
    ICZ_CHECK_HIP(hipDeviceSynchronize());      // the zero-fills above ran on the NULL stream; callers use non-blocking streams (see ensure_train)
    if (const char* e = getenv("ICZ_EARLY_OUT")) early_out = atoi(e) != 0;
    if (const char* e = getenv("ICZ_MERGE_SMALL")) { const int n = atoi(e); if (n >= 0 && n <= 32) merge_small = n; }
    return ICZ_OK;
}

Butd::~Butd() {
    if (side_st) (void)hipStreamDestroy(side_st);
    if (low_st) (void)hipStreamDestroy(low_st);
    if (ev_fork2) (void)hipEventDestroy(ev_fork2);
    if (ev_join2) (void)hipEventDestroy(ev_join2);
    if (ev_fork3) (void)hipEventDestroy(ev_fork3);
    if (ev_join3) (void)hipEventDestroy(ev_join3);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    clear_graphs();
    if (cap_st) (void)hipStreamDestroy(cap_st);
    if (bm.n_live_host) (void)hipHostFree(bm.n_live_host);
    for (void* p : tallocs) (void)hipFree(p);
    for (void* p : allocs) (void)hipFree(p);
}

// ------------------------------------------------------------------------------------------------
int Butd::refresh(hipStream_t st) {
    ICZ_REQUIRE(bound, "butd: parameters not bound");
    const int A = dims.A, D = dims.D, H = dims.H, V = dims.V;
    WeightNormTable wt = {};
    int nb = 0;
    auto add = [&](const float* v, const float* g, float* w, float* norm, int rows, int cols) {
        wt.j[wt.count++] = {v, g, w, norm, rows, cols, nb};
        nb += cdiv(rows, 4);
    };
    add(P.enc_att_v, P.enc_att_g, w_enc, n_enc, A, D);
    add(P.dec_att_v, P.dec_att_g, w_dec, n_dec, A, H);
    add(P.affine_v, P.affine_g, w_aff, n_aff, 1, A);
    add(P.predict_v, P.predict_g, w_pred, n_pred, V, H);
    hipLaunchKernelGGL(weight_norm_multi_kernel, dim3(nb), dim3(256), 0, st, wt);
    ICZ_CHECK_HIP(hipGetLastError());
    fresh = true;
    wt_fresh = false;       // the transposed copies (117 MB) are rebuilt by the first backward pass that follows, outside its captured
    return ICZ_OK;          // graph (sample_backward / xe_backward*): evaluation loops refresh per batch and never read them
}

// the transposed copies serve 33..64-row dgrad steps through gemm_resident_x3: K = 4H in whole k ranges, N = D + H / H wide enough
bool Butd::wt_possible() const {
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {w_enc, w_enc, 4 * dims.H, 4 * dims.H, 4 * dims.H, nullptr};     // placeholders: only the shape is looked at
    g.M = 64; g.N = dims.D + dims.H;
    GemmArgs p = g, q = g;
    p.N = q.N = dims.H;
    p.nseg = 2; p.seg[1] = p.seg[0];
    return dims.H % 64 == 0 && dims.D % 64 == 0 && gemm_resident_x3_fits(g) && gemm_resident_x3_pair_fits(p, q);
}

int Butd::refresh_transposes(hipStream_t st) {
    ICZ_REQUIRE(bound && wt_lm_ih, "butd: no transposed weight buffers");
    const int H = dims.H, D = dims.D, E = dims.E;
    TransposeTable tt = {};
    int nb = 0;
    auto add = [&](const float* src, int ld, int cols, float* dst) {
        tt.j[tt.count++] = {src, ld, 4 * H, cols, dst, nb};
        nb += (4 * H / 64) * (cols / 64);
    };
    add(P.lm_w_ih, D + H, D + H, wt_lm_ih);
    add(P.lm_w_hh, H, H, wt_lm_hh);
    add(P.td_w_ih, H + D + E, H, wt_td_ih_h2);          // the h2 columns only: mean features and embedding have no recurrent gradient
    add(P.td_w_hh, H, H, wt_td_hh);
    hipLaunchKernelGGL(transpose_multi_kernel, dim3(nb), dim3(256), 0, st, tt);
    ICZ_CHECK_HIP(hipGetLastError());
    wt_fresh = true;
    return ICZ_OK;
}

// generic "y = x W^T" with automatic split-K into the workspace; returns the split used (1 = direct into out)
int Butd::gemm_nt(GemmArgs& g, int* nsplit_out, hipStream_t st) {
    g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, TARGET_WGS), ws_floats);
    if (g.nsplit > 1) {
        ICZ_REQUIRE(gemm_slab_floats(g.M, g.N, g.nsplit) <= ws_floats, "butd: workspace too small for %dx%dx%d slabs", g.nsplit, g.M, g.N);
        g.out = ws;
        g.bias = nullptr;
    }
    *nsplit_out = g.nsplit;
    return gemm_f32(GEMM_NT, g, st);
}

// Once per batch of images (K0 in SURVEY.md 2.3): mean features, the time-invariant part of the TD-LSTM gates
// (mean . W_ih[:, H:H+D]^T) and enc_att(feats) -- the reference recomputes the latter every step (:57).
int Butd::prologue(const float* feats, int n_img, hipStream_t st) {
    const int R = dims.R, D = dims.D, H = dims.H, E = dims.E, A = dims.A;
    ICZ_REQUIRE(fresh, "butd: call icz_butd_refresh_weights after binding/updating parameters");
    ICZ_REQUIRE(n_img > 0 && n_img <= dims.max_rows, "butd: %d images exceed capacity %d", n_img, dims.max_rows);
    hipLaunchKernelGGL(mean_feats_kernel, dim3(cdiv(D, 1024), n_img), dim3(256), 0, st, feats, mean, R, D);
    {   // premean = mean . W_ih[:, H:H+D]^T   [n_img, 4H]
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {mean, P.td_w_ih + H, D, H + D + E, D, nullptr};
        g.M = n_img; g.N = 4 * H; g.out = premean; g.ldo = 4 * H;
        int ns;
        ICZ_TRY(gemm_nt(g, &ns, st));
        if (ns > 1) {
            size_t MN = (size_t)n_img * 4 * H;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, ws, ns, MN, 4 * H, (const float*)nullptr, premean);
        }
    }
    {   // enc_ctx = feats . w_enc^T + b_enc   [n_img*R, A]
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {feats, w_enc, D, D, D, nullptr};
        g.M = n_img * R; g.N = A; g.out = enc_ctx; g.ldo = A; g.bias = P.enc_att_b;
        int ns;
        const float* bias = g.bias;
        ICZ_TRY(gemm_nt(g, &ns, st));
        if (ns > 1) {
            size_t MN = (size_t)n_img * R * A;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, ws, ns, MN, A, bias, enc_ctx);
        }
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// One decoder step (:172-182) for `rows` decoder rows.  State is read from s.*_in and written to s.*_out.
int Butd::step(const StepIO& s, hipStream_t st) {
    const int R = dims.R, D = dims.D, H = dims.H, E = dims.E, A = dims.A, V = dims.V;
    const int Vp = pad_vocab(V);
    const int rows = s.rows;
    float* const ws = s.ws_alt ? s.ws_alt : this->ws;
    float* const scores = s.scores_alt ? s.scores_alt : this->scores;
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    // embedding -> relu -> dropout (greedy decode gets it from the previous step's fused argmax epilogue)
    if (!s.emb_ready)
        hipLaunchKernelGGL(embed_kernel, dim3(cdiv(E, 1024), rows), dim3(256), 0, st, P.embed_weight, s.it, s.emb_out ? s.emb_out : emb, rows, E, s.drop_emb);
    const float* embp = s.emb_out ? s.emb_out : emb;
    int ns;
    {   // TD-attention LSTM: [h2, mean, emb] W_ih^T + h1 W_hh^T  (mean part hoisted into premean)
        GemmArgs g = {};
        g.nseg = 3;
        g.seg[0] = {s.h2_in, P.td_w_ih, H, H + D + E, H, nullptr};
        g.seg[1] = {embp, P.td_w_ih + H + D, E, H + D + E, E, nullptr};
        g.seg[2] = {s.h1_in, P.td_w_hh, H, H, H, nullptr};
        g.M = rows; g.N = 4 * H; g.out = ws; g.ldo = 4 * H;
        g.live = s.live;
        const size_t ws_cap = s.ws_alt ? (tb.xfloats < ws_floats ? tb.xfloats : ws_floats) : ws_floats;      // ws_alt = tb.X[0]
        g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, STEP_WGS), ws_cap);
        ns = g.nsplit;
        ICZ_REQUIRE(gemm_slab_floats(rows, 4 * H, ns) <= ws_cap && (size_t)rows * 4 * H <= ws_cap, "butd: workspace too small");
        if (ns == 1) { g.out = ws; }
        ICZ_TRY(gemm_f32(GEMM_NT, g, st));
        LstmPointArgs a = {ws, ns, premean, s.img_of_row, P.td_b_ih, P.td_b_hh, s.c1_in, s.h1_out, s.c1_out, s.gates_td_out, nullptr, rows, H, s.live};
        kprof_mark(KP_LSTM_POINT, true, st);
        launch_lstm_point(a, off, st);
        kprof_mark(KP_LSTM_POINT, false, st);
    }
    {   // attention
        kprof_mark(KP_ATTENTION, true, st);
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {s.h1_out, w_dec, H, H, H, nullptr};
        g.M = rows; g.N = A; g.out = ws; g.ldo = A;
        g.live = s.live;
        g.nsplit = gemm_pick_split(g, STEP_WGS);
        ns = g.nsplit;
        ICZ_TRY(gemm_f32(GEMM_NT, g, st));
        AttScoreArgs a = {enc_ctx, s.img_of_row, ws, ns, P.dec_att_b, w_aff, P.affine_b, s.dec_ctx_out, scores, rows, R, A, s.live};
        const int G = s.rows_per_img;
        // compare the accumulation order: the per-row kernel sums a region row in the lane order of three regions at a time, the
        // grouped one region by region -- identical per (row, region): a wave's lanes cover the same columns in the same order
        if (G > 1 && G <= ATT_CTX_MAX_G && rows % G == 0 && s.drop_att.mode == 0 && sizeof(float) * A * G <= 60 * 1024)
            hipLaunchKernelGGL(att_scores_group_kernel, dim3(rows / G, ATT_PARTS), dim3(256), sizeof(float) * A * G, st, a, G);
        else
            hipLaunchKernelGGL(att_scores_kernel, dim3(rows, ATT_PARTS), dim3(256), sizeof(float) * A, st, a, s.drop_att);
        if (G > 1 && G <= ATT_CTX_MAX_G && rows % G == 0 && !s.alpha_out2)
            hipLaunchKernelGGL(att_ctx_group_kernel, dim3(rows / G, cdiv(D, 512)), dim3(256), 0, st, s.feats, (const float*)scores,
                               s.alpha_out ? s.alpha_out : alpha, s.ctx_out ? s.ctx_out : ctx, R, D, G);
        else
            hipLaunchKernelGGL(att_ctx_kernel, dim3(rows, cdiv(D, 512)), dim3(256), 0, st, s.feats, s.img_of_row, (const float*)scores,
                               s.alpha_out ? s.alpha_out : alpha, s.alpha_out2, s.alpha2_stride, s.ctx_out ? s.ctx_out : ctx, R, D, s.live);
        kprof_mark(KP_ATTENTION, false, st);
    }
    const float* ctxp = s.ctx_out ? s.ctx_out : ctx;
    {   // language LSTM: [ctx, h1] W_ih^T + h2 W_hh^T
        GemmArgs g = {};
        g.nseg = 3;
        g.seg[0] = {ctxp, P.lm_w_ih, D, D + H, D, nullptr};
        g.seg[1] = {s.h1_out, P.lm_w_ih + D, H, D + H, H, nullptr};
        g.seg[2] = {s.h2_in, P.lm_w_hh, H, H, H, nullptr};
        g.M = rows; g.N = 4 * H; g.out = ws; g.ldo = 4 * H;
        g.live = s.live;
        const size_t ws_cap = s.ws_alt ? (tb.xfloats < ws_floats ? tb.xfloats : ws_floats) : ws_floats;
        g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, STEP_WGS), ws_cap);
        ns = g.nsplit;
        ICZ_REQUIRE(gemm_slab_floats(rows, 4 * H, ns) <= ws_cap && (size_t)rows * 4 * H <= ws_cap, "butd: workspace too small");
        ICZ_TRY(gemm_f32(GEMM_NT, g, st));
        LstmPointArgs a = {ws, ns, nullptr, nullptr, P.lm_b_ih, P.lm_b_hh, s.c2_in, s.h2_out, s.c2_out, s.gates_lm_out,
                           s.h2drop_out ? s.h2drop_out : h2drop, rows, H, s.live};
        launch_lstm_point(a, s.drop_out, st);
    }
    if (!s.skip_predict) {   // predict: logits = drop(h2) w_pred^T + b  (finished logits, or split-K slabs in the chain's workspace for a consumer that sums them)
        const size_t ws_cap = s.ws_alt ? (tb.xfloats < ws_floats ? tb.xfloats : ws_floats) : ws_floats;      // ws_alt = tb.X[0]
        ICZ_TRY(gemm_predict(s.h2drop_out ? s.h2drop_out : h2drop, H, w_pred, P.predict_b, rows, V, Vp, s.logits_out ? s.logits_out : logits,
                             s.logits_ld ? s.logits_ld : Vp, ws, ws_cap, s.pred_nsplit, st, s.live));
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Butd::zero_state(int rows, int which, hipStream_t st) {
    ZeroList z = {{h1[which], c1[which], h2[which], c2[which]}, 4};
    const size_t n = (size_t)rows * dims.H;
    hipLaunchKernelGGL(zero_bufs_kernel, dim3(cdiv((int)(n / 4), 256)), dim3(256), 0, st, z, n);
    return ICZ_OK;
}

int Butd::greedy(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st) {
    ICZ_REQUIRE(feats && ids_out && B > 0 && B <= dims.max_rows && max_len > 0, "butd greedy: bad arguments");
    ICZ_REQUIRE(fresh, "butd: call icz_butd_refresh_weights after binding/updating parameters");
    const std::vector<uintptr_t> key = {1, (uintptr_t)feats, (uintptr_t)B, (uintptr_t)max_len, (uintptr_t)ids_out, (uintptr_t)alphas_out};
    return run_cached(key, st, [&](hipStream_t s) { return greedy_impl(feats, B, max_len, ids_out, alphas_out, s); });
}

int Butd::greedy_impl(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st) {
    ICZ_TRY(prologue(feats, B, st));
    return greedy_chain(feats, B, max_len, ids_out, alphas_out, st);
}

// the decode loop after the per-image prologue
// scst = true (the baseline of an SCST step, rollouts_impl): the chain keeps count of the rows that have not emitted <end> yet and,
// once there is none, the kernels of the remaining steps return at entry (ids = 0 there).  The reference's greedy loop has no break
// (BUTD_Model.py:171-186), but nothing behind a row's <end> reaches the reward (Utils.py:354 cuts there): the SCST step's results
// are the same.  icz_butd_greedy (evaluation, `sampler`) never does this: its ids are the reference's in every column.
int Butd::greedy_chain(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st, bool scst) {
    ICZ_TRY(zero_state(B, 0, st));
    int* const gn = (scst && early_out && tb.gnunf && tb.T >= max_len && tb.B >= B) ? tb.gnunf : nullptr;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(cdiv(B > max_len ? B : max_len, 256)), dim3(256), 0, st, it, B, gn, max_len);   // <sta>
    int cur = 0;
    const int Vp = pad_vocab(dims.V);
    bool track = false;
    for (int t = 0; t < max_len; ++t) {
        StepIO s = {};
        s.rows = B; s.feats = feats; s.it = it;
        s.emb_ready = t > 0;           // produced by the previous step's embed_argmax_kernel
        s.h1_in = h1[cur]; s.c1_in = c1[cur]; s.h2_in = h2[cur]; s.c2_in = c2[cur];
        s.h1_out = h1[cur ^ 1]; s.c1_out = c1[cur ^ 1]; s.h2_out = h2[cur ^ 1]; s.c2_out = c2[cur ^ 1];
        if (alphas_out) { s.alpha_out2 = alphas_out + (size_t)t * dims.R; s.alpha2_stride = max_len * dims.R; }
        int pns = 1;
        s.pred_nsplit = &pns;
        if (track && t > 0) s.live = gn + (t - 1);
        ICZ_TRY(step(s, st));
        if (t == 0) track = gn && pns > 1;     // the one-launch select below keeps the count (the two-kernel argmax of <= 32 rows does not)
        kprof_mark(KP_GREEDY_SELECT, true, st);
        if (pns > 1)         // 33 - 64 rows: the slabs of the vocabulary projection -> token + next embedding in one launch
            hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(1024), 0, st, (const float*)ws, dims.V, Vp, pns, (size_t)B * Vp,
                               (const float*)P.predict_b, P.embed_weight, dims.E, emb, it, ids_out, max_len, t, 1,
                               track ? tb.gunf : (uint8_t*)nullptr, track ? gn : (int*)nullptr);
        else {
            hipLaunchKernelGGL(argmax_part_kernel, dim3(B, ARGMAX_PARTS), dim3(256), 0, st, logits, dims.V, Vp, ARGMAX_PARTS, amax_val, amax_idx);
            hipLaunchKernelGGL(embed_argmax_kernel, dim3(cdiv(dims.E, 1024), B), dim3(256), 0, st, amax_val, amax_idx, ARGMAX_PARTS,
                               P.embed_weight, dims.E, emb, it, ids_out, max_len, t);
        }
        kprof_mark(KP_GREEDY_SELECT, false, st);
        cur ^= 1;
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

// ================================================================================================
using namespace icz;

extern "C" {

const char* icz_last_error(void) { return icz::g_err; }
const char* icz_version(void) { return "libicz 0.1 (gfx950)"; }

int icz_butd_create(const icz_butd_dims* dims, icz_butd_t** out) {
    ICZ_REQUIRE(dims && out, "icz_butd_create: null argument");
    Butd* b = new Butd();
    int s = b->init(*dims);
    if (s != ICZ_OK) { delete b; return s; }
    *out = reinterpret_cast<icz_butd_t*>(b);
    return ICZ_OK;
}

int icz_butd_destroy(icz_butd_t* h) {
    delete reinterpret_cast<Butd*>(h);
    return ICZ_OK;
}

int icz_butd_bind_params(icz_butd_t* h, const icz_butd_params* p) {
    ICZ_REQUIRE(h && p, "icz_butd_bind_params: null argument");
    const float* const* q = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(icz_butd_params) / sizeof(float*); ++i) {
        ICZ_REQUIRE(q[i] != nullptr, "icz_butd_bind_params: parameter pointer %zu is null", i);
        ICZ_REQUIRE(((uintptr_t)q[i] & 15) == 0 || i == 16 || i == 17, "icz_butd_bind_params: parameter %zu not 16-byte aligned", i);
    }
    Butd* b = reinterpret_cast<Butd*>(h);
    // the captured graphs carry the old parameter pointers in their kernel arguments (embed_weight, the LSTM weights and
    // biases ...): a rebind to other tensors must not replay them
    if (b->bound && memcmp(&b->P, p, sizeof(*p)) != 0) {
        ICZ_CHECK_HIP(hipDeviceSynchronize());
        b->clear_graphs();
    }
    b->P = *p;
    b->bound = true;
    b->fresh = false;
    return ICZ_OK;
}

int icz_butd_set_option(icz_butd_t* h, const char* name, int32_t value) {
    ICZ_REQUIRE(h && name, "icz_butd_set_option: null argument");
    Butd* b = reinterpret_cast<Butd*>(h);
    if (strcmp(name, "graphs") == 0) { b->use_graphs = value != 0; return ICZ_OK; }
    if (strcmp(name, "concurrent") == 0) { b->concurrent = value != 0; return ICZ_OK; }
    if (strcmp(name, "early_out") == 0) { b->early_out = value != 0; return ICZ_OK; }
    if (strcmp(name, "small_nt") == 0) { b->small_nt = value != 0; return ICZ_OK; }
    if (strcmp(name, "merge_small") == 0) {
        ICZ_REQUIRE(value >= 0 && value <= 32, "icz_butd_set_option: merge_small %d outside 0..32", value);
        b->merge_small = value;
        return ICZ_OK;
    }
    set_error("icz_butd_set_option: unknown option '%s'", name);
    return ICZ_ERR_INVALID;
}

int icz_butd_set_mask_sum_global(icz_butd_t* h, const float* mask_sum_global_dev, void* stream) {
    ICZ_REQUIRE(h && mask_sum_global_dev, "icz_butd_set_mask_sum_global: null argument");
    Butd* b = reinterpret_cast<Butd*>(h);
    ICZ_CHECK_HIP(hipMemcpyAsync(b->d_msum_global, mask_sum_global_dev, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ICZ_OK;
}

int icz_butd_set_grad_callback(icz_butd_t* h, icz_grad_ready_cb cb, void* user) {
    ICZ_REQUIRE(h, "null handle");
    Butd* b = reinterpret_cast<Butd*>(h);
    b->grad_cb = cb; b->grad_cb_user = user;
    return ICZ_OK;
}

int icz_butd_refresh_weights(icz_butd_t* h, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->refresh((hipStream_t)stream);
}

int icz_butd_greedy(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, int64_t* ids_out,
                    float* alphas_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->greedy(feats, B, max_len, ids_out, alphas_out, (hipStream_t)stream);
}

int icz_butd_step(icz_butd_t* h, const float* feats, int32_t B, const int64_t* it, float* h1, float* c1,
                  float* h2, float* c2, float* ctx_out, float* alpha_out, float* logits_out, void* stream) {
    ICZ_REQUIRE(h && feats && it && h1 && c1 && h2 && c2 && logits_out, "icz_butd_step: null argument");
    Butd* b = reinterpret_cast<Butd*>(h);
    hipStream_t st = (hipStream_t)stream;
    ICZ_REQUIRE(B > 0 && B <= b->dims.max_rows, "icz_butd_step: B out of range");
    ICZ_TRY(b->prologue(feats, B, st));
    StepIO s = {};
    s.rows = B; s.feats = feats; s.it = it;
    s.h1_in = h1; s.c1_in = c1; s.h2_in = h2; s.c2_in = c2;
    // in-place update is safe: every consumer of *_in has been launched before the kernel that overwrites it
    // only if input/output buffers differ, so run into the handle's buffers and copy back.
    s.h1_out = b->h1[0]; s.c1_out = b->c1[0]; s.h2_out = b->h2[0]; s.c2_out = b->c2[0];
    s.ctx_out = ctx_out; s.alpha_out = alpha_out; s.logits_out = logits_out; s.logits_ld = b->dims.V;
    ICZ_TRY(b->step(s, st));
    const size_t n = sizeof(float) * B * b->dims.H;
    ICZ_CHECK_HIP(hipMemcpyAsync(h1, b->h1[0], n, hipMemcpyDeviceToDevice, st));
    ICZ_CHECK_HIP(hipMemcpyAsync(c1, b->c1[0], n, hipMemcpyDeviceToDevice, st));
    ICZ_CHECK_HIP(hipMemcpyAsync(h2, b->h2[0], n, hipMemcpyDeviceToDevice, st));
    ICZ_CHECK_HIP(hipMemcpyAsync(c2, b->c2[0], n, hipMemcpyDeviceToDevice, st));
    return ICZ_OK;
}

size_t icz_gemm_workspace_floats(int32_t M, int32_t N) {
    const size_t base = (size_t)Butd::TARGET_WGS * 4096 * 2 + (size_t)M * N;
    // 65..128 rows: the 128-row resident kernel leaves up to 32 slabs of M x N (one per 256-deep k range of K <= 8192)
    const size_t m128 = (M > 64 && M <= 128) ? (size_t)32 * M * N : 0;
    return base > m128 ? base : m128;
}

int icz_gemm_f32(int32_t layout, const float* X, int32_t ldx, const float* W, int32_t ldw, const float* bias,
                 float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t nsplit, float* workspace,
                 size_t workspace_floats, void* stream) {
    ICZ_REQUIRE(layout >= 0 && layout <= 2, "icz_gemm_f32: layout %d", layout);
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {X, W, ldx, ldw, K, nullptr};
    g.M = M; g.N = N; g.out = C; g.ldo = ldc; g.bias = bias;
    g.nsplit = nsplit > 0 ? nsplit : gemm_fit_split((GemmLayout)layout, g, gemm_pick_split(g, Butd::TARGET_WGS, (GemmLayout)layout), workspace_floats);
    g.nsplit = gemm_normalize_split((GemmLayout)layout, g, g.nsplit);   // no empty splits
    hipStream_t st = (hipStream_t)stream;
    if (g.nsplit > 1) {
        ICZ_REQUIRE(workspace, "icz_gemm_f32: split-K needs a workspace");
        ICZ_REQUIRE(gemm_slab_floats(M, N, g.nsplit) <= workspace_floats, "icz_gemm_f32: workspace of %zu floats too small for %d slabs of %dx%d", workspace_floats, g.nsplit, M, N);
        ICZ_REQUIRE(ldc == N, "icz_gemm_f32: split-K path needs ldc == N");
        g.out = workspace; g.bias = nullptr;
        ICZ_TRY(gemm_f32((GemmLayout)layout, g, st));
        size_t MN = (size_t)M * N;
        ICZ_REQUIRE(MN % 4 == 0, "icz_gemm_f32: M*N must be a multiple of 4 for the split-K reduce");
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, workspace, g.nsplit, MN, N, bias, C);
        ICZ_CHECK_HIP(hipGetLastError());
        return ICZ_OK;
    }
    return gemm_f32((GemmLayout)layout, g, st);
}

int icz_gemm_set_big_cfg(int32_t cfg) {
    ICZ_REQUIRE(cfg >= -2 && cfg <= 5, "icz_gemm_set_big_cfg: %d", cfg);
    gemm_set_big_cfg(cfg);
    return ICZ_OK;
}

int icz_gemm_big_cfg_for(int32_t layout, int32_t M, int32_t N, int32_t K, int32_t nsplit) {
    if (layout < 0 || layout > 2 || M <= 0 || N <= 0 || K <= 0) return -1;
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0].K = K;
    g.M = M; g.N = N; g.nsplit = nsplit > 0 ? nsplit : 1;
    return gemm_big_cfg((GemmLayout)layout, g);
}

int icz_gemm_tn_grouped(const float* dY, int32_t ldy, int32_t M, int32_t K, int32_t ngroups, const float* const* X, const int32_t* ldx,
                        const int32_t* cols, float* const* out, const int32_t* ldo, const int32_t* rows_live, void* stream) {
    ICZ_REQUIRE(ngroups >= 1 && ngroups <= GEMM_MAX_COLGROUPS && X && ldx && cols && out && ldo, "icz_gemm_tn_grouped: bad arguments");
    GemmColGroup g[GEMM_MAX_COLGROUPS];
    for (int j = 0; j < ngroups; ++j) g[j] = {X[j], ldx[j], cols[j], out[j], ldo[j]};
    ICZ_REQUIRE(gemm_tn_grouped_fits(M, K, g, ngroups), "icz_gemm_tn_grouped: shape not taken (M %d, K %d)", M, K);
    return gemm_tn_grouped(dY, ldy, M, K, g, ngroups, rows_live, (hipStream_t)stream);
}

}  // extern "C"
