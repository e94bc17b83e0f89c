// BUTD training paths: sample_rl rollout + REINFORCE backward, teacher-forced XE forward + backward.
// Both share one BPTT (Butd::bptt): per-step dgrad GEMMs (NN) and pointwise kernels run sequentially in
// reverse time; every weight gradient is one batched GEMM (TN) over all (time, row) pairs at the end.
#include <math.h>

#include "butd_impl.h"

namespace icz {

static inline int round4(int x) { return pad_vocab(x); }   // (historic name) padded vocabulary size

__global__ void sample_init_kernel(uint8_t* unf, int* nunf, int64_t* tok, int B, int T) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { unf[i] = 1; tok[i] = 1; }
    if (i < T) nunf[i] = 0;
}

int Butd::ensure_train(int B, int T) {
    if (tb.B >= B && tb.T >= T) return ICZ_OK;
    ICZ_REQUIRE(B <= 2 * dims.max_rows, "butd: batch %d exceeds capacity %d", B, dims.max_rows);      // (2 x: the merged chain of a small SCST batch)
    ICZ_REQUIRE(T <= XE_MAX_T, "butd: %d steps exceed the limit of %d", T, XE_MAX_T);
    // Sized by what the batches ask for, not by the handle's row capacity (which also covers images x beam rows): the
    // reference never truncates captions (Datasets.py:47-51), so the step count of an XE batch is only known when it
    // arrives.  Growing re-allocates everything: the rollout / XE activations kept for a pending backward are lost (callers
    // run forward and backward of one batch back to back) and every captured graph holds stale addresses.
    if (tb.B > B) B = tb.B;
    if (tb.T > T) T = tb.T;
    if (dims.max_len > T) T = dims.max_len;
    if (!tallocs.empty()) {
        ICZ_CHECK_HIP(hipDeviceSynchronize());
        clear_graphs();
        for (void* p : tallocs) (void)hipFree(p);
        tallocs.clear();
        tb = TrainBuf();
        mode = 0;
    }
    if (!wt_lm_ih && wt_possible()) {      // permanent (not re-allocated when the training buffers grow)
        const size_t G4 = 4 * (size_t)dims.H;
        ICZ_TRY(alloc((void**)&wt_lm_ih, sizeof(float) * G4 * (dims.D + dims.H)));
        ICZ_TRY(alloc((void**)&wt_lm_hh, sizeof(float) * G4 * dims.H));
        ICZ_TRY(alloc((void**)&wt_td_ih_h2, sizeof(float) * G4 * dims.H));
        ICZ_TRY(alloc((void**)&wt_td_hh, sizeof(float) * G4 * dims.H));
        wt_fresh = false;
    }
    struct Scope { bool& f; Scope(bool& x) : f(x) { f = true; } ~Scope() { f = false; } } scope(alloc_train);
    const size_t H = dims.H, D = dims.D, E = dims.E, A = dims.A, R = dims.R, V = dims.V;
    const size_t Vp = round4(dims.V);
    const size_t TB = (size_t)T * B;
    auto zalloc = [&](void** p, size_t bytes) -> int {
        ICZ_TRY(alloc(p, bytes));
        ICZ_CHECK_HIP(hipMemset(*p, 0, bytes ? bytes : 16));
        return ICZ_OK;
    };
    ICZ_TRY(zalloc((void**)&tb.tok, sizeof(int64_t) * (TB + B)));
    ICZ_TRY(zalloc((void**)&tb.emb, sizeof(float) * TB * E));
    ICZ_TRY(zalloc((void**)&tb.h1, sizeof(float) * (TB + B) * H));
    ICZ_TRY(zalloc((void**)&tb.c1, sizeof(float) * (TB + B) * H));
    ICZ_TRY(zalloc((void**)&tb.h2, sizeof(float) * (TB + B) * H));
    ICZ_TRY(zalloc((void**)&tb.c2, sizeof(float) * (TB + B) * H));
    ICZ_TRY(zalloc((void**)&tb.gtd, sizeof(float) * TB * 4 * H));
    ICZ_TRY(zalloc((void**)&tb.glm, sizeof(float) * TB * 4 * H));
    ICZ_TRY(zalloc((void**)&tb.dec, sizeof(float) * TB * A));
    ICZ_TRY(zalloc((void**)&tb.alpha, sizeof(float) * TB * R));
    ICZ_TRY(zalloc((void**)&tb.ctx, sizeof(float) * TB * D));
    ICZ_TRY(zalloc((void**)&tb.h2d, sizeof(float) * TB * H));
    ICZ_TRY(zalloc((void**)&tb.logit, sizeof(float) * TB * Vp));
    ICZ_TRY(zalloc((void**)&tb.draw, sizeof(int32_t) * TB));
    ICZ_TRY(zalloc((void**)&tb.lse, sizeof(float) * TB));
    ICZ_TRY(zalloc((void**)&tb.unf, B));
    ICZ_TRY(zalloc((void**)&tb.nunf, sizeof(int) * T));
    ICZ_TRY(zalloc((void**)&tb.gunf, B));
    ICZ_TRY(zalloc((void**)&tb.gnunf, sizeof(int) * T));
    ICZ_TRY(zalloc((void**)&tb.live_rows, 16));
    ICZ_TRY(zalloc((void**)&tb.nany, sizeof(int) * T));
    ICZ_TRY(zalloc((void**)&tb.img2, sizeof(int32_t) * B));
    ICZ_TRY(zalloc((void**)&tb.coef, sizeof(float) * TB));
    ICZ_TRY(zalloc((void**)&tb.loss_rows, sizeof(float) * TB));
    ICZ_TRY(zalloc((void**)&tb.dGtd, sizeof(float) * TB * 4 * H));
    ICZ_TRY(zalloc((void**)&tb.dGlm, sizeof(float) * TB * 4 * H));
    ICZ_TRY(zalloc((void**)&tb.dDec, sizeof(float) * TB * A));
    ICZ_TRY(zalloc((void**)&tb.dEmb, sizeof(float) * TB * E));
    ICZ_TRY(zalloc((void**)&tb.dH2d, sizeof(float) * TB * H));
    ICZ_TRY(zalloc((void**)&tb.dEnc, sizeof(float) * (size_t)B * R * A));
    ICZ_TRY(zalloc((void**)&tb.dwaff, sizeof(float) * (size_t)B * ATT_PARTS * A));
    ICZ_TRY(zalloc((void**)&tb.dalpha, sizeof(float) * (size_t)B * R * cdiv((int)D, DALPHA_COLS)));
    ICZ_TRY(zalloc((void**)&tb.dS, sizeof(float) * TB * R));
    ICZ_TRY(zalloc((void**)&tb.dGsum, sizeof(float) * (size_t)B * 4 * H));
    for (int i = 0; i < 2; ++i) {
        ICZ_TRY(zalloc((void**)&tb.dc1[i], sizeof(float) * (size_t)B * H));
        ICZ_TRY(zalloc((void**)&tb.dc2[i], sizeof(float) * (size_t)B * H));
    }
    tb.xfloats = (size_t)TARGET_WGS * 4096 * 2 + (size_t)B * (D + H);
    {   // X[0] is the sampled chain's slab workspace (train_step): same rule as Butd::init's ws_floats
        const size_t kmax = 2 * H + (E > D ? E : D), r128 = B < 128 ? B : 128;
        const size_t need = (kmax / 256 + 1) * r128 * 4 * H;
        if (need > tb.xfloats) tb.xfloats = need;
    }
    for (int i = 0; i < 4; ++i) ICZ_TRY(zalloc((void**)&tb.X[i], sizeof(float) * tb.xfloats));
    ICZ_TRY(zalloc((void**)&tb.dWp, sizeof(float) * Vp * H));
    ICZ_TRY(zalloc((void**)&tb.dWenc, sizeof(float) * A * D));
    ICZ_TRY(zalloc((void**)&tb.dWdec, sizeof(float) * A * H));
    {   // slabs of the two attention weight gradients (A x H over T B rows, A x D over B R rows), one buffer: they run one after the other
        const size_t n1 = (size_t)gemm_tn_split_pick((int)A, (int)H, (int)TB) * A * H, n2 = (size_t)gemm_tn_split_pick((int)A, (int)D, (int)(B * R)) * A * D;
        tb.wslab_floats = n1 > n2 ? n1 : n2;
        if (tb.wslab_floats > A * (H > D ? H : D)) ICZ_TRY(zalloc((void**)&tb.wslab, sizeof(float) * tb.wslab_floats));
        else tb.wslab_floats = 0;
    }
    ICZ_TRY(zalloc((void**)&tb.dWaff, sizeof(float) * A));
    ICZ_TRY(zalloc((void**)&tb.scalars, sizeof(float) * 16));
    ICZ_TRY(zalloc((void**)&tb.scalars_i, sizeof(int) * 2 * T));
    tb.scalars_i_cap = 2 * T;
    (void)V;
    // The hipMemset calls above run on the NULL stream; callers enqueue on NON-BLOCKING streams (torch's), which are not ordered behind
    // it: without this, a kernel of the first call after a (re)allocation could run BEFORE the zero-fill of its buffer and then be
    // wiped by it (round 5: sample_init_kernel's unfinished flags, seen as an all-zero rollout in 1 of 3 five-rank runs).
    ICZ_CHECK_HIP(hipDeviceSynchronize());
    tb.B = B;
    tb.T = T;
    return ICZ_OK;
}

static DropCfg make_drop(const uint64_t* seed_p, bool train, const uint8_t* base, size_t per_step, uint32_t stream, int t, int row0 = 0) {
    DropCfg d = {0, nullptr, seed_p, stream, (uint32_t)t, row0};
    if (!train) return d;
    if (base) { d.mode = 1; d.mask = base + per_step * t; }
    else d.mode = 2;
    return d;
}

// ------------------------------------------------------------------------------------------------
// forward step into the saved-activation slots of time t (rows = active rows; slot stride = Bs rows)
// row0 > 0: the merged chain -- rows [0, row0) are evaluation-mode rows, the dropout arrays / Philox indices belong to the Bs - row0 rows behind
int Butd::train_step(const float* feats, int rows, int Bs, int t, bool train, hipStream_t st, bool emb_ready, int* pred_nsplit, bool skip_predict,
                     const int* live, int row0) {
    const size_t H = dims.H, D = dims.D, E = dims.E, A = dims.A, R = dims.R;
    const size_t Vp = round4(dims.V);
    const size_t slot = (size_t)t * Bs;
    StepIO s = {};
    s.emb_ready = emb_ready;         // written by the previous step's sample_select_kernel
    s.rows = rows; s.feats = feats; s.it = tb.tok + slot;
    s.h1_in = tb.h1 + slot * H; s.c1_in = tb.c1 + slot * H; s.h2_in = tb.h2 + slot * H; s.c2_in = tb.c2 + slot * H;
    s.h1_out = tb.h1 + (slot + Bs) * H; s.c1_out = tb.c1 + (slot + Bs) * H;
    s.h2_out = tb.h2 + (slot + Bs) * H; s.c2_out = tb.c2 + (slot + Bs) * H;
    s.emb_out = tb.emb + slot * E;
    s.gates_td_out = tb.gtd + slot * 4 * H; s.gates_lm_out = tb.glm + slot * 4 * H;
    s.dec_ctx_out = tb.dec + slot * A; s.alpha_out = tb.alpha + slot * R; s.ctx_out = tb.ctx + slot * D;
    s.h2drop_out = tb.h2d + slot * H; s.logits_out = tb.logit + slot * Vp; s.logits_ld = (int)Vp;
    // own slab workspace / score scratch (backward-only buffers, idle during the forward): the sampled rollout can
    // then run concurrently with the greedy rollout, which uses the handle's
    s.ws_alt = tb.X[0]; s.scores_alt = tb.dalpha;
    const size_t Bd = (size_t)(Bs - row0);            // rows the dropout arrays are laid out for
    s.drop_emb = make_drop(d_seed, train, rng.emb_mask, Bd * E, RNG_EMB, t, row0);
    s.drop_att = make_drop(d_seed, train, rng.att_mask, Bd * R * A, RNG_ATT, t, row0);
    s.drop_out = make_drop(d_seed, train, rng.out_mask, Bd * H, RNG_OUT, t, row0);
    if (row0 > 0) s.img_of_row = tb.img2;
    s.pred_nsplit = pred_nsplit;
    s.skip_predict = skip_predict;
    s.live = live;
    return step(s, st);
}

int Butd::sample(const float* feats, int B, int T, const icz_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seq_out && logp_out && r, "butd sample: null argument");
    ICZ_REQUIRE(B > 0 && T > 0, "butd sample: bad B/T");
    ICZ_REQUIRE(fresh, "butd: call icz_butd_refresh_weights after binding/updating parameters");
    ICZ_TRY(ensure_train(B, T));
    rng = *r;
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 1; cur_B = B; cur_T = T; cur_train = true; cur_feats = feats;
    cur_rows = B; cur_row0 = 0;
    rows_t.assign(T, B);
    cur_seq = seq_out; cur_logp = logp_out;
    const bool explicit_rng = rng.uniforms || rng.emb_mask || rng.att_mask || rng.out_mask;
    if (explicit_rng || !use_graphs) return sample_impl(feats, B, T, seq_out, logp_out, st);
    const std::vector<uintptr_t> key = {2, (uintptr_t)feats, (uintptr_t)B, (uintptr_t)T, (uintptr_t)seq_out, (uintptr_t)logp_out};
    return run_cached(key, st, [&](hipStream_t s) { return sample_impl(feats, B, T, seq_out, logp_out, s); });
}

int Butd::sample_impl(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_TRY(prologue(feats, B, st));
    return sample_chain(feats, B, T, seq_out, logp_out, st);
}

// Greedy baseline and sampled rollout of one SCST step (Engine.py:258-262) as two concurrent chains: they share the
// per-image prologue and the weights, nothing else, so they run on two streams (fork / join with events; inside a
// capture this becomes two parallel branches of the graph) and fill each other's idle CUs -- a skinny decoder-step
// GEMM occupies one workgroup per CU with its MFMA pipe about half busy.
int Butd::rollouts(const float* feats, int B, int T, const icz_rng* r, int64_t* ids_out, int64_t* seq_out, float* logp_out,
                   hipStream_t st) {
    ICZ_REQUIRE(feats && ids_out && seq_out && logp_out && r, "butd rollouts: null argument");
    ICZ_REQUIRE(B > 0 && T > 0 && B <= dims.max_rows, "butd rollouts: bad B/T");
    ICZ_REQUIRE(fresh, "butd: call icz_butd_refresh_weights after binding/updating parameters");
    // <= 32 images: ONE chain of 2 B decoder rows (sample_chain with row0 = B) instead of two chains that each stream the weights
    const bool merged = B <= merge_small;
    ICZ_TRY(ensure_train(merged ? 2 * B : B, T));
    if (!side_st) {
        ICZ_CHECK_HIP(hipStreamCreateWithFlags(&side_st, hipStreamNonBlocking));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    rng = *r;
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 1; cur_B = B; cur_T = T; cur_train = true; cur_feats = feats;
    cur_rows = merged ? 2 * B : B; cur_row0 = merged ? B : 0;
    rows_t.assign(T, B);
    cur_seq = seq_out; cur_logp = logp_out;
    const bool explicit_rng = rng.uniforms || rng.emb_mask || rng.att_mask || rng.out_mask;
    if (explicit_rng || !use_graphs) return rollouts_impl(feats, B, T, ids_out, seq_out, logp_out, st);
    const std::vector<uintptr_t> key = {4, (uintptr_t)feats, (uintptr_t)B, (uintptr_t)T, (uintptr_t)ids_out, (uintptr_t)seq_out, (uintptr_t)logp_out};
    return run_cached(key, st, [&](hipStream_t s) { return rollouts_impl(feats, B, T, ids_out, seq_out, logp_out, s); });
}

int Butd::rollouts_impl(const float* feats, int B, int T, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_TRY(prologue(feats, B, st));
    if (cur_row0 > 0) return sample_chain(feats, B, T, seq_out, logp_out, st, cur_row0, ids_out);
    if (!concurrent) {
        ICZ_TRY(greedy_chain(feats, B, T, ids_out, nullptr, st, true));
        return sample_chain(feats, B, T, seq_out, logp_out, st);
    }
    ICZ_CHECK_HIP(hipEventRecord(ev_fork, st));
    ICZ_CHECK_HIP(hipStreamWaitEvent(side_st, ev_fork, 0));
    // the sampled chain (the longer one: multinomial draw, dropout) is issued -- and captured -- first: when kernels of both chains are
    // ready the runtime then takes its first.  Measured round 4, three same-box rounds: rollouts 2.758 / 2.774 / 2.769 ms against
    // 2.783 / 2.785 / 2.806 ms with the greedy chain first (profiles/r04_chain_issue_order.log)
    const int ss = sample_chain(feats, B, T, seq_out, logp_out, st);
    const int sg = greedy_chain(feats, B, T, ids_out, nullptr, side_st, true);
    ICZ_CHECK_HIP(hipEventRecord(ev_join, side_st));       // always join, also on error (a capture must be closed)
    ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
    return sg != ICZ_OK ? sg : ss;
}

__global__ void merged_init_kernel(int32_t* img2, int B, int* nany, int T, int64_t* tok_greedy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 2 * B) img2[i] = i % B;
    if (i < T) nany[i] = 0;
    if (i < B) tok_greedy[i] = 1;
}

// The sampled rollout of B rows into the saved-activation slots.  row0 = B (and ids_out): the MERGED chain of a small SCST batch --
// every step carries 2 B decoder rows, the greedy baseline's (evaluation mode, argmax) in front of the sampled rollout's, so that
// the weights are streamed once per step pair instead of once per chain (at <= 32 rows a decoder step is bound by its weight
// stream: 196.68 MB against 8 - 32 x 0.49 MB of per-row data).  Per-row modes live in the kernels (DropCfg::row0,
// sample_select_kernel<true>); the slots hold 2 B rows per step, the backward pass works on the second half (Butd::bptt).
int Butd::sample_chain(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st, int row0, int64_t* ids_out) {
    const size_t H = dims.H;
    const size_t Vp = round4(dims.V);
    const int Bs = B + row0;
    // slot 0 of the state buffers = zeros; unfinished flags = 1, counters = 0, first token = <sta>
    {
        ZeroList z = {{tb.h1, tb.c1, tb.h2, tb.c2}, 4};
        const size_t n = (size_t)Bs * H;
        hipLaunchKernelGGL(zero_bufs_kernel, dim3(cdiv((int)(n / 4), 256)), dim3(256), 0, st, z, n);
    }
    hipLaunchKernelGGL(sample_init_kernel, dim3(cdiv(B > T ? B : T, 256)), dim3(256), 0, st, tb.unf, tb.nunf, tb.tok + row0, B, T);
    if (row0) hipLaunchKernelGGL(merged_init_kernel, dim3(cdiv(2 * B > T ? 2 * B : T, 256)), dim3(256), 0, st, tb.img2, B, tb.nany, T, tb.tok);
    const int* const counts = row0 ? tb.nany : tb.nunf;           // what keeps a step alive
    for (int t = 0; t < T; ++t) {
        int pns = 1;
        // step t > 0 is dead when no row was left unfinished by step t - 1 (the reference breaks out of its loop there, :233): every
        // kernel of it returns at entry, sample_select_kernel writes the zeros the reference's pre-allocated outputs keep
        ICZ_TRY(train_step(feats, Bs, Bs, t, true, st, t > 0, &pns, false, (t > 0 && early_out) ? counts + (t - 1) : nullptr, row0));
        SampleSelArgs a = {};
        a.logits = tb.logit + (size_t)t * Bs * Vp; a.V = dims.V; a.ldl = (int)Vp;
        if (pns > 1) {          // the predict GEMM left split-K slabs in the chain's workspace (train_step: tb.X[0])
            a.logits = tb.X[0]; a.ns = pns; a.slab_stride = (size_t)Bs * Vp; a.bias = P.predict_b;
            a.logits_store = tb.logit + (size_t)t * Bs * Vp;
        }
        a.uniforms = rng.uniforms ? rng.uniforms + (size_t)t * B : nullptr;
        a.seed_p = d_seed; a.t = t; a.T = T;
        a.unfinished = tb.unf; a.n_unfinished = tb.nunf;
        a.live_rows = tb.live_rows;
        a.seq_out = seq_out; a.logp_out = logp_out;
        a.row0 = row0; a.ids_out = ids_out; a.g_unfinished = tb.gunf; a.n_any = tb.nany;
        a.it_next = tb.tok + (size_t)(t + 1) * Bs;
        a.draw_out = tb.draw + (size_t)t * Bs; a.lse_out = tb.lse + (size_t)t * Bs;
        if (t + 1 < T) {             // the next step's input embedding, fused (step t + 1's slot and dropout stream)
            a.emb_table = P.embed_weight; a.emb_next = tb.emb + (size_t)(t + 1) * Bs * dims.E; a.E = dims.E;
            a.emb_drop = make_drop(d_seed, true, rng.emb_mask, (size_t)B * dims.E, RNG_EMB, t + 1);
        }
        kprof_mark(KP_SAMPLE_SELECT, true, st);
        launch_sample_select(st, Bs, a);
        kprof_mark(KP_SAMPLE_SELECT, false, st);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Butd::sample_mask_sum(float* out, hipStream_t st) {
    ICZ_REQUIRE(mode == 1, "butd: no rollout stored (call icz_butd_sample first)");
    ICZ_REQUIRE(out, "null output");
    // reuse the loss kernel with zero reward (coef scratch is overwritten later by backward)
    ICZ_CHECK_HIP(hipMemsetAsync(tb.loss_rows, 0, sizeof(float) * cur_B * cur_T, st));
    hipLaunchKernelGGL(reinforce_loss_kernel, dim3(1), dim3(256), 0, st, cur_logp, cur_seq, tb.loss_rows, cur_B, cur_T,
                       (const float*)nullptr, tb.coef, (float*)nullptr, out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Butd::sample_backward(const float* reward, const icz_butd_params* G, float* loss_out, float* mask_sum_out,
                          float mask_sum_global, hipStream_t st) {
    ICZ_REQUIRE(mode == 1, "butd: no rollout stored (call icz_butd_sample first)");
    ICZ_REQUIRE(reward && G, "butd sample_backward: null argument");
    if (mask_sum_global >= 0.f)      // < 0: keep the device value set by icz_butd_set_mask_sum_global
        hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, (uint64_t*)nullptr, (uint64_t)0, d_msum_global, mask_sum_global);
    mode = 0;   // the saved logits are consumed
    bptt_early_out = true;
    ICZ_TRY(bptt_prelude(st));      // outside the captured graph
    const bool explicit_rng = rng.uniforms || rng.emb_mask || rng.att_mask || rng.out_mask;
    if (explicit_rng || !use_graphs) return sample_backward_impl(reward, *G, loss_out, mask_sum_out, st);
    std::vector<uintptr_t> key = {3, (uintptr_t)reward, (uintptr_t)loss_out, (uintptr_t)mask_sum_out, (uintptr_t)cur_B, (uintptr_t)cur_T,
                                  (uintptr_t)cur_feats, (uintptr_t)cur_seq, (uintptr_t)cur_logp,
                                  (uintptr_t)cur_rows, (uintptr_t)cur_row0};      // merged / unmerged rollouts share the caller's buffers: slot strides differ
    const float* const* gp = reinterpret_cast<const float* const*>(G);
    for (size_t i = 0; i < sizeof(icz_butd_params) / sizeof(float*); ++i) key.push_back((uintptr_t)gp[i]);
    const icz_butd_params Gc = *G;
    if (!grad_cb) return run_cached(key, st, [&](hipStream_t s) { return sample_backward_impl(reward, Gc, loss_out, mask_sum_out, s); });
    // The DP hook must fire on every call (a replayed graph would not call it): the backward is cut at the three points where a
    // gradient group is complete, every piece is its own captured graph, and the hook is called between the replays -- the
    // all-reduce of a group then starts beside the remaining pieces exactly as in the eager form.
    for (int ph = 0; ph < 4; ++ph) {
        std::vector<uintptr_t> k2 = key;
        k2.push_back(0x100 + ph);
        ICZ_TRY(run_cached(k2, st, [&](hipStream_t s) { return sample_backward_impl(reward, Gc, loss_out, mask_sum_out, s, 1 << ph, false); }));
        if (ph < 3) grad_cb(grad_cb_user, ph);
    }
    return ICZ_OK;
}

int Butd::sample_backward_impl(const float* reward, const icz_butd_params& G, float* loss_out, float* mask_sum_out, hipStream_t st,
                               int phases, bool fire_cb) {
    const int B = cur_B, T = cur_T;
    const int Vp = round4(dims.V);
    if (phases & 1) {
        hipLaunchKernelGGL(reinforce_loss_kernel, dim3(1), dim3(256), 0, st, cur_logp, cur_seq, reward, B, T, (const float*)d_msum_global,
                           tb.coef, loss_out, mask_sum_out);
        hipLaunchKernelGGL(reinforce_dlogits_kernel, dim3(cdiv(Vp, 256), T * cur_rows), dim3(256), 0, st, tb.logit, dims.V, Vp,
                           tb.draw, tb.lse, tb.coef, B, T, cur_rows, cur_row0);
        ICZ_CHECK_HIP(hipGetLastError());
    }
    return bptt(G, st, phases, fire_cb);
}

// ------------------------------------------------------------------------------------------------
__global__ void captions_to_tok_kernel(const int64_t* __restrict__ cap, int B, int L, int T, int64_t* __restrict__ tok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;   // i = t*B + b
    if (i >= T * B) return;
    int t = i / B, b = i % B;
    tok[i] = cap[(size_t)b * L + t];
}
__global__ void gather_packed_kernel(const float* __restrict__ logit, int V, int ldl, int B, const int* __restrict__ row_off,
                                     const int* __restrict__ rows_t, int T, float* __restrict__ out) {
    // grid (V/256, T*B): copy logits of active (t,b) to packed row row_off[t] + b
    const int tb_ = blockIdx.y, t = tb_ / B, b = tb_ % B;
    if (b >= rows_t[t]) return;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    out[(size_t)(row_off[t] + b) * V + v] = logit[(size_t)tb_ * ldl + v];
}

int Butd::xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_rng* r,
                     int train, float* packed_out, hipStream_t st) {
    ICZ_REQUIRE(feats && captions && lengths && B > 0 && L > 1, "butd xe_forward: bad arguments");
    int T = 0;
    for (int b = 0; b < B; ++b) {
        ICZ_REQUIRE(lengths[b] >= 1 && lengths[b] <= L - 1, "butd xe_forward: length %d out of range 1..%d", lengths[b], L - 1);
        ICZ_REQUIRE(b == 0 || lengths[b] <= lengths[b - 1], "butd xe_forward: lengths must be sorted in decreasing order (Engine.py:179)");
        if (lengths[b] > T) T = lengths[b];
    }
    ICZ_TRY(ensure_train(B, T));
    if (r) rng = *r; else { rng = {}; }
    ICZ_REQUIRE(!train || r, "butd xe_forward: training mode needs an icz_rng");
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 2; cur_B = B; cur_T = T; cur_train = train != 0; cur_feats = feats;
    cur_rows = B; cur_row0 = 0;
    rows_t.assign(T, 0);
    n_tokens = 0;
    for (int t = 0; t < T; ++t) {
        int c = 0;
        for (int b = 0; b < B; ++b) c += lengths[b] > t;
        rows_t[t] = c;
        n_tokens += c;
    }
    const size_t H = dims.H;
    const size_t Vp = round4(dims.V);
    ICZ_TRY(prologue(feats, B, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tb.h1, 0, sizeof(float) * B * H, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tb.c1, 0, sizeof(float) * B * H, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tb.h2, 0, sizeof(float) * B * H, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tb.c2, 0, sizeof(float) * B * H, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tb.logit, 0, sizeof(float) * (size_t)T * B * Vp, st));
    hipLaunchKernelGGL(captions_to_tok_kernel, dim3(cdiv(T * B, 256)), dim3(256), 0, st, captions, B, L, T, tb.tok);
    cur_captions = captions; cur_L = L;
    // Teacher forcing: no step needs the previous step's logits unless scheduled sampling draws from them -> one vocabulary projection
    // over all time steps after the loop (T B >= 128 rows: a single GEMM on the big-tile kernel instead of T decoder-step GEMMs)
    const bool batched_predict = ss_prob <= 0.f && T * B >= 128;
    const bool batched_embed = ss_prob <= 0.f && T <= EMB_MAX_T && dims.E % 4 == 0;       // likewise the embeddings of all steps: one launch
    if (batched_embed) {
        EmbRows er = {};
        for (int t = 0; t < T; ++t) er.n[t] = rows_t[t];
        hipLaunchKernelGGL(embed_steps_kernel, dim3(cdiv(dims.E, 1024), T * B), dim3(256), 0, st, P.embed_weight, tb.tok, tb.emb, B, dims.E, er,
                           make_drop(d_seed, train != 0, rng.emb_mask, (size_t)B * dims.E, RNG_EMB, 0));
    }
    for (int t = 0; t < T; ++t) {
        if (t >= 2 && ss_prob > 0.f)          // BUTD_Model.py:120-130: this step's tokens, mixed with draws from the previous step's logits
            ICZ_TRY(ss_select_launch(st, rows_t[t], tb.logit + (size_t)(t - 1) * B * Vp, (int)Vp, dims.V, t, B, ss_prob, ss_gate, ss_draw,
                                     d_seed, tb.tok + (size_t)t * B));
        ICZ_TRY(train_step(feats, rows_t[t], B, t, train != 0, st, batched_embed, nullptr, batched_predict));
    }
    if (batched_predict) {
        // logits of all (t, b) rows at once: [T B, H] x w_pred^T + b through the 128 x 128 split-precision kernel (one pass over the
        // vocabulary matrix instead of T).  Rows b >= rows_t[t] hold whatever their h2 slots held: nothing reads them before
        // xe_loss_dlogits_kernel / scatter_packed_kernel overwrite them with zeros.
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {tb.h2d, w_pred, (int)H, (int)H, (int)H, nullptr};
        g.M = T * B; g.N = dims.V; g.out = tb.logit; g.ldo = (int)Vp; g.bias = P.predict_b; g.nsplit = 1;
        ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    }
    if (packed_out) {
        ICZ_TRY(upload_pack_index(st));
        hipLaunchKernelGGL(gather_packed_kernel, dim3(cdiv(dims.V, 256), T * B), dim3(256), 0, st, tb.logit, dims.V, (int)Vp, B,
                           tb.scalars_i, tb.scalars_i + T, T, packed_out);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// device copy of the packed-sequence index: row_off[t] (first packed row of step t) and rows_t[t]
int Butd::upload_pack_index(hipStream_t st) {
    const int T = cur_T;
    std::vector<int> hostv(2 * T);
    int acc = 0;
    for (int t = 0; t < T; ++t) { hostv[t] = acc; hostv[T + t] = rows_t[t]; acc += rows_t[t]; }
    ICZ_REQUIRE(tb.scalars_i && 2 * T <= tb.scalars_i_cap, "butd: pack index capacity");
    ICZ_CHECK_HIP(hipMemcpyAsync(tb.scalars_i, hostv.data(), sizeof(int) * 2 * T, hipMemcpyHostToDevice, st));
    ICZ_CHECK_HIP(hipStreamSynchronize(st));   // the host vector goes out of scope
    return ICZ_OK;
}

__global__ void scatter_packed_kernel(const float* __restrict__ dpacked, int V, int ldl, int B, const int* __restrict__ row_off,
                                      const int* __restrict__ rows_t, int T, float* __restrict__ logit) {
    // inverse of gather_packed_kernel; inactive rows and pad columns become zero
    const int tb_ = blockIdx.y, t = tb_ / B, b = tb_ % B;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= ldl) return;
    float g = 0.f;
    if (b < rows_t[t] && v < V) g = dpacked[(size_t)(row_off[t] + b) * V + v];
    logit[(size_t)tb_ * ldl + v] = g;
}

int Butd::xe_backward_dlogits(const float* dpacked, const icz_butd_params* G, hipStream_t st) {
    ICZ_REQUIRE(mode == 2, "butd: no XE forward stored (call icz_butd_xe_forward first)");
    ICZ_REQUIRE(dpacked && G, "butd xe_backward_dlogits: null argument");
    const int B = cur_B, T = cur_T;
    const int Vp = round4(dims.V);
    ICZ_TRY(upload_pack_index(st));
    hipLaunchKernelGGL(scatter_packed_kernel, dim3(cdiv(Vp, 256), T * B), dim3(256), 0, st, dpacked, dims.V, Vp, B, tb.scalars_i,
                       tb.scalars_i + T, T, tb.logit);
    ICZ_CHECK_HIP(hipGetLastError());
    mode = 0;
    bptt_early_out = false;
    ICZ_TRY(bptt_prelude(st));
    return bptt(*G, st);
}

__global__ void transpose_coef_kernel(const float* __restrict__ dlogp, int n, float* __restrict__ coef) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) coef[i] = dlogp[i];
}

int Butd::sample_backward_dlogp(const float* dlogp, const icz_butd_params* G, hipStream_t st) {
    ICZ_REQUIRE(mode == 1, "butd: no rollout stored (call icz_butd_sample first)");
    ICZ_REQUIRE(dlogp && G, "butd sample_backward_dlogp: null argument");
    const int B = cur_B, T = cur_T;
    const int Vp = round4(dims.V);
    ICZ_CHECK_HIP(hipMemcpyAsync(tb.coef, dlogp, sizeof(float) * B * T, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(reinforce_dlogits_kernel, dim3(cdiv(Vp, 256), T * cur_rows), dim3(256), 0, st, tb.logit, dims.V, Vp,
                       tb.draw, tb.lse, tb.coef, B, T, cur_rows, cur_row0);
    ICZ_CHECK_HIP(hipGetLastError());
    mode = 0;
    bptt_early_out = true;
    ICZ_TRY(bptt_prelude(st));
    return bptt(*G, st);
}

int Butd::xe_backward(float smoothing, const icz_butd_params* G, float* loss_out, float n_tokens_global, hipStream_t st) {
    ICZ_REQUIRE(mode == 2, "butd: no XE forward stored (call icz_butd_xe_forward first)");
    ICZ_REQUIRE(G, "butd xe_backward: null grads");
    const int B = cur_B, T = cur_T;
    const int Vp = round4(dims.V);
    const float n = n_tokens_global > 0.f ? n_tokens_global : (float)n_tokens;
    const float* n_dev = n_tokens_global < 0.f ? d_msum_global : nullptr;      // < 0: the device scalar handed over by *_set_*_global
    ICZ_CHECK_HIP(hipMemsetAsync(tb.loss_rows, 0, sizeof(float) * T * B, st));
    {
        ICZ_REQUIRE(T <= XE_MAX_T, "xe_backward: %d steps exceed %d", T, XE_MAX_T);
        XeRows xr = {};
        for (int t = 0; t < T; ++t) xr.n[t] = rows_t[t];
        hipLaunchKernelGGL(xe_loss_dlogits_kernel, dim3(B, T), dim3(256), 0, st, tb.logit, dims.V, Vp, cur_captions, cur_L, B, xr, smoothing, 1.0f / n, n_dev,
                           tb.loss_rows);
    }
    if (loss_out) hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, st, tb.loss_rows, T * B, 1.0f / n, n_dev, loss_out);
    ICZ_CHECK_HIP(hipGetLastError());
    mode = 0;
    bptt_early_out = false;
    ICZ_TRY(bptt_prelude(st));
    return bptt(*G, st);
}

// ------------------------------------------------------------------------------------------------
// helpers for the backward GEMMs
int Butd::gemm_auto(GemmLayout layout, GemmArgs& g, float* slab, size_t slab_floats, int* ns_out, hipStream_t st) {
    // direct output (nsplit 1) if there are already enough tiles, else slabs
    g.nsplit = g.M <= 64 ? gemm_pick_split(g, STEP_WGS, layout) : gemm_pick_split_balanced(g, layout, slab ? slab_floats : 0);
    if (g.nsplit > 1) {
        ICZ_REQUIRE(slab && gemm_slab_floats(g.M, g.N, g.nsplit) <= slab_floats, "butd: slab buffer too small (%d x %d x %d)", g.nsplit, g.M, g.N);
        g.out = slab; g.ldo = g.N; g.bias = nullptr; g.accumulate = 0;
    }
    if (ns_out) *ns_out = g.nsplit;
    return gemm_f32(layout, g, st);
}

// C (ldc) = A^T B over K rows, written directly (no split): weight gradients
int Butd::wgrad(const float* dY, int ldy, int M, const float* X, int ldx, int N, int K, float* out, int ldo, hipStream_t st, const int* rows_live) {
    // too few 128 x 128 tiles to fill the chip (the attention projections): split K on the large-tile split-precision kernel, sum the slabs
    if (tb.wslab && ldo == N) {
        const int ns = gemm_tn_split_pick(M, N, K);
        if (ns > 1 && (size_t)ns * M * N <= tb.wslab_floats && ((size_t)M * N) % 4 == 0) {
            ICZ_TRY(gemm_tn_split(dY, ldy, M, X, ldx, N, K, ns, tb.wslab, rows_live, st));
            const size_t MN = (size_t)M * N;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, (const float*)tb.wslab, ns, MN, N, (const float*)nullptr, out);
            ICZ_CHECK_HIP(hipGetLastError());
            return ICZ_OK;
        }
    }
    GemmArgs g = {};
    g.nseg = 1;
    g.rows_live = rows_live;
    g.seg[0] = {dY, X, ldy, ldx, K, nullptr};
    g.M = M; g.N = N; g.out = out; g.ldo = ldo; g.nsplit = 1;
    return gemm_f32(GEMM_TN, g, st);
}

int Butd::colsum(const float* X, int K, int N, int ldx, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 32)), dim3(256), 0, st, X, K, N, ldx, out);
    return ICZ_OK;
}

// The transposed weight copies of the per-step dgrad products are rebuilt lazily, by the first backward call after an optimizer
// step: every entry point calls this in front of bptt(), OUTSIDE its captured graph (a replayed graph must not rebuild them).
int Butd::bptt_prelude(hipStream_t st) {
    if (wt_lm_ih && !wt_fresh) ICZ_TRY(refresh_transposes(st));
    return ICZ_OK;
}

// phases: bit 0 = predict layer + reverse-time loop, bit 1 = embedding and TD-LSTM weight gradients, bit 2 = LM-LSTM weight
// gradients, bit 3 = attention block, biases, joins.  After each of the first three the corresponding gradient group is complete
// in stream order (icz_butd_set_grad_callback); fire_cb = false leaves the callbacks to the caller, which replays every phase
// as its own captured graph and calls them in between.
//
// Every ICZ_TRY inside the loop / behind it sits in a lambda: whatever fails, the side stream is joined before the status is
// returned (inside a capture an unjoined fork would hide the original error behind a capture failure).
int Butd::bptt(const icz_butd_params& G, hipStream_t st, int phases, bool fire_cb) {
    // B rows are worked on per step; the buffers hold Bs rows per step, of which those are rows [roff, roff + B) (Bs = B, roff = 0
    // except behind a merged chain, whose evaluation-mode rows in front carry zero gradient rows through the GEMMs over all steps)
    const int B = cur_B, T = cur_T, Bs = cur_rows, roff = cur_row0;
    const int H = dims.H, D = dims.D, E = dims.E, A = dims.A, R = dims.R, V = dims.V;
    const int Vp = round4(V);
    const int TB = T * Bs;
    const float* feats = cur_feats;
    const size_t sH = (size_t)Bs * H;
    // backward of a sampled rollout: the GEMMs over all (t, b) rows stop behind the last step the rollout ran (GemmArgs::rows_live)
    const int* const rl = (bptt_early_out && early_out) ? tb.live_rows : nullptr;

    // ---- predict layer, all time steps at once.  d h2drop feeds the BPTT chain; the weight / bias gradients of
    //      `predict` depend only on dlogits, so they run on the side stream concurrently with the (skinny,
    //      latency-bound) BPTT chain and are joined at the end.
    if (!low_st) {
        int lo = 0, hi = 0;
        ICZ_CHECK_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = least urgent
        ICZ_CHECK_HIP(hipStreamCreateWithPriority(&low_st, hipStreamNonBlocking, lo));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork2, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork3, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join3, hipEventDisableTiming));
    }
    hipEvent_t ev_fork = ev_fork2, ev_join = ev_join2;
    if (phases & 1) {
    bptt_joined = false;
    // The side branch is forked here (it depends on dlogits only) but ISSUED behind the d h2drop GEMM below, the head of the critical
    // chain: issued first, its 10112 x 1024 x 1280 GEMM took the CUs ahead of that product.  Round 4, three same-box rounds: backward
    // 2.738 / 2.718 / 2.729 ms against 2.777 / 2.828 / 2.803 ms (profiles/r04_backward_issue_order.log); forking it behind the
    // product as well (no overlap with it at all): 2.762 / 2.779 / 2.750 ms.
    auto side_branch = [&]() -> int {
        ICZ_CHECK_HIP(hipStreamWaitEvent(low_st, ev_fork, 0));
        hipStream_t sb = concurrent ? low_st : st;   // low priority: the big GEMM only fills CUs the BPTT chain leaves idle
        int s1 = wgrad(tb.logit, Vp, Vp, tb.h2d, H, H, TB, tb.dWp, H, sb, rl);
        hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(V, 32)), dim3(256), 0, sb, tb.logit, TB, V, (int)Vp, G.predict_b);
        hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(cdiv(V, 4)), dim3(256), 0, sb, tb.dWp, H, P.predict_v, P.predict_g, n_pred,
                           G.predict_v, G.predict_g, V, H);
        ICZ_CHECK_HIP(hipEventRecord(ev_join, sb));
        if (s1 != ICZ_OK) { (void)hipStreamWaitEvent(st, ev_join, 0); return s1; }
        return ICZ_OK;
    };
    ICZ_CHECK_HIP(hipEventRecord(ev_fork, st));
    {
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {tb.logit, w_pred, Vp, H, Vp, nullptr};
        g.M = TB; g.N = H; g.out = tb.dH2d; g.ldo = H; g.rows_live = rl;
        int ns;
        const int sg = gemm_auto(GEMM_NN, g, ws, ws_floats, &ns, st);
        if (sg != ICZ_OK) return sg;          // nothing is forked yet: the side branch is issued behind this product
        if (ns > 1) {
            size_t MN = (size_t)TB * H;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, ws, ns, MN, H, (const float*)nullptr, tb.dH2d);
        }
    }
    ICZ_TRY(side_branch());
    // ---- XE only: rows that dropped out of the batch must contribute zero (the sample path writes every row, and its
    //      accumulators are initialised by the first processed step)
    const bool ragged = rows_t[T - 1] < B || roff > 0;
    if (ragged) {
        ICZ_CHECK_HIP(hipMemsetAsync(tb.dGtd, 0, sizeof(float) * (size_t)TB * 4 * H, st));
        ICZ_CHECK_HIP(hipMemsetAsync(tb.dGlm, 0, sizeof(float) * (size_t)TB * 4 * H, st));
        ICZ_CHECK_HIP(hipMemsetAsync(tb.dDec, 0, sizeof(float) * (size_t)TB * A, st));
        ICZ_CHECK_HIP(hipMemsetAsync(tb.dS, 0, sizeof(float) * (size_t)TB * R, st));
    }

    // ---- reverse-time loop (a lambda: whatever it returns, the side stream is joined behind it)
    auto loop = [&]() -> int {
    int cur = 0;
    int ns1 = 1, ns2 = 1, ns3 = 1, ns4 = 1;
    int bnext = 0;
    for (int t = T - 1; t >= 0; --t) {
        const int bt = rows_t[t];
        const size_t slot = (size_t)t * Bs + roff;
        // REINFORCE backward of a sampled rollout: the steps behind the reference's break never ran (sample_chain) -- their kernels
        // return at entry, the producers of d gates / d dec / ds rows write zeros (the GEMMs over all steps read them), and the
        // first live step takes no carry from the dead one behind it
        const int* const live = (bptt_early_out && early_out && t > 0) ? tb.nunf + (t - 1) : nullptr;
        const int* const carry_live = (bptt_early_out && early_out && t + 1 < T) ? tb.nunf + t : nullptr;
        DropCfg d_out = make_drop(d_seed, cur_train, rng.out_mask, (size_t)B * H, RNG_OUT, t);
        DropCfg d_att = make_drop(d_seed, cur_train, rng.att_mask, (size_t)B * R * A, RNG_ATT, t);
        DropCfg d_off = {0, nullptr, nullptr, 0, 0};
        {   // language LSTM backward (pointwise)
            LstmBwdArgs a = {};
            a.dh_a = bnext ? tb.X[2] : nullptr; a.ns_a = ns3; a.lda_a = H; a.rows_a = bnext;
            a.dhdrop = tb.dH2d + slot * H;
            a.dc_in = bnext ? tb.dc2[cur] : nullptr; a.dc_in_rows = bnext;
            a.gates = tb.glm + slot * 4 * H;
            a.c_prev = tb.c2 + slot * H; a.c_cur = tb.c2 + slot * H + sH;
            a.dgates = tb.dGlm + slot * 4 * H; a.dc_prev = tb.dc2[cur ^ 1];
            a.rows = bt; a.H = H; a.live = live; a.carry_live = carry_live;
            hipLaunchKernelGGL(lstm_bwd_point_kernel, dim3(cdiv(H, 256), bt), dim3(256), 0, st, a, d_out);
        }
        // 33 .. 64 rows: the per-step dgrad products as NT products on the transposed weight copies (resident-activation kernel)
        // (capacity: 4H / 256 slabs of bt x (D + H) resp. (2 x) 4H / 256 slabs of bt x H must fit tb.X[*]; larger models -- H = 1536
        // with D = 2048 already -- take the NN path below, whose split is fitted to the buffer)
        const bool rdg = wt_lm_ih && wt_fresh && bt > 32 && bt <= 64 && gemm_slab_floats(bt, D + H, 4 * H / 256) <= tb.xfloats &&
                         gemm_slab_floats(bt, H, 2 * (4 * H / 256)) <= tb.xfloats;
        // <= 32 rows (small batches, the tail of a ragged XE batch): the NN kernel streams W[k][n] at ~2 TB/s there, the fp32 NT kernel
        // the transposed copies at 3.4 - 4.4 (profiles/r05_scst_b8_*: 12 us for 26 MB against 12 us for 40 MB)
        const bool tdg = small_nt && wt_lm_ih && wt_fresh && bt <= 32;
        if (rdg) {   // X1 = dG_lm . W_ih_lm   [bt, D+H]
            GemmArgs g = {};
            g.nseg = 1;
            g.seg[0] = {tb.dGlm + slot * 4 * H, wt_lm_ih, 4 * H, 4 * H, 4 * H, nullptr};
            g.M = bt; g.N = D + H; g.out = tb.X[0]; g.ldo = D + H; g.live = live;
            g.nsplit = ns1 = gemm_resident_x3_nsplit(g);
            ICZ_REQUIRE(gemm_slab_floats(bt, D + H, ns1) <= tb.xfloats, "butd: slab buffer too small for the dgrad slabs");
            ICZ_TRY(gemm_f32(GEMM_NT, g, st));
        } else if (tdg) {   // <= 32 rows: the same product on the transposed copy through the fp32 NT kernel
            GemmArgs g = {};
            g.nseg = 1;
            g.seg[0] = {tb.dGlm + slot * 4 * H, wt_lm_ih, 4 * H, 4 * H, 4 * H, nullptr};
            g.M = bt; g.N = D + H; g.out = tb.X[0]; g.ldo = D + H; g.live = live;
            ICZ_TRY(gemm_auto(GEMM_NT, g, tb.X[0], tb.xfloats, &ns1, st));
        } else {
            GemmArgs g = {};
            g.nseg = 1;
            g.seg[0] = {tb.dGlm + slot * 4 * H, P.lm_w_ih, 4 * H, D + H, 4 * H, nullptr};
            g.M = bt; g.N = D + H; g.out = tb.X[0]; g.ldo = D + H; g.live = live;
            ICZ_TRY(gemm_auto(GEMM_NN, g, tb.X[0], tb.xfloats, &ns1, st));
        }
        {   // attention backward
            const int dparts = cdiv(D, DALPHA_COLS);
            hipLaunchKernelGGL(att_bwd_dalpha_kernel, dim3(bt, dparts), dim3(256), 0, st, tb.X[0], ns1, D + H, bt, feats, R, D, tb.dalpha, live);
            AttBwdDdecArgs da = {enc_ctx, tb.dec + slot * A, w_aff, tb.alpha + slot * R, tb.dalpha, tb.dDec + slot * A, tb.dS + slot * R, R, A, dparts, live};
            hipLaunchKernelGGL(att_bwd_ddec_kernel, dim3(bt, cdiv(A, 256)), dim3(256), 0, st, da, d_att);
            // X2 = dDec . w_dec   [bt, H]
            GemmArgs g = {};
            g.nseg = 1;
            g.seg[0] = {tb.dDec + slot * A, w_dec, A, H, A, nullptr};
            g.M = bt; g.N = H; g.out = tb.X[1]; g.ldo = H; g.live = live;
            ICZ_TRY(gemm_auto(GEMM_NN, g, tb.X[1], tb.xfloats, &ns2, st));
        }
        {   // TD LSTM backward (pointwise): dh1 = carry + X1[:, D:] + X2
            LstmBwdArgs a = {};
            a.dh_a = bnext ? tb.X[3] : nullptr; a.ns_a = ns4; a.lda_a = H; a.rows_a = bnext;
            a.dh_b = tb.X[0] + D; a.ns_b = ns1; a.lda_b = D + H; a.rows_b = bt;
            a.dh_c = tb.X[1]; a.ns_c = ns2; a.lda_c = H; a.rows_c = bt;
            a.dc_in = bnext ? tb.dc1[cur] : nullptr; a.dc_in_rows = bnext;
            a.gates = tb.gtd + slot * 4 * H;
            a.c_prev = tb.c1 + slot * H; a.c_cur = tb.c1 + slot * H + sH;
            a.dgates = tb.dGtd + slot * 4 * H; a.dc_prev = tb.dc1[cur ^ 1];
            a.rows = bt; a.H = H; a.live = live; a.carry_live = carry_live;
            hipLaunchKernelGGL(lstm_bwd_point_kernel, dim3(cdiv(H, 256), bt), dim3(256), 0, st, a, d_off);
        }
        if (t > 0 && rdg) {   // d h2_{t-1} (X3) and d h1_{t-1} (X4) in one launch
            GemmArgs g3 = {}, g4 = {};
            g3.nseg = 2;
            g3.seg[0] = {tb.dGlm + slot * 4 * H, wt_lm_hh, 4 * H, 4 * H, 4 * H, nullptr};
            g3.seg[1] = {tb.dGtd + slot * 4 * H, wt_td_ih_h2, 4 * H, 4 * H, 4 * H, nullptr};
            g3.M = bt; g3.N = H; g3.out = tb.X[2]; g3.ldo = H; g3.live = live;
            g4.nseg = 1; g4.live = live;
            g4.seg[0] = {tb.dGtd + slot * 4 * H, wt_td_hh, 4 * H, 4 * H, 4 * H, nullptr};
            g4.M = bt; g4.N = H; g4.out = tb.X[3]; g4.ldo = H;
            ns3 = gemm_resident_x3_nsplit(g3); ns4 = gemm_resident_x3_nsplit(g4);
            ICZ_REQUIRE(gemm_slab_floats(bt, H, ns3) <= tb.xfloats, "butd: slab buffer too small for the dgrad slabs");
            ICZ_TRY(gemm_resident_x3_pair(g3, g4, st));
        } else if (t > 0 && tdg) {
            {
                GemmArgs g = {};
                g.nseg = 2;
                g.seg[0] = {tb.dGlm + slot * 4 * H, wt_lm_hh, 4 * H, 4 * H, 4 * H, nullptr};
                g.seg[1] = {tb.dGtd + slot * 4 * H, wt_td_ih_h2, 4 * H, 4 * H, 4 * H, nullptr};
                g.M = bt; g.N = H; g.out = tb.X[2]; g.ldo = H; g.live = live;
                ICZ_TRY(gemm_auto(GEMM_NT, g, tb.X[2], tb.xfloats, &ns3, st));
            }
            {
                GemmArgs g = {};
                g.nseg = 1;
                g.seg[0] = {tb.dGtd + slot * 4 * H, wt_td_hh, 4 * H, 4 * H, 4 * H, nullptr};
                g.M = bt; g.N = H; g.out = tb.X[3]; g.ldo = H; g.live = live;
                ICZ_TRY(gemm_auto(GEMM_NT, g, tb.X[3], tb.xfloats, &ns4, st));
            }
        } else if (t > 0) {
            {   // X3 = dG_lm . W_hh_lm + dG_td . W_ih_td[:, :H]   -> d h2_{t-1}
                GemmArgs g = {};
                g.nseg = 2;
                g.seg[0] = {tb.dGlm + slot * 4 * H, P.lm_w_hh, 4 * H, H, 4 * H, nullptr};
                g.seg[1] = {tb.dGtd + slot * 4 * H, P.td_w_ih, 4 * H, H + D + E, 4 * H, nullptr};
                g.M = bt; g.N = H; g.out = tb.X[2]; g.ldo = H; g.live = live;
                ICZ_TRY(gemm_auto(GEMM_NN, g, tb.X[2], tb.xfloats, &ns3, st));
            }
            {   // X4 = dG_td . W_hh_td   -> d h1_{t-1}
                GemmArgs g = {};
                g.nseg = 1;
                g.seg[0] = {tb.dGtd + slot * 4 * H, P.td_w_hh, 4 * H, H, 4 * H, nullptr};
                g.M = bt; g.N = H; g.out = tb.X[3]; g.ldo = H; g.live = live;
                ICZ_TRY(gemm_auto(GEMM_NN, g, tb.X[3], tb.xfloats, &ns4, st));
            }
        }
        bnext = bt;
        cur ^= 1;
    }
    return ICZ_OK;
    };
    const int s_loop = loop();
    if (grad_cb || s_loop != ICZ_OK) {      // the predict branch has long finished beside the loop: join it now so that its gradients can be reduced
        ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
        bptt_joined = true;
    }
    if (s_loop != ICZ_OK) return s_loop;
    }   // phase 0
    if ((phases & 1) && grad_cb && fire_cb) grad_cb(grad_cb_user, 0);
    const int ldtd = H + D + E, ldlm = D + H;
    // The attention block's tail (phase 3: d enc_ctx, two small weight gradients, bias sums, weight-norm backward: ~0.2 ms of kernels
    // that fill a fraction of the chip) depends on the loop only.  When the whole backward is one call without a DP callback it is
    // forked HERE and issued LAST, on the low-priority stream, beside the LSTM weight gradients: backward 2.685 / 2.684 / 2.698 ->
    // 2.659 / 2.662 / 2.649 ms in three same-box rounds (profiles/r04_backward_issue_order.log; forked behind the TD gradients
    // instead: slower, 2.73 - 2.75 ms; round 2 had issued such a branch FIRST and lost 0.25 ms).  With a callback the phases are
    // separate graphs and stay in line.
    const bool tail_side = phases == 0xF && !grad_cb && concurrent;
    if (tail_side) ICZ_CHECK_HIP(hipEventRecord(ev_fork3, st));
    bool tail_forked = false;
    hipStream_t const main_st = st;
    auto behind_loop = [&]() -> int {
    if (phases & 2) {
    // ---- embedding gradient: dEmb = dG_td . W_ih_td[:, H+D:] for all steps, then ordered scatter
    {
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {tb.dGtd, P.td_w_ih + H + D, 4 * H, H + D + E, 4 * H, nullptr};
        g.M = TB; g.N = E; g.out = tb.dEmb; g.ldo = E; g.rows_live = rl;
        int ns;
        ICZ_TRY(gemm_auto(GEMM_NN, g, ws, ws_floats, &ns, st));
        if (ns > 1) {
            size_t MN = (size_t)TB * E;
            hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, ws, ns, MN, E, (const float*)nullptr, tb.dEmb);
        }
        // inactive (t,b) rows have dG = 0 -> dEmb = 0; their token ids are whatever the buffer held (valid ids)
        ICZ_CHECK_HIP(embed_grad_launch(st, tb.tok, TB, tb.dEmb, 1, (size_t)0, tb.emb, cur_train ? 2.0f : 1.0f, E, G.embed_weight, V, 1, rl));
    }
    // ---- weight gradients: one TN GEMM each over all (t, b)
    // (round 5) the three products over all (t, b) share d gates: one launch over the column groups [h2 | emb | h1] (4096 x 3072 at the
    // BASELINE sizes: 183 us against 3 x 85, gemm_big_x3.hip); shapes it does not take go one by one
    const GemmColGroup td_groups[3] = {{tb.h2, H, H, G.td_w_ih, ldtd},                       // h2_{t-1}
                                       {tb.emb, E, E, G.td_w_ih + H + D, ldtd},              // embedding
                                       {tb.h1, H, H, G.td_w_hh, H}};                         // h1_{t-1}
    const bool td_grouped = gemm_tn_grouped_fits(4 * H, TB, td_groups, 3);
    if (td_grouped) ICZ_TRY(gemm_tn_grouped(tb.dGtd, 4 * H, 4 * H, TB, td_groups, 3, rl, st));
    else ICZ_TRY(wgrad(tb.dGtd, 4 * H, 4 * H, tb.h2, H, H, TB, G.td_w_ih, ldtd, st, rl));
    hipLaunchKernelGGL(timesum_kernel, dim3(cdiv((int)((size_t)Bs * 4 * H / 4), 256)), dim3(256), 0, st, tb.dGtd, T, (size_t)Bs * 4 * H, tb.dGsum);
    ICZ_TRY(wgrad(tb.dGsum + (size_t)roff * 4 * H, 4 * H, 4 * H, mean, D, D, B, G.td_w_ih + H, ldtd, st));      // mean features
    if (!td_grouped) {
        ICZ_TRY(wgrad(tb.dGtd, 4 * H, 4 * H, tb.emb, E, E, TB, G.td_w_ih + H + D, ldtd, st, rl));
        ICZ_TRY(wgrad(tb.dGtd, 4 * H, 4 * H, tb.h1, H, H, TB, G.td_w_hh, H, st, rl));
    }
    }   // phase 1
    if ((phases & 2) && grad_cb && fire_cb) grad_cb(grad_cb_user, 1);
    if (phases & 4) {
    const GemmColGroup lm_groups[3] = {{tb.ctx, D, D, G.lm_w_ih, ldlm},                      // ctx_t
                                       {tb.h1 + sH, H, H, G.lm_w_ih + D, ldlm},              // h1_t
                                       {tb.h2, H, H, G.lm_w_hh, H}};                         // h2_{t-1}
    if (gemm_tn_grouped_fits(4 * H, TB, lm_groups, 3)) {
        ICZ_TRY(gemm_tn_grouped(tb.dGlm, 4 * H, 4 * H, TB, lm_groups, 3, rl, st));            // 4096 x 4096: 227 us against 123 + 2 x 85
    } else {
        ICZ_TRY(wgrad(tb.dGlm, 4 * H, 4 * H, tb.ctx, D, D, TB, G.lm_w_ih, ldlm, st, rl));
        ICZ_TRY(wgrad(tb.dGlm, 4 * H, 4 * H, tb.h1 + sH, H, H, TB, G.lm_w_ih + D, ldlm, st, rl));
        ICZ_TRY(wgrad(tb.dGlm, 4 * H, 4 * H, tb.h2, H, H, TB, G.lm_w_hh, H, st, rl));
    }
    }   // phase 2
    if ((phases & 4) && grad_cb && fire_cb) grad_cb(grad_cb_user, 2);
    if (phases & 8) {
    if (tail_side) {
        ICZ_CHECK_HIP(hipStreamWaitEvent(low_st, ev_fork3, 0));
        tail_forked = true;
        st = low_st;
    }
    ICZ_TRY(wgrad(tb.dDec, A, A, tb.h1 + sH, H, H, TB, tb.dWdec, H, st));
    {   // d enc_ctx (sum over time) and the affine-weight partials, from the ds_t recorded by the loop
        AttBwdDencArgs ea = {enc_ctx, tb.dec + (size_t)roff * A, tb.dS + (size_t)roff * R, w_aff, tb.dEnc, tb.dwaff, B, R, A, T,
                             cur_train ? (rng.att_mask ? 1 : 2) : 0, rng.att_mask, (size_t)B * R * A, d_seed, (uint32_t)RNG_ATT, Bs};
        hipLaunchKernelGGL(att_bwd_denc_kernel<20>, dim3(B, ATT_PARTS), dim3(256), sizeof(float) * T * R, st, ea);
    }
    ICZ_TRY(wgrad(tb.dEnc, A, A, feats, D, D, B * R, tb.dWenc, D, st));
    {   // bias gradients: five column sums (+ the b_hh copies, + the identically zero affine bias: softmax shift invariance) in one launch
        ColsumTable ct = {};
        int nb = 0;
        auto add = [&](const float* X, int K, int N, int ldx, float* out, float* out2) {
            ct.j[ct.count++] = {X, K, N, ldx, out, out2, nb};
            nb += cdiv(N, 32);
        };
        add(tb.dGtd, TB, 4 * H, 4 * H, G.td_b_ih, G.td_b_hh);
        add(tb.dGlm, TB, 4 * H, 4 * H, G.lm_b_ih, G.lm_b_hh);
        add(tb.dDec, TB, A, A, G.dec_att_b, nullptr);
        add(tb.dEnc, B * R, A, A, G.enc_att_b, nullptr);
        add(tb.dwaff, B * ATT_PARTS, A, A, tb.dWaff, nullptr);
        add(tb.dwaff, 0, 1, 1, G.affine_b, nullptr);
        hipLaunchKernelGGL(colsum_multi_kernel, dim3(nb), dim3(256), 0, st, ct);
    }
    {   // the attention block's three weight-normed layers
        WeightNormBwdTable wt = {};
        int nb = 0;
        auto add = [&](const float* dw, int lddw, const float* v, const float* g, const float* norm, float* dv, float* dg, int rows, int cols) {
            wt.j[wt.count++] = {dw, lddw, v, g, norm, dv, dg, rows, cols, nb};
            nb += cdiv(rows, 4);
        };
        add(tb.dWenc, D, P.enc_att_v, P.enc_att_g, n_enc, G.enc_att_v, G.enc_att_g, A, D);
        add(tb.dWdec, H, P.dec_att_v, P.dec_att_g, n_dec, G.dec_att_v, G.dec_att_g, A, H);
        add(tb.dWaff, A, P.affine_v, P.affine_g, n_aff, G.affine_v, G.affine_g, 1, A);
        hipLaunchKernelGGL(weight_norm_bwd_multi_kernel, dim3(nb), dim3(256), 0, st, wt);
    }
    }   // phase 3
    return ICZ_OK;
    };
    const int s_tail = behind_loop();
    st = main_st;
    // joins, also on an error (inside a capture an unjoined side stream would hide the original error behind a capture failure)
    if (tail_forked) {
        ICZ_CHECK_HIP(hipEventRecord(ev_join3, low_st));
        ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join3, 0));
    }
    if (((phases & 8) || s_tail != ICZ_OK) && !bptt_joined) {      // join the predict-gradient branch
        ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
        bptt_joined = true;
    }
    if (s_tail != ICZ_OK) return s_tail;
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

// ================================================================================================
using namespace icz;
extern "C" {

int icz_butd_sample(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, const icz_rng* rng,
                    int64_t* seq_out, float* logprobs_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->sample(feats, B, max_len, rng, seq_out, logprobs_out, (hipStream_t)stream);
}
int icz_butd_scst_rollouts(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, const icz_rng* rng,
                           int64_t* greedy_ids_out, int64_t* seq_out, float* logprobs_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->rollouts(feats, B, max_len, rng, greedy_ids_out, seq_out, logprobs_out, (hipStream_t)stream);
}
int icz_butd_sample_backward(icz_butd_t* h, const float* reward, const icz_butd_params* grads, float* loss_out,
                             float* mask_sum_out, float mask_sum_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->sample_backward(reward, grads, loss_out, mask_sum_out, mask_sum_global, (hipStream_t)stream);
}
int icz_butd_sample_backward_dlogp(icz_butd_t* h, const float* dlogp, const icz_butd_params* grads, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->sample_backward_dlogp(dlogp, grads, (hipStream_t)stream);
}
int icz_butd_xe_backward_dlogits(icz_butd_t* h, const float* dpacked, const icz_butd_params* grads, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->xe_backward_dlogits(dpacked, grads, (hipStream_t)stream);
}
int icz_butd_sample_mask_sum(icz_butd_t* h, float* mask_sum_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->sample_mask_sum(mask_sum_out, (hipStream_t)stream);
}
int icz_butd_saved_alphas(icz_butd_t* h, float* alphas_out, void* stream) {
    ICZ_REQUIRE(h && alphas_out, "icz_butd_saved_alphas: null argument");
    Butd* b = reinterpret_cast<Butd*>(h);
    ICZ_REQUIRE(b->mode != 0 && b->tb.alpha, "icz_butd_saved_alphas: no forward pass stored");
    ICZ_REQUIRE(b->cur_row0 == 0, "icz_butd_saved_alphas: not available behind a merged SCST rollout (set option merge_small = 0)");
    const int n = b->cur_B * b->cur_T * b->dims.R;
    hipLaunchKernelGGL(saved_alphas_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, b->tb.alpha, b->cur_T, b->cur_B, 1, b->dims.R, alphas_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}
int icz_butd_set_scheduled_sampling(icz_butd_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms) {
    ICZ_REQUIRE(h, "icz_butd_set_scheduled_sampling: null handle");
    ICZ_REQUIRE(ss_prob >= 0.f && ss_prob <= 1.f, "icz_butd_set_scheduled_sampling: ss_prob %g outside [0, 1]", (double)ss_prob);
    Butd* b = reinterpret_cast<Butd*>(h);
    b->ss_prob = ss_prob; b->ss_gate = gate_uniforms; b->ss_draw = draw_uniforms;
    return ICZ_OK;
}
int icz_butd_xe_forward(icz_butd_t* h, const float* feats, const int64_t* captions, int32_t B, int32_t L,
                        const int32_t* lengths_host, const icz_rng* rng, int32_t train, float* packed_logits_out,
                        void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->xe_forward(feats, captions, B, L, lengths_host, rng, train, packed_logits_out, (hipStream_t)stream);
}
int icz_butd_xe_backward(icz_butd_t* h, float smoothing, const icz_butd_params* grads, float* loss_out,
                         float n_tokens_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Butd*>(h)->xe_backward(smoothing, grads, loss_out, n_tokens_global, (hipStream_t)stream);
}

int icz_adam_clamp_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                        float clip, int32_t step, void* stream) {
    ICZ_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "icz_adam_clamp_step: bad arguments");
    const double b1 = 0.9, b2 = 0.999;
    const float bc1 = (float)(1.0 - pow(b1, (double)step));
    const float sbc2 = (float)sqrt(1.0 - pow(b2, (double)step));
    hipLaunchKernelGGL(adam_clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, param, grad,
                       exp_avg, exp_avg_sq, (size_t)n, lr, clip, bc1, sbc2);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int icz_adam_clamp_multi(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, float lr, float clip, int32_t step, void* stream) {
    ICZ_REQUIRE(count >= 1 && count <= ADAM_MAX_TENSORS, "icz_adam_clamp_multi: count %d out of range 1..%d", count, ADAM_MAX_TENSORS);
    ICZ_REQUIRE(params && grads && exp_avg && exp_avg_sq && numel && step >= 1, "icz_adam_clamp_multi: bad arguments");
    AdamTable tab;
    tab.count = count;
    size_t blocks = 0;
    for (int i = 0; i < count; ++i) {
        ICZ_REQUIRE(params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i] && numel[i] > 0, "icz_adam_clamp_multi: tensor %d invalid", i);
        tab.t[i] = {params[i], grads[i], exp_avg[i], exp_avg_sq[i], (size_t)numel[i], blocks};
        blocks += ((size_t)numel[i] + 1023) / 1024;
    }
    const double b1 = 0.9, b2 = 0.999;
    const float bc1 = (float)(1.0 - pow(b1, (double)step));
    const float sbc2 = (float)sqrt(1.0 - pow(b2, (double)step));
    hipLaunchKernelGGL(adam_clamp_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, tab, lr, clip, bc1, sbc2);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // extern "C"
