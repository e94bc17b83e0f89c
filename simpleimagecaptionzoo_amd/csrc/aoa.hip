// AoADetection captioner, inference paths: feature projection + AoA refiner, decoder step, greedy and beam decoding
// (Models/AoA_Model.py:122-162, 319-336, 403-502, 698-753).  Training paths live in aoa_train.hip.
#include "aoa_impl.h"

namespace icz {

int Aoa::init(const icz_aoa_dims& d) {
    dims = d;
    ICZ_REQUIRE(d.NH > 0 && d.Hd % d.NH == 0 && (d.Hd / d.NH) % 4 == 0, "aoa: hidden size %d must split into %d heads of a multiple of 4 columns", d.Hd, d.NH);
    ICZ_REQUIRE(d.E % 4 == 0 && d.Hd % 4 == 0 && d.D % 4 == 0 && d.V > 3 && d.max_rows > 0 && d.max_len > 0, "aoa: bad dimensions");
    ICZ_REQUIRE(d.R >= 1 && d.R <= 128, "aoa: %d regions per image (supported: 1..128)", d.R);
    const size_t dh = d.Hd / d.NH;
    // per-(image, head) tiles live in LDS: K, V [R][dh+1] + a chunk of queries with its P rows in the refiner, K, V in the
    // decoder.  gfx950 has 160 KB per CU; above the 64 KB default the kernels need the explicit opt-in below (49 regions x
    // 128 columns: 86 KB; 100 regions: K and V take 103 KB and the queries go through in chunks, see self_qc())
    cur_R = d.R;
    ICZ_REQUIRE(self_qc(d.R) >= 1, "aoa: K and V head tiles of %d regions do not fit the LDS budget", d.R);
    const size_t lds_self = self_lds(d.R, self_qc(d.R));
    const size_t lds_dec = (2 * d.R * (dh + 1) + dh + 128 + 4) * sizeof(float);
    if (lds_self > 48 * 1024)
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(mha_self_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_self));
    if (lds_dec > 48 * 1024)
        ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(aoa_dec_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dec));
    Vp = pad_vocab(d.V);
    const size_t rows = d.max_rows, Hd = d.Hd, E = d.E, RR = rows * d.R;
    ICZ_TRY(alloc((void**)&w_pred, sizeof(float) * Vp * Hd));
    ICZ_TRY(alloc((void**)&n_pred, sizeof(float) * d.V));
    ICZ_TRY(alloc((void**)&w_rec, sizeof(float) * 4 * Hd * 2 * Hd));
    static_assert(NL <= AOA_QKV_MAX_LAYERS, "QkvPackTable too small");
    for (int l = 0; l < NL; ++l) {
        ICZ_TRY(alloc((void**)&w_qkv[l], sizeof(float) * 3 * Hd * Hd));
        ICZ_TRY(alloc((void**)&b_qkv[l], sizeof(float) * 3 * Hd));
    }
    ICZ_TRY(alloc((void**)&zeros, sizeof(float) * rows * Hd));
    const size_t nmax = 4 * Hd > (size_t)Vp ? 4 * Hd : (size_t)Vp;
    ws_floats = (size_t)TARGET_WGS * 4096 * 2 + rows * nmax;
    {   // the resident decoder-step GEMMs (33..128 rows) leave one slab per 256-deep k range (Butd::init's rule): the LSTM gates take
        // K = E + 2 Hd, the AoA linear K = 2 Hd with N = 2 Hd -- without this the split shrinks to fit and the launch falls to the fp32 kernel
        const size_t kmax = E + 2 * Hd, r128 = rows < 128 ? rows : 128;
        const size_t need = (kmax / 256 + 1) * r128 * 4 * Hd;
        if (need > ws_floats) ws_floats = need;
    }
    for (int b = 0; b < 2; ++b) {
        Bank& s = bank[b];
        float** ref[] = {&s.xa, &s.xb, &s.ln, &s.o, &s.od, &s.nd, &s.refined, &s.Kd, &s.Vd};
        for (float** p : ref) ICZ_TRY(alloc((void**)p, sizeof(float) * RR * Hd));
        ICZ_TRY(alloc((void**)&s.qkv, sizeof(float) * RR * 3 * Hd));
        ICZ_TRY(alloc((void**)&s.z, sizeof(float) * RR * 2 * Hd));
        ICZ_TRY(alloc((void**)&s.meanf, sizeof(float) * rows * Hd));
        ICZ_TRY(alloc((void**)&s.ws, sizeof(float) * ws_floats));
        ICZ_TRY(alloc((void**)&s.off, sizeof(int32_t) * (rows + 1)));
        ICZ_TRY(alloc((void**)&s.rowmap, sizeof(int32_t) * RR));
    }
    own[0] = bank[0]; own[1] = bank[1];
    use_bank(0);
    for (int i = 0; i < 2; ++i) {
        ICZ_TRY(alloc((void**)&h[i], sizeof(float) * rows * Hd));
        ICZ_TRY(alloc((void**)&m[i], sizeof(float) * rows * Hd));
        ICZ_TRY(alloc((void**)&ctx[i], sizeof(float) * rows * Hd));
    }
    ICZ_TRY(alloc((void**)&emb, sizeof(float) * rows * E));
    float** st[] = {&u, &qn, &Qp, &xatt, &ctxdrop};
    for (float** p : st) ICZ_TRY(alloc((void**)p, sizeof(float) * rows * Hd));
    ICZ_TRY(alloc((void**)&logits, sizeof(float) * rows * Vp));
    ICZ_TRY(alloc((void**)&it, sizeof(int64_t) * rows));
    ICZ_TRY(alloc((void**)&amax_val, sizeof(float) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&amax_idx, sizeof(int) * rows * ARGMAX_PARTS));
    ICZ_TRY(alloc((void**)&d_seed, 16));
    ICZ_TRY(alloc((void**)&d_msum, 16));
    ICZ_CHECK_HIP(hipDeviceSynchronize());      // alloc() zero-fills on the NULL stream; callers use non-blocking streams (see ensure_train)
    return ICZ_OK;
}

__global__ __launch_bounds__(256) void aoa_pack_rec_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, float* __restrict__ w_rec,
                                                           int Hd, int E) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)4 * Hd * 2 * Hd) return;
    const size_t row = i / (2 * Hd);
    const int c = (int)(i % (2 * Hd));
    w_rec[i] = c < Hd ? w_ih[row * (E + Hd) + E + c] : w_hh[row * Hd + (c - Hd)];
}

int Aoa::refresh(hipStream_t st) {
    ICZ_REQUIRE(bound, "aoa: parameters not bound");
    hipLaunchKernelGGL(weight_norm_kernel, dim3(cdiv(dims.V, 4)), dim3(256), 0, st, P.predict_v, P.predict_g, w_pred, n_pred, dims.V, dims.Hd);
    const size_t n = (size_t)4 * dims.Hd * 2 * dims.Hd;
    hipLaunchKernelGGL(aoa_pack_rec_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, P.lstm_w_ih, P.lstm_w_hh, w_rec, dims.Hd, dims.E);
    QkvPackTable qt = {};
    for (int l = 0; l < NL; ++l) {
        const icz_aoa_block& b = P.layer[l];
        qt.w[l][0] = b.q_w; qt.w[l][1] = b.k_w; qt.w[l][2] = b.v_w;
        qt.b[l][0] = b.q_b; qt.b[l][1] = b.k_b; qt.b[l][2] = b.v_b;
        qt.wdst[l] = w_qkv[l]; qt.bdst[l] = b_qkv[l];
    }
    hipLaunchKernelGGL(aoa_qkv_pack_kernel, dim3(cdiv(dims.Hd * dims.Hd / 4, 256), 3, NL), dim3(256), 0, st, qt, dims.Hd);
    ICZ_CHECK_HIP(hipGetLastError());
    fresh = true;
    return ICZ_OK;
}

// C = sum_s A_s W_s^T + bias, dense [M,N]; split-K slabs through `ws` when one pass would leave most CUs idle
static int aoa_linear(Aoa& a, GemmArgs& g, const float* bias, float* out, hipStream_t st) {
    g.out = out; g.ldo = g.N;
    g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, Aoa::STEP_WGS), a.ws_floats);
    if (g.nsplit == 1) {
        g.bias = bias;
        return gemm_f32(GEMM_NT, g, st);
    }
    ICZ_REQUIRE(gemm_slab_floats(g.M, g.N, g.nsplit) <= a.ws_floats, "aoa: workspace too small");
    g.out = a.ws;
    ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    const size_t MN = (size_t)g.M * g.N;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, a.ws, g.nsplit, MN, g.N, bias, out);
    return ICZ_OK;
}

// refiner self-attention of one layer over n_img images: the matrix-pipe kernel up to 64 regions, the register-blocked one beyond
void Aoa::launch_mha_self(int n_img, int R, int qc, size_t lds, const RegionRows& rr, const float* qkv_, float* o_, const DropP& dp, hipStream_t st) {
    const int Hd = dims.Hd, NH = dims.NH;
    if (mha_mfma && R <= 64 && (Hd / NH) % 64 == 0)
        hipLaunchKernelGGL(mha_self_mfma_kernel, dim3(n_img, NH), dim3(256), sizeof(float) * 64 * 68, st, qkv_, qkv_ + Hd, qkv_ + 2 * Hd, o_, R, Hd, NH, rr, dp, 3 * Hd);
    else
        hipLaunchKernelGGL(mha_self_kernel, dim3(n_img, NH), dim3(256), lds, st, qkv_, qkv_ + Hd, qkv_ + 2 * Hd, o_, R, Hd, NH, qc, rr, dp, 3 * Hd);
}

int Aoa::lin(const float* A, int M, int K, const float* W, const float* bias, int N, float* out, hipStream_t st) {
    GemmArgs g = {};
    g.nseg = 1;
    g.seg[0] = {A, W, K, K, K, nullptr};
    g.M = M; g.N = N;
    return aoa_linear(*this, g, bias, out, st);
}

// img_feats_porjection + AoA_Refine_Core (AoA_Model.py:661-665, 140-162) -> refined [n_img,R,Hd], its region mean, and the
// decoder block's linear_K / linear_V of it (time-invariant, hoisted out of the decoding loop)
// proj != null: img_feats_porjection(feats) has been computed (Aoa::project, packed rows for 'adaptive' batches): only its ReLU / dropout runs
int Aoa::refine(const float* feats, int n_img, bool train, hipStream_t st, const float* proj) {
    point_bank_at_own(cur_bank);          // (a pass of its own never writes into the halves of a paired pass another chain may still read)
    const int R = cur_R, Hd = dims.Hd, NH = dims.NH;
    ICZ_REQUIRE(!lens || lens_n == n_img, "aoa: region counts were set for %d images, the batch has %d (icz_aoa_set_regions)", lens_n, n_img);
    const int rows = (int)region_row_count(n_img);
    const RegionRows rr = region_rows();
    const size_t nel = (size_t)rows * Hd;
    const unsigned eb = (unsigned)((nel + 255) / 256);
    const float* xin = feats;
    if (lens && !proj) {      // 'adaptive' features: the refiner runs on the valid rows only (packed)
        if (!bank[0].featp)
        {
            for (int b = 0; b < 2; ++b) ICZ_TRY(alloc((void**)&bank[b].featp, sizeof(float) * (size_t)dims.max_rows * dims.R * dims.D));
            ICZ_CHECK_HIP(hipDeviceSynchronize());      // alloc() zero-fills on the NULL stream (see ensure_train)
        }
        float* featp = bank[cur_bank].featp;
        hipLaunchKernelGGL(aoa_offsets_kernel, dim3(1), dim3(256), 0, st, lens, n_img, R, off, rowmap);
        const size_t n4 = (size_t)rows * (dims.D / 4);
        hipLaunchKernelGGL(aoa_pack_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, feats, (const int32_t*)rowmap, featp,
                           (size_t)rows, dims.D);
        xin = featp;
    }
    if (!proj) ICZ_TRY(lin(xin, rows, dims.D, P.proj_w, P.proj_b, Hd, xa, st));
    hipLaunchKernelGGL(relu_drop_kernel, dim3(eb), dim3(256), 0, st, proj ? proj : (const float*)xa, xa, nel,
                       dropp(train, rng.proj_mask, 0, AOA_RNG_PROJ, 0, 0.5f), rr, Hd);
    const int qc = self_qc(R);
    const size_t lds = self_lds(R, qc);
    float *cur = xa, *nxt = xb;
    for (int l = 0; l < NL; ++l) {
        const icz_aoa_block& b = P.layer[l];
        hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, cur, b.ln_g, b.ln_b, ln, rows, Hd, (float*)nullptr);
        ICZ_TRY(lin(ln, rows, Hd, w_qkv[l], b_qkv[l], 3 * Hd, qkv, st));         // linear_Q | linear_K | linear_V in one GEMM
        launch_mha_self(n_img, R, qc, lds, rr, qkv, o, dropp(train, rng.ref_att_mask, (size_t)l * n_img * NH * R * R, AOA_RNG_REF_ATT, l, 0.1f), st);
        const float *xo = o, *xn = ln;
        if (train) {
            hipLaunchKernelGGL(drop_concat_kernel, dim3(eb), dim3(256), 0, st, o, ln, od, nd, (size_t)rows, Hd, rr,
                               dropp(true, rng.ref_aoa_mask, (size_t)l * n_img * R * 2 * Hd, AOA_RNG_REF_AOA, l, 0.3f));
            xo = od; xn = nd;
        }
        GemmArgs g = {};
        g.nseg = 2;
        g.seg[0] = {xo, b.aoa_w, Hd, 2 * Hd, Hd, nullptr};
        g.seg[1] = {xn, b.aoa_w + Hd, Hd, 2 * Hd, Hd, nullptr};
        g.M = rows; g.N = 2 * Hd;
        ICZ_TRY(aoa_linear(*this, g, b.aoa_b, z, st));
        hipLaunchKernelGGL(glu_residual_kernel, dim3(eb), dim3(256), 0, st, z, cur, nxt, (size_t)rows, Hd, rr,
                           dropp(train, rng.ref_sc_mask, (size_t)l * n_img * R * Hd, AOA_RNG_REF_SC, l, 0.1f));
        float* t_ = cur; cur = nxt; nxt = t_;
    }
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, st, cur, P.ref_ln_g, P.ref_ln_b, refined, rows, Hd, (float*)nullptr);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(cdiv(Hd, 256), n_img), dim3(256), 0, st, refined, meanf, Hd, rr);
    ICZ_TRY(lin(refined, rows, Hd, P.dec.k_w, P.dec.k_b, Hd, Kd, st));
    ICZ_TRY(lin(refined, rows, Hd, P.dec.v_w, P.dec.v_b, Hd, Vd, st));
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// The two refiner passes of an SCST step -- evaluation mode for the greedy baseline (Engine.py:256-258), training mode for the sampled
// rollout (:259-262) -- as ONE pass over [evaluation rows; training rows] (fixed region counts): every Linear is one GEMM of twice the rows
// (at 2 x 2304: the fused Q/K/V projection fills 216 tiles of 256 x 256 and runs on the eight-wave kernel), every small kernel one launch;
// dropout applies to the second half only, with the indices a pass of its own would use (DropP::idx0), so both halves hold the bits of
// the separate passes.  proj = img_feats_porjection(feats) before ReLU / dropout (Aoa::project), shared by both halves.
int Aoa::refine_pair(int n_img, hipStream_t st, const float* proj) {
    ICZ_REQUIRE(!lens && proj && dual.xa, "aoa refine_pair: fixed region counts, a shared projection and the pair buffers are needed");
    const int R = cur_R, Hd = dims.Hd, NH = dims.NH;
    const int rows = n_img * R, rows2 = 2 * rows, n2 = 2 * n_img;
    const RegionRows rr = region_rows();
    const size_t nel = (size_t)rows * Hd;
    const unsigned eb = (unsigned)((nel + 255) / 256), eb2 = (unsigned)((2 * nel + 255) / 256);
    const Bank& w = dual;
    auto second = [&](DropP d, size_t elems_first) { d.idx0 = elems_first; return d; };
    hipLaunchKernelGGL(relu_drop_kernel, dim3(eb), dim3(256), 0, st, proj, w.xa, nel, dropp(false, nullptr, 0, AOA_RNG_PROJ, 0, 0.5f), rr, Hd);
    hipLaunchKernelGGL(relu_drop_kernel, dim3(eb), dim3(256), 0, st, proj, w.xa + nel, nel, dropp(true, rng.proj_mask, 0, AOA_RNG_PROJ, 0, 0.5f), rr, Hd);
    const int qc = self_qc(R);
    const size_t lds = self_lds(R, qc);
    float *cur = w.xa, *nxt = w.xb;
    for (int l = 0; l < NL; ++l) {
        const icz_aoa_block& b = P.layer[l];
        hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(rows2, 4)), dim3(256), 0, st, cur, b.ln_g, b.ln_b, w.ln, rows2, Hd, (float*)nullptr);
        ICZ_TRY(lin(w.ln, rows2, Hd, w_qkv[l], b_qkv[l], 3 * Hd, w.qkv, st));
        launch_mha_self(n2, R, qc, lds, rr, w.qkv, w.o,
                        second(dropp(true, rng.ref_att_mask, (size_t)l * n_img * NH * R * R, AOA_RNG_REF_ATT, l, 0.1f), (size_t)n_img * NH * R * R), st);
        // dropout of [attention output | normed input] for the training half, in place (each thread rewrites the element it read)
        hipLaunchKernelGGL(drop_concat_kernel, dim3(eb), dim3(256), 0, st, w.o + nel, w.ln + nel, w.o + nel, w.ln + nel, (size_t)rows, Hd, rr,
                           dropp(true, rng.ref_aoa_mask, (size_t)l * n_img * R * 2 * Hd, AOA_RNG_REF_AOA, l, 0.3f));
        GemmArgs g = {};
        g.nseg = 2;
        g.seg[0] = {w.o, b.aoa_w, Hd, 2 * Hd, Hd, nullptr};
        g.seg[1] = {w.ln, b.aoa_w + Hd, Hd, 2 * Hd, Hd, nullptr};
        g.M = rows2; g.N = 2 * Hd;
        ICZ_TRY(aoa_linear(*this, g, b.aoa_b, w.z, st));
        hipLaunchKernelGGL(glu_residual_kernel, dim3(eb2), dim3(256), 0, st, w.z, cur, nxt, (size_t)rows2, Hd, rr,
                           second(dropp(true, rng.ref_sc_mask, (size_t)l * n_img * R * Hd, AOA_RNG_REF_SC, l, 0.1f), nel));
        float* t_ = cur; cur = nxt; nxt = t_;
    }
    hipLaunchKernelGGL(layer_norm_kernel, dim3(cdiv(rows2, 4)), dim3(256), 0, st, cur, P.ref_ln_g, P.ref_ln_b, w.refined, rows2, Hd, (float*)nullptr);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(cdiv(Hd, 256), n2), dim3(256), 0, st, w.refined, w.meanf, Hd, rr);
    ICZ_TRY(lin(w.refined, rows2, Hd, P.dec.k_w, P.dec.k_b, Hd, w.Kd, st));
    ICZ_TRY(lin(w.refined, rows2, Hd, P.dec.v_w, P.dec.v_b, Hd, w.Vd, st));
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// img_feats_porjection alone (fixed region counts), for the two refiner passes of an SCST step to share (Aoa::rollouts)
int Aoa::project(const float* feats, int n_img, float* out, hipStream_t st) {
    return lin(feats, n_img * cur_R, dims.D, P.proj_w, P.proj_b, dims.Hd, out, st);
}

// One decoder step (AoA_Model.py:319-336)
int Aoa::step(const AoaStepIO& s, hipStream_t st) {
    const int rows = s.rows, Hd = dims.Hd, E = dims.E, NH = dims.NH, R = cur_R, dh = Hd / NH;
    const unsigned eb = (unsigned)(((size_t)rows * Hd + 255) / 256);
    if (!s.emb_ready) hipLaunchKernelGGL(embed_kernel, dim3(cdiv(E, 1024), rows), dim3(256), 0, st, P.embed_weight, s.it, s.emb, rows, E, s.d_emb, 1);
    if (!s.u_ready) hipLaunchKernelGGL(aoa_u_kernel, dim3(eb), dim3(256), 0, st, meanf, s.img_of_row, s.ctx_in, s.u, rows, Hd, s.d_ctx);
    GemmArgs g = {};
    g.nseg = 3;
    g.seg[0] = {s.emb, P.lstm_w_ih, E, E + Hd, E, nullptr};
    g.seg[1] = {s.u, P.lstm_w_ih + E, Hd, E + Hd, Hd, nullptr};
    g.seg[2] = {s.h_in, P.lstm_w_hh, Hd, Hd, Hd, nullptr};
    g.M = rows; g.N = 4 * Hd; g.out = ws; g.ldo = 4 * Hd; g.live = s.live;
    g.nsplit = gemm_fit_split(GEMM_NT, g, gemm_pick_split(g, STEP_WGS), ws_floats);
    ICZ_REQUIRE(gemm_slab_floats(g.M, g.N, g.nsplit) <= ws_floats, "aoa: workspace too small");
    ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    LstmPointArgs a = {ws, g.nsplit, nullptr, nullptr, P.lstm_b_ih, P.lstm_b_hh, s.m_in, s.h_out, s.m_out, s.gates_out, nullptr, rows, Hd, s.live};
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    launch_lstm_point(a, off, st);
    hipLaunchKernelGGL(layer_norm_kernel, dim3(rows), dim3(64), 0, st, s.h_out, P.dec.ln_g, P.dec.ln_b, s.qn, rows, Hd, s.ln_stats, s.live);
    const size_t lds = sizeof(float) * (2 * R * (dh + 1) + dh + 128 + 4);
    {   // query projection; when it is split over K its slabs go straight to the attention kernel, which sums them
        GemmArgs qg = {};
        qg.nseg = 1;
        qg.seg[0] = {s.qn, P.dec.q_w, Hd, Hd, Hd, nullptr};
        qg.M = rows; qg.N = Hd; qg.ldo = Hd; qg.live = s.live;
        qg.nsplit = gemm_fit_split(GEMM_NT, qg, gemm_pick_split(qg, STEP_WGS), ws_floats);
        if (qg.nsplit == 1) {
            qg.out = s.Qp; qg.bias = P.dec.q_b;
            ICZ_TRY(gemm_f32(GEMM_NT, qg, st));
            hipLaunchKernelGGL(aoa_dec_attn_kernel, dim3(rows, NH), dim3(256), lds, st, s.Qp, Kd, Vd, s.img_of_row, s.xatt, s.P_out, s.Pd_out, R, Hd, NH,
                               region_rows(), s.d_att, 1, (size_t)0, (const float*)nullptr, (float*)nullptr, s.live);
        } else {
            qg.out = ws;
            ICZ_TRY(gemm_f32(GEMM_NT, qg, st));
            hipLaunchKernelGGL(aoa_dec_attn_kernel, dim3(rows, NH), dim3(256), lds, st, (const float*)ws, Kd, Vd, s.img_of_row, s.xatt, s.P_out, s.Pd_out,
                               R, Hd, NH, region_rows(), s.d_att, qg.nsplit, (size_t)rows * Hd, (const float*)P.dec.q_b, s.Qp, s.live);
        }
    }
    GemmArgs zg = {};
    zg.nseg = 2;
    zg.seg[0] = {s.xatt, P.dec.aoa_w, Hd, 2 * Hd, Hd, nullptr};
    zg.seg[1] = {s.qn, P.dec.aoa_w + Hd, Hd, 2 * Hd, Hd, nullptr};
    zg.M = rows; zg.N = 2 * Hd; zg.out = ws; zg.ldo = 2 * Hd; zg.live = s.live;
    zg.nsplit = gemm_fit_split(GEMM_NT, zg, gemm_pick_split(zg, STEP_WGS), ws_floats);
    ICZ_REQUIRE(gemm_slab_floats(zg.M, zg.N, zg.nsplit) <= ws_floats, "aoa: workspace too small");
    ICZ_TRY(gemm_f32(GEMM_NT, zg, st));
    hipLaunchKernelGGL(aoa_glu_kernel, dim3(eb), dim3(256), 0, st, ws, zg.nsplit, P.dec.aoa_b, s.z_out, s.ctx_out, s.ctxdrop, rows, Hd, s.d_out,
                       (const float*)meanf, s.img_of_row, s.u_next, s.d_ctx_next, s.live);
    if (!s.skip_predict) ICZ_TRY(gemm_predict(s.ctxdrop, Hd, w_pred, P.predict_b, rows, dims.V, Vp, s.logits, Vp, ws, ws_floats, s.pred_nsplit, st, s.live));
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

static AoaStepIO scratch_io(Aoa& a, int rows, const int32_t* img_of_row, int cur) {
    AoaStepIO s = {};
    s.rows = rows; s.img_of_row = img_of_row; s.it = a.it;
    s.h_in = a.h[cur]; s.m_in = a.m[cur]; s.ctx_in = a.ctx[cur];
    s.h_out = a.h[cur ^ 1]; s.m_out = a.m[cur ^ 1]; s.ctx_out = a.ctx[cur ^ 1];
    s.emb = a.emb; s.u = a.u; s.qn = a.qn; s.Qp = a.Qp; s.xatt = a.xatt; s.ctxdrop = a.ctxdrop; s.logits = a.logits;
    s.d_emb = a.dropbits(false, nullptr, 0, 0, 0);
    s.d_ctx = s.d_att = s.d_out = a.dropp(false, nullptr, 0, 0, 0, 0.5f);
    return s;
}

static int zero_state(Aoa& a, int rows, hipStream_t st) {
    ZeroList zl = {};
    zl.p[0] = a.h[0]; zl.p[1] = a.m[0]; zl.p[2] = a.ctx[0]; zl.count = 3;
    const size_t n = (size_t)rows * a.dims.Hd;
    hipLaunchKernelGGL(zero_bufs_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, zl, n);
    return ICZ_OK;
}

// AoA_Decoder.sample (AoA_Model.py:289-345) behind AoADetection_Captioner.sampler (:698-714)
// scst = true (the baseline of an SCST step, rollouts_impl): once EVERY row has emitted <end> the kernels of the remaining steps return
// at entry and their ids are 0 -- nothing behind a row's <end> reaches the reward (Utils.py:354); icz_aoa_greedy never does this
int Aoa::greedy(const float* feats, int B, int T, int64_t* ids_out, hipStream_t st, const float* proj, bool scst, bool refined_ready) {
    ICZ_REQUIRE(feats && ids_out && B > 0 && B <= dims.max_rows && T > 0, "aoa greedy: bad arguments");
    ICZ_REQUIRE(fresh, "aoa: call icz_aoa_refresh_weights after binding/updating parameters");
    use_bank(0);
    if (!refined_ready) ICZ_TRY(refine(feats, B, false, st, proj));
    ICZ_TRY(zero_state(*this, B, st));
    int* const gn = (scst && early_out && gnunf && tcap_T >= T && tcap_B >= B) ? gnunf : nullptr;
    hipLaunchKernelGGL(greedy_init_kernel, dim3(cdiv(B > T ? B : T, 256)), dim3(256), 0, st, it, B, gn, T);
    int cur = 0;
    bool track = false;
    for (int t = 0; t < T; ++t) {
        AoaStepIO s = scratch_io(*this, B, nullptr, cur);
        s.emb_ready = t > 0;
        s.u_ready = t > 0;                       // left by the previous step's GLU kernel (evaluation mode: no dropout)
        if (t + 1 < T) { s.u_next = u; s.d_ctx_next = s.d_ctx; }
        int pns = 1;
        s.pred_nsplit = &pns;
        if (track && t > 0) s.live = gn + (t - 1);
        ICZ_TRY(step(s, st));
        if (t == 0) track = gn && pns > 1;       // the one-launch select keeps the count (the two-kernel argmax of <= 32 rows does not)
        if (pns > 1)         // 33 - 64 rows: slabs of the vocabulary projection -> token + next embedding in one launch
            hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(1024), 0, st, (const float*)ws, dims.V, Vp, pns, (size_t)B * Vp,
                               (const float*)P.predict_b, P.embed_weight, dims.E, emb, it, ids_out, T, t, 1,
                               track ? gunf : (uint8_t*)nullptr, track ? gn : (int*)nullptr);
        else {
            hipLaunchKernelGGL(argmax_part_kernel, dim3(B, ARGMAX_PARTS), dim3(256), 0, st, logits, dims.V, Vp, ARGMAX_PARTS, amax_val, amax_idx);
            hipLaunchKernelGGL(embed_argmax_kernel, dim3(cdiv(dims.E, 1024), B), dim3(256), 0, st, amax_val, amax_idx, ARGMAX_PARTS,
                               P.embed_weight, dims.E, emb, it, ids_out, T, t, 1);
        }
        cur ^= 1;
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

// AoA_Decoder.beam_search_sample (AoA_Model.py:403-502), batched over images; state (h, m, ctx) re-gathered by source beam
int Aoa::beam_search(const float* feats, int n_img, int kb, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seqs_out && lens_out, "aoa beam: null argument");
    ICZ_REQUIRE(kb >= 1 && kb <= BEAM_MAX_K, "aoa beam: beam size %d out of range 1..%d", kb, BEAM_MAX_K);
    ICZ_REQUIRE(n_img > 0 && (long)n_img * kb <= dims.max_rows, "aoa beam: %d images x %d beams exceed row capacity %d", n_img, kb, dims.max_rows);
    ICZ_REQUIRE(max_steps >= 1 && max_steps <= 256, "aoa beam: max_steps out of range");
    ICZ_REQUIRE(fresh, "aoa: call icz_aoa_refresh_weights after binding/updating parameters");
    const int rows = n_img * kb, L = max_steps + 1, Hd = dims.Hd;
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
        ICZ_CHECK_HIP(hipHostMalloc((void**)&bm.n_live_host, sizeof(int) * 4, 0));
        ICZ_CHECK_HIP(hipDeviceSynchronize());      // alloc() zero-fills on the NULL stream (see ensure_train)
        bm.cap_rows = (int)R_;
        bm.cap_L = (int)L_;
    }
    use_bank(0);
    ICZ_TRY(refine(feats, n_img, false, st));
    ICZ_CHECK_HIP(hipMemsetAsync(bm.n_live, 0, sizeof(int) * 260, st));
    ICZ_CHECK_HIP(hipMemsetAsync(bm.run, 0, sizeof(float) * rows, st));
    hipLaunchKernelGGL(beam_init_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, n_img, kb, L, bm.n_act, bm.seqs[0], bm.img_of_row, it,
                       bm.has_complete, bm.best_score);
    ICZ_TRY(zero_state(*this, rows, st));
    int sb = 0, steps_done = 0;
    for (int stp = 1; stp <= max_steps; ++stp) {
        // step 1: the kb rows of an image are identical and only row 0 is scored -> one decoder row per image (butd_beam.hip)
        const bool compact = stp == 1 && kb > 1;
        AoaStepIO s = compact ? scratch_io(*this, n_img, nullptr, 0) : scratch_io(*this, rows, bm.img_of_row, 0);
        s.emb_ready = false;
        ICZ_TRY(step(s, st));
        BeamArgs a = {logits, dims.V, Vp, kb, stp, L, bm.n_act, bm.run, bm.seqs[sb], bm.seqs[sb ^ 1], bm.src_row, it,
                      bm.best_score, bm.best_len, bm.best_seq, bm.has_complete, bm.n_live + stp};
        launch_beam_rowtopk(st, rows, a.logits, a.V, a.ldl, a.k, a.step, (const int*)bm.n_act, (const float*)bm.run, bm.cand_val, bm.cand_idx,
                            compact ? 1 : 0);
        hipLaunchKernelGGL(beam_merge_kernel, dim3(n_img), dim3(64), 0, st, a, (const float*)bm.cand_val, (const int*)bm.cand_idx);
        hipLaunchKernelGGL(beam_gather_kernel, dim3(cdiv(Hd, 1024), rows), dim3(256), 0, st, bm.src_row, Hd, h[1], m[1], ctx[1], h[1],
                           h[0], m[0], ctx[0], u, compact ? kb : 1);
        sb ^= 1;
        steps_done = stp;
        if (stp >= 6 && (stp % 3) == 0 && stp < max_steps) {
            ICZ_CHECK_HIP(hipMemcpyAsync(bm.n_live_host, bm.n_live + stp, sizeof(int), hipMemcpyDeviceToHost, st));
            ICZ_CHECK_HIP(hipStreamSynchronize(st));
            if (bm.n_live_host[0] == 0) break;
        }
    }
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(n_img), dim3(64), 0, st, kb, L, steps_done, bm.n_act, bm.run, bm.seqs[sb], bm.has_complete,
                       bm.best_len, bm.best_seq, seqs_out, lens_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

// ================================================================================================
using namespace icz;
extern "C" {

int icz_aoa_create(const icz_aoa_dims* dims, icz_aoa_t** out) {
    ICZ_REQUIRE(dims && out, "icz_aoa_create: null argument");
    Aoa* n = new Aoa();
    int s = n->init(*dims);
    if (s != ICZ_OK) { delete n; return s; }
    *out = reinterpret_cast<icz_aoa_t*>(n);
    return ICZ_OK;
}
int icz_aoa_destroy(icz_aoa_t* h) { delete reinterpret_cast<Aoa*>(h); return ICZ_OK; }
int icz_aoa_bind_params(icz_aoa_t* h, const icz_aoa_params* p) {
    ICZ_REQUIRE(h && p, "icz_aoa_bind_params: null argument");
    const float* const* q = reinterpret_cast<const float* const*>(p);
    for (size_t i = 0; i < sizeof(icz_aoa_params) / sizeof(float*); ++i) {
        ICZ_REQUIRE(q[i] != nullptr, "icz_aoa_bind_params: parameter pointer %zu is null", i);
        ICZ_REQUIRE(((uintptr_t)q[i] & 15) == 0, "icz_aoa_bind_params: parameter %zu not 16-byte aligned", i);
    }
    Aoa* n = reinterpret_cast<Aoa*>(h);
    if (n->bound && memcmp(&n->P, p, sizeof(*p)) != 0) {      // captured graphs carry the old parameter addresses
        ICZ_CHECK_HIP(hipDeviceSynchronize());
        n->gc.clear();
    }
    n->P = *p; n->bound = true; n->fresh = false;
    return ICZ_OK;
}
int icz_aoa_set_option(icz_aoa_t* h, const char* name, int32_t value) {
    ICZ_REQUIRE(h && name, "icz_aoa_set_option: null argument");
    Aoa* n = reinterpret_cast<Aoa*>(h);
    if (strcmp(name, "graphs") == 0) { n->use_graphs = value != 0; return ICZ_OK; }
    const bool eo = strcmp(name, "early_out") == 0, rp = strcmp(name, "refine_pair") == 0, mm = strcmp(name, "mha_mfma") == 0;
    if (eo || rp || mm) {
        ICZ_CHECK_HIP(hipDeviceSynchronize());      // a replay of a graph about to be destroyed may still be in flight
        n->gc.clear();
        if (eo) n->early_out = value != 0;
        if (rp) n->pair_refine = value != 0;
        if (mm) n->mha_mfma = value != 0;
        return ICZ_OK;
    }
    set_error("icz_aoa_set_option: unknown option '%s'", name);
    return ICZ_ERR_INVALID;
}
int icz_aoa_refresh_weights(icz_aoa_t* h, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->refresh((hipStream_t)stream);
}
int icz_aoa_refine(icz_aoa_t* h, const float* feats, int32_t B, float* refined_out, void* stream) {
    ICZ_REQUIRE(h && feats && refined_out, "icz_aoa_refine: null argument");
    Aoa* n = reinterpret_cast<Aoa*>(h);
    ICZ_REQUIRE(B > 0 && B <= n->dims.max_rows, "icz_aoa_refine: B out of range");
    ICZ_REQUIRE(n->fresh, "aoa: call icz_aoa_refresh_weights after binding/updating parameters");
    n->use_bank(0);
    hipStream_t st = (hipStream_t)stream;
    ICZ_TRY(n->refine(feats, B, false, st));
    const size_t bytes = sizeof(float) * (size_t)B * n->cur_R * n->dims.Hd;
    if (!n->lens) {
        ICZ_CHECK_HIP(hipMemcpyAsync(refined_out, n->refined, bytes, hipMemcpyDeviceToDevice, st));
    } else {         // packed rows back to [B, regions, Hd]; the padding rows (never computed) are zero
        ICZ_CHECK_HIP(hipMemsetAsync(refined_out, 0, bytes, st));
        const size_t rows = n->region_row_count(B), n4 = rows * (n->dims.Hd / 4);
        hipLaunchKernelGGL(aoa_unpack_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, (const float*)n->refined,
                           (const int32_t*)n->rowmap, refined_out, rows, n->dims.Hd);
    }
    return ICZ_OK;
}
int icz_aoa_set_regions(icz_aoa_t* h, int32_t regions, const int32_t* counts_dev, const int32_t* counts_host, int32_t n_img) {
    ICZ_REQUIRE(h, "null handle");
    Aoa* n = reinterpret_cast<Aoa*>(h);
    ICZ_REQUIRE(regions >= 1 && regions <= n->dims.R, "icz_aoa_set_regions: %d regions outside 1..%d (the handle's capacity)", regions, n->dims.R);
    ICZ_REQUIRE((counts_dev == nullptr) == (counts_host == nullptr), "icz_aoa_set_regions: pass the counts on both sides or on neither");
    if (counts_host) {
        ICZ_REQUIRE(n_img >= 1 && n_img <= n->dims.max_rows, "icz_aoa_set_regions: %d images out of range", n_img);
        for (int i = 0; i < n_img; ++i)
            ICZ_REQUIRE(counts_host[i] >= 1 && counts_host[i] <= regions, "icz_aoa_set_regions: image %d has %d regions (1..%d)", i, counts_host[i], regions);
    }
    n->cur_R = regions; n->lens = counts_dev; n->lens_n = counts_dev ? n_img : 0;
    n->cur_total = 0;
    for (int i = 0; counts_host && i < n_img; ++i) n->cur_total += counts_host[i];
    n->mode = 0;
    return ICZ_OK;
}
int icz_aoa_greedy(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, int64_t* ids_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->greedy(feats, B, max_len, ids_out, (hipStream_t)stream);
}
int icz_aoa_beam_search(icz_aoa_t* h, const float* feats, int32_t n_img, int32_t beam, int32_t max_steps, float* seqs_out,
                        int32_t* lens_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->beam_search(feats, n_img, beam, max_steps, seqs_out, lens_out, (hipStream_t)stream);
}

}  // extern "C"
