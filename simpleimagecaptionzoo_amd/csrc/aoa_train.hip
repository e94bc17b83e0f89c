// AoADetection captioner, training paths: sampled rollout (sampler_rl, AoA_Model.py:716-734 -> AoA_Decoder.sample_rl :347-401),
// teacher-forced forward (:676-696 -> AoA_Decoder.forward :231-287) and BPTT for the decoder parameters.
//
// Backward layout (same scheme as the BUTD decoder): the reverse-time loop only runs what is truly sequential -- per step
// three NN dgrad GEMMs (GLU input, attention query, LSTM recurrence) and four pointwise kernels; every weight gradient is
// one TN GEMM over all (t, b) rows after the loop.  The gradients of the hoisted linear_K / linear_V outputs are
// accumulated per image over time by the one workgroup that owns the (image, head) slice (no atomics, fixed order).
#include "aoa_impl.h"

namespace icz {
namespace {

// GLU backward + the two dropouts that feed ctx_t:
//   dctx = drop_out'(dCd) + drop_ctx'(du_{t+1})      (ctx_t is the predict input at t and the LSTM input at t+1)
//   z = [a | b], ctx = a * sigmoid(b):  da = dctx * s,  db = dctx * a * s * (1 - s)
__global__ __launch_bounds__(256) void aoa_glu_bwd_kernel(const float* __restrict__ dCd, DropP d_out, const float* __restrict__ du_next, int ns,
                                                          int rows_next, DropP d_ctx_next, const float* __restrict__ z, float* __restrict__ dz,
                                                          int rows, int Hd, const int* __restrict__ live = nullptr,
                                                          const int* __restrict__ carry_live = nullptr) {
    if (step_dead(live)) return;               // the step never ran (icz_common.h); its dz rows were zeroed in front of the loop
    if (step_dead(carry_live)) du_next = nullptr;      // the step behind this one never ran: no carry
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)rows * Hd) return;
    const size_t row = i / Hd;
    const int c = (int)(i % Hd);
    float d = d_out.apply(dCd[i], i);
    if (du_next && row < (size_t)rows_next) {
        const float g = sum_slabs1(du_next, ns, (size_t)rows_next * 2 * Hd, row * 2 * Hd + c);
        d += d_ctx_next.apply(g, i);
    }
    const float a = z[row * 2 * Hd + c], b = z[row * 2 * Hd + Hd + c];
    const float s = sigmoidf_(b);
    dz[row * 2 * Hd + c] = d * s;
    dz[row * 2 * Hd + Hd + c] = d * a * s * (1.f - s);
}

// Decoder attention backward, one workgroup per (row, head); row b attends image b (training paths have no beams).
//   dPd_r = dx_h . V_h[r];  dP = keep/(1-p) * dPd;  dS = P (dP - sum P dP);  dQ_h = sum_r dS_r K_h[r] / sqrt(d)
// dK_h[r] = sum_t dS_t[r] Q_t,h / sqrt(d) and dV_h[r] = sum_t Pd_t[r] dx_t,h are sums over time: the steps only record dS_t
// (already scaled by 1/sqrt(d)) and dx_t, and aoa_dkv_kernel forms both sums once after the loop -- accumulating them
// step by step would read-modify-write the two [B, R, Hd] tensors (38 MB) in every step.
// dx = columns [0, Hd) of the GLU-input gradient slabs [ns][rows][2Hd].
__global__ __launch_bounds__(256) void aoa_dec_attn_bwd_kernel(const float* __restrict__ dxq, int ns, int rows, const float* __restrict__ Pm,
                                                               const float* __restrict__ Pdm, const float* __restrict__ Qp,
                                                               const float* __restrict__ Kd, const float* __restrict__ Vd,
                                                               float* __restrict__ dQp, float* __restrict__ dS_out, float* __restrict__ dx_out, int R, int Hd,
                                                               int NH, RegionRows rr, float keep_scale, const int* __restrict__ live = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float sm_db[];    // K tile, V tile [R][d+1], q [d], dx [d], dS [128], red [4]
    if (step_dead(live)) return;
    const int row = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = Hd / NH, ld = d + 1;
    float* sk = sm_db;
    float* sv = sk + R * ld;
    float* sq = sv + R * ld;
    float* sdx = sq + d;
    float* sds = sdx + d;
    float* red = sds + 128;
    const int len = rr.count(row);
    const size_t base = rr.first(row) * Hd + (size_t)hd * d;
    aoa_stage_kv<256>(Kd + base, Vd + base, sk, sv, len, d, Hd, tid);
    const size_t MN = (size_t)rows * 2 * Hd;
    for (int j = tid; j < d; j += 256) {
        sq[j] = Qp[(size_t)row * Hd + (size_t)hd * d + j];
        sdx[j] = sum_slabs1(dxq, ns, MN, (size_t)row * 2 * Hd + (size_t)hd * d + j);
    }
    __syncthreads();
    const size_t pidx = ((size_t)row * NH + hd) * R + tid;
    float p = 0.f, dP = 0.f;
    if (tid < len) {
        p = Pm[pidx];
        float acc = 0.f;
        for (int j = 0; j < d; ++j) acc += sdx[j] * sv[tid * ld + j];
        dP = Pdm[pidx] != 0.f ? acc * keep_scale : 0.f;
    }
    const float wdot = wave_sum(p * dP);
    if (lane == 0) red[wave] = wdot;
    __syncthreads();
    const float dot = red[0] + red[1];
    if (tid < 128) {
        const float dS = p * (dP - dot) / sqrtf((float)d);
        sds[tid] = dS;
        if (tid < R) dS_out[pidx] = dS;
    }
    __syncthreads();
    for (int j = tid; j < d; j += 256) {
        float acc = 0.f;
        for (int r = 0; r < len; ++r) acc += sds[r] * sk[r * ld + j];
        dQp[(size_t)row * Hd + (size_t)hd * d + j] = acc;
        dx_out[(size_t)row * Hd + (size_t)hd * d + j] = sdx[j];
    }
}

// dKd[img, r, head cols] = sum_t dS_t[img, head, r] Qp_t[img, head cols];  dVd = sum_t Pd_t[r] dx_t   (grid (B, NH), 256 threads).
// The steps go through LDS in chunks of TC (all of them at the usual 20 steps x 36 regions).
__global__ __launch_bounds__(256) void aoa_dkv_kernel(const float* __restrict__ dS_all, const float* __restrict__ Pd_all,
                                                      const float* __restrict__ Qp_all, const float* __restrict__ dx_all,
                                                      float* __restrict__ dKd, float* __restrict__ dVd, int B, int T, int TC, int R, int Hd, int NH,
                                                      RegionRows rr) {
    extern __shared__ __attribute__((aligned(16))) float sm_kv[];       // dS [TC][R], Pd [TC][R], Qp [TC][d], dx [TC][d]
    const int img = blockIdx.x, hd = blockIdx.y, tid = threadIdx.x;
    const int d = Hd / NH;
    float* sds = sm_kv;
    float* spd = sds + TC * R;
    float* sq = spd + TC * R;
    float* sdx = sq + TC * d;
    const size_t base = rr.first(img) * Hd + (size_t)hd * d;
    const int len = rr.count(img);
    for (int t0 = 0; t0 < T; t0 += TC) {
        const int nt = min(TC, T - t0);
        for (int i = tid; i < nt * R; i += 256) {
            const int t = t0 + i / R, r = i % R;
            const size_t g = (((size_t)t * B + img) * NH + hd) * R + r;
            sds[i] = dS_all[g];
            spd[i] = Pd_all[g];
        }
        for (int i = tid; i < nt * d; i += 256) {
            const int t = t0 + i / d, j = i % d;
            const size_t g = ((size_t)t * B + img) * Hd + (size_t)hd * d + j;
            sq[i] = Qp_all[g];
            sdx[i] = dx_all[g];
        }
        __syncthreads();
        for (int i = tid; i < len * d; i += 256) {
            const int r = i / d, j = i % d;
            const size_t o = base + (size_t)r * Hd + j;
            float dk = t0 ? dKd[o] : 0.f, dv = t0 ? dVd[o] : 0.f;
            for (int t = 0; t < nt; ++t) {
                dk += sds[t * R + r] * sq[t * d + j];
                dv += spd[t * R + r] * sdx[t * d + j];
            }
            dKd[o] = dk;
            dVd[o] = dv;
        }
        __syncthreads();
    }
}

// Custom LayerNorm backward (AoA_Model.py:14-25: unbiased std, eps outside the sqrt), one workgroup per row.
//   dq = dq_a (slabs [ns_a][rows][Hd], linear_Q dgrad) + dq_b (slabs [ns_b][rows][2Hd] at column offset Hd, GLU-input dgrad)
//   y = g (x - mean) inv + b, inv = 1/(std + eps):  dy = dq g;  dstd = -inv^2 sum(dy xh);
//   dxh = dy inv + dstd xh / (std (n-1));  dx = dxh - mean(dxh)
__global__ __launch_bounds__(256) void aoa_ln_bwd_kernel(const float* __restrict__ dq_a, int ns_a, const float* __restrict__ dq_b, int ns_b, int rows,
                                                         const float* __restrict__ x, const float* __restrict__ stats,
                                                         const float* __restrict__ gain, float* __restrict__ dq_tot, float* __restrict__ dx, int n,
                                                         const int* __restrict__ live = nullptr) {
    // one workgroup per row; a thread keeps its columns (4 adjacent ones per 1024) in registers between the two passes
    __shared__ float sm_red[4];
    if (step_dead(live)) return;
    constexpr int NV = 4;                         // n <= 4096, n % 4 == 0 (checked by the host)
    const int row = blockIdx.x, tid = threadIdx.x;
    const float mean = stats[2 * row], inv = stats[2 * row + 1];
    const float stdv = 1.0f / inv - 1e-6f;
    const float* xr = x + (size_t)row * n;
    f32x4 dy[NV], xc[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = 4 * tid + 1024 * u;
        dy[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        xc[u] = dy[u];
        if (c < n) {
            const f32x4 dq = sum_slabs4(dq_a, ns_a, (size_t)rows * n, (size_t)row * n + c) +
                             sum_slabs4(dq_b, ns_b, (size_t)rows * 2 * n, (size_t)row * 2 * n + n + c);
            *reinterpret_cast<f32x4*>(dq_tot + (size_t)row * n + c) = dq;
            const f32x4 g = *reinterpret_cast<const f32x4*>(gain + c);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dy[u][j] = dq[j] * g[j];
                xc[u][j] = xv[j] - mean;
                s1 += dy[u][j];
                s2 += dy[u][j] * xc[u][j];
            }
        }
    }
    s1 = block_sum_256(s1, sm_red);
    s2 = block_sum_256(s2, sm_red);
    const float dstd = -inv * inv * s2;
    const float kx = dstd / (stdv * (float)(n - 1));
    const float mdx = inv * s1 / (float)n;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int c = 4 * tid + 1024 * u;
        if (c < n) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = dy[u][j] * inv + kx * xc[u][j] - mdx;
            *reinterpret_cast<f32x4*>(dx + (size_t)row * n + c) = o;
        }
    }
}

// prod[row, c] = dq[row, c] * (x[row, c] - mean) * inv     (column sums of it = d gain)
__global__ __launch_bounds__(256) void aoa_ln_prod_kernel(const float* __restrict__ dq, const float* __restrict__ x, const float* __restrict__ stats,
                                                          float* __restrict__ prod, size_t rows, int n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * n) return;
    const size_t row = i / n;
    prod[i] = dq[i] * (x[i] - stats[2 * row]) * stats[2 * row + 1];
}

__global__ void aoa_captions_to_tok_kernel(const int64_t* __restrict__ cap, int B, int L, int T, int64_t* __restrict__ tok) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * B) return;
    tok[i] = cap[(size_t)(i % B) * L + i / B];
}
__global__ void aoa_gather_packed_kernel(const float* __restrict__ logit, int V, int ldl, int B, const int* __restrict__ row_off,
                                         const int* __restrict__ rows_t, float* __restrict__ out) {
    const int tb_ = blockIdx.y, t = tb_ / B, b = tb_ % B;
    if (b >= rows_t[t]) return;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < V) out[(size_t)(row_off[t] + b) * V + v] = logit[(size_t)tb_ * ldl + v];
}

}  // namespace

// Training buffers sized by what the batches ask for (the reference never truncates captions, Datasets.py:47-51: the step
// count of an XE batch is only known when it arrives); growing re-allocates them all.
int Aoa::ensure_train(int Bq, int Tq) {
    if (Bq <= tcap_B && Tq <= tcap_T) return ICZ_OK;
    ICZ_REQUIRE(dims.Hd <= 4096, "aoa: training paths keep a LayerNorm row in registers (hidden size %d > 4096)", dims.Hd);
    ICZ_REQUIRE(Tq <= XE_MAX_T, "aoa: %d steps exceed the limit of %d", Tq, XE_MAX_T);
    if (tcap_B > Bq) Bq = tcap_B;
    if (tcap_T > Tq) Tq = tcap_T;
    if (dims.max_len > Tq) Tq = dims.max_len;
    if (!tallocs.empty()) {
        ICZ_CHECK_HIP(hipDeviceSynchronize());
        for (void* p : tallocs) (void)hipFree(p);
        tallocs.clear();
        tcap_B = tcap_T = 0; mode = 0;
        gc.clear();     // the captured rollout / backward graphs carry the freed addresses in their kernel arguments
    }
    struct Scope { bool& f; Scope(bool& x) : f(x) { f = true; } ~Scope() { f = false; } } scope(alloc_train);
    const size_t B = Bq, T = Tq, Hd = dims.Hd, E = dims.E, R = dims.R, NH = dims.NH;
    {
        const size_t dh = Hd / NH, lds_bwd = sizeof(float) * (2 * R * (dh + 1) + 2 * dh + 128 + 4);
        if (lds_bwd > 48 * 1024)
            ICZ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(aoa_dec_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                              (int)lds_bwd));
    }
    const size_t TB = T * B;
    ICZ_TRY(alloc((void**)&tok, sizeof(int64_t) * (TB + B)));
    float** st1[] = {&th, &tm, &tctx};
    for (float** p : st1) ICZ_TRY(alloc((void**)p, sizeof(float) * (TB + B) * Hd));
    ICZ_TRY(alloc((void**)&temb, sizeof(float) * TB * E));
    float** sth[] = {&tu, &tqn, &tQp, &txatt, &tcd, &dCd, &dQp, &dQn, &prod, &tdX};
    for (float** p : sth) ICZ_TRY(alloc((void**)p, sizeof(float) * TB * Hd));
    ICZ_CHECK_HIP(hipMemset(tcd, 0, sizeof(float) * TB * Hd));      // the batched vocabulary projection of xe_forward reads every (t, b) row
    ICZ_TRY(alloc((void**)&tg, sizeof(float) * TB * 4 * Hd));
    ICZ_TRY(alloc((void**)&dG, sizeof(float) * TB * 4 * Hd));
    ICZ_TRY(alloc((void**)&tz, sizeof(float) * TB * 2 * Hd));
    ICZ_TRY(alloc((void**)&dZ, sizeof(float) * TB * 2 * Hd));
    ICZ_TRY(alloc((void**)&tstats, sizeof(float) * TB * 2));
    ICZ_TRY(alloc((void**)&tP, sizeof(float) * TB * NH * R));
    ICZ_TRY(alloc((void**)&tPd, sizeof(float) * TB * NH * R));
    ICZ_TRY(alloc((void**)&tdS, sizeof(float) * TB * NH * R));
    ICZ_TRY(alloc((void**)&tlogit, sizeof(float) * TB * Vp));
    ICZ_TRY(alloc((void**)&dEmb, sizeof(float) * TB * E));
    ICZ_TRY(alloc((void**)&dKd, sizeof(float) * B * R * Hd));
    ICZ_TRY(alloc((void**)&dVd, sizeof(float) * B * R * Hd));
    ICZ_TRY(alloc((void**)&dHln, sizeof(float) * B * Hd));
    ICZ_TRY(alloc((void**)&dcb[0], sizeof(float) * B * Hd));
    ICZ_TRY(alloc((void**)&dcb[1], sizeof(float) * B * Hd));
    xfloats = (size_t)TARGET_WGS * 4096 * 2 + TB * (Hd > E ? Hd : E) + B * 4 * Hd;
    ICZ_TRY(alloc((void**)&X, sizeof(float) * xfloats));
    ICZ_TRY(alloc((void**)&X2, sizeof(float) * xfloats));
    ICZ_TRY(alloc((void**)&dWp, sizeof(float) * (size_t)Vp * Hd));
    ICZ_TRY(alloc((void**)&coef, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&lse, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&loss_rows, sizeof(float) * TB));
    ICZ_TRY(alloc((void**)&draw, sizeof(int32_t) * TB));
    ICZ_TRY(alloc((void**)&unf, B));
    ICZ_TRY(alloc((void**)&nunf, sizeof(int) * T));
    ICZ_TRY(alloc((void**)&gunf, B));
    ICZ_TRY(alloc((void**)&gnunf, sizeof(int) * T));
    ICZ_TRY(alloc((void**)&live_rows, 16));
    ICZ_TRY(alloc((void**)&pack_idx, sizeof(int) * 2 * T));
    // The hipMemset calls above run on the NULL stream; callers enqueue on NON-BLOCKING streams (torch's), which are not ordered behind
    // it: without this, a kernel of the first call after a (re)allocation could run BEFORE the zero-fill of its buffer and then be
    // wiped by it (round 5: sample_init_kernel's unfinished flags, seen as an all-zero rollout in 1 of 3 five-rank runs).
    ICZ_CHECK_HIP(hipDeviceSynchronize());
    tcap_B = Bq; tcap_T = Tq;
    return ICZ_OK;
}

// step t of a training-mode pass over cur_B rows: inputs / outputs are slots of the saved [T, B, ...] tensors
AoaStepIO Aoa::train_io(int rows, int t, bool train) {
    const size_t B = cur_B, Hd = dims.Hd, E = dims.E, NH = dims.NH, R = cur_R;
    const size_t s0 = (size_t)t * B, s1 = s0 + B;
    AoaStepIO s = {};
    s.rows = rows; s.img_of_row = nullptr; s.it = tok + s0; s.emb_ready = false;
    s.h_in = th + s0 * Hd; s.m_in = tm + s0 * Hd; s.ctx_in = tctx + s0 * Hd;
    s.h_out = th + s1 * Hd; s.m_out = tm + s1 * Hd; s.ctx_out = tctx + s1 * Hd;
    s.emb = temb + s0 * E; s.u = tu + s0 * Hd; s.gates_out = tg + s0 * 4 * Hd; s.ln_stats = tstats + s0 * 2;
    s.qn = tqn + s0 * Hd; s.Qp = tQp + s0 * Hd; s.P_out = tP + s0 * NH * R; s.Pd_out = tPd + s0 * NH * R;
    s.xatt = txatt + s0 * Hd; s.z_out = tz + s0 * 2 * Hd; s.ctxdrop = tcd + s0 * Hd; s.logits = tlogit + s0 * Vp;
    s.d_emb = dropbits(train, rng.emb_mask, s0 * E, RNG_EMB, t);
    s.d_ctx = dropp(train, rng.ctx_mask, s0 * Hd, AOA_RNG_CTX, t, 0.5f);
    s.d_att = dropp(train, rng.att_mask, s0 * NH * R, AOA_RNG_ATT, t, 0.1f);
    s.d_out = dropp(train, rng.out_mask, s0 * Hd, AOA_RNG_OUT, t, 0.5f);
    return s;
}

// everything of a sampled rollout that is host state or must not sit inside a captured graph: buffers, the Philox seed in device
// memory, what the backward pass will read back
int Aoa::sample_prelude(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_REQUIRE(feats && seq_out && logp_out && r && B > 0 && B <= dims.max_rows && T > 0, "aoa sample: bad arguments");
    ICZ_REQUIRE(fresh, "aoa: call icz_aoa_refresh_weights after binding/updating parameters");
    ICZ_TRY(ensure_train(B, T));
    rng = *r;
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 1; cur_B = B; cur_T = T; cur_train = true; cur_seq = seq_out; cur_logp = logp_out;
    rows_t.assign(T, B);
    return ICZ_OK;
}

int Aoa::sample(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_TRY(sample_prelude(feats, B, T, r, seq_out, logp_out, st));
    return sample_impl(feats, B, T, seq_out, logp_out, st);
}

int Aoa::sample_impl(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st, const float* proj, bool refined_ready) {
    use_bank(1);
    if (!refined_ready) ICZ_TRY(refine(feats, B, true, st, proj));
    const size_t sH = (size_t)B * dims.Hd;
    ICZ_CHECK_HIP(hipMemsetAsync(th, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tm, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tctx, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(unf, 1, B, st));
    ICZ_CHECK_HIP(hipMemsetAsync(nunf, 0, sizeof(int) * T, st));
    hipLaunchKernelGGL(fill_i64_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, tok, (int64_t)1, B);
    for (int t = 0; t < T; ++t) {
        const size_t slot = (size_t)t * B;
        AoaStepIO io = train_io(B, t, true);
        io.emb_ready = t > 0;           // written by the previous step's sample_select_kernel
        io.u_ready = t > 0;             // written by the previous step's GLU kernel
        if (t + 1 < T) { const AoaStepIO nx = train_io(B, t + 1, true); io.u_next = nx.u; io.d_ctx_next = nx.d_ctx; }
        int pns = 1;
        io.pred_nsplit = &pns;
        // step t > 0 is dead when step t - 1 left no row unfinished (the reference breaks out of its loop there, AoA_Model.py:400)
        if (t > 0 && early_out) io.live = nunf + (t - 1);
        ICZ_TRY(step(io, st));
        SampleSelArgs a = {};
        a.live_rows = live_rows;
        a.logits = tlogit + slot * Vp; a.V = dims.V; a.ldl = Vp;
        if (pns > 1) {          // the predict GEMM left split-K slabs in the bank's workspace
            a.logits = ws; a.ns = pns; a.slab_stride = (size_t)B * Vp; a.bias = P.predict_b; a.logits_store = tlogit + slot * Vp;
        }
        a.uniforms = rng.uniforms ? rng.uniforms + slot : nullptr;
        a.seed_p = d_seed; a.t = t; a.T = T;
        a.unfinished = unf; a.n_unfinished = nunf; a.seq_out = seq_out; a.logp_out = logp_out;
        a.it_next = tok + slot + B; a.draw_out = draw + slot; a.lse_out = lse + slot;
        if (t + 1 < T) {                // the next step's input embedding (its slot, its dropout stream), fused
            a.emb_table = P.embed_weight; a.emb_next = temb + (slot + B) * dims.E; a.E = dims.E;
            a.emb_drop = dropbits(true, rng.emb_mask, (slot + B) * dims.E, RNG_EMB, t + 1);
        }
        launch_sample_select(st, B, a);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

static bool aoa_explicit_rng(const icz_aoa_rng& r) {
    return r.uniforms || r.proj_mask || r.ref_att_mask || r.ref_aoa_mask || r.ref_sc_mask || r.emb_mask || r.ctx_mask || r.att_mask || r.out_mask;
}

// Greedy baseline (evaluation mode, bank 0, side stream) and sampled rollout (training mode, bank 1) of one SCST step
// (Engine.py:256-261) enqueued as two concurrent chains; identical to greedy() followed by sample().  Round 5: the feature projection
// (2048 -> 1024 over B x 36 rows, 9.7 GFLOP at B = 64) is the same product in both passes up to its ReLU / dropout epilogue
// (AoA_Model.py:661-665, 712-713): computed once in front of the fork; and with option "graphs" the pair is ONE captured hipGraph
// (fixed region counts, Philox randomness), replayed like the BUTD pair.
int Aoa::rollouts(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st) {
    ICZ_REQUIRE(ids_out, "aoa rollouts: null argument");
    if (!side_st) {
        ICZ_CHECK_HIP(hipStreamCreateWithFlags(&side_st, hipStreamNonBlocking));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    ICZ_TRY(sample_prelude(feats, B, T, r, seq_out, logp_out, st));
    if (!lens && !proj_shared) {
        ICZ_TRY(alloc((void**)&proj_shared, sizeof(float) * (size_t)dims.max_rows * dims.R * dims.Hd));
        ICZ_CHECK_HIP(hipDeviceSynchronize());
    }
    if (!lens && pair_refine && !dual.xa) {      // the pair buffers: twice the rows of a bank's
        const size_t RR2 = (size_t)2 * dims.max_rows * dims.R, Hd = dims.Hd;
        float** ref[] = {&dual.xa, &dual.xb, &dual.ln, &dual.o, &dual.refined, &dual.Kd, &dual.Vd};
        for (float** p : ref) ICZ_TRY(alloc((void**)p, sizeof(float) * RR2 * Hd));
        ICZ_TRY(alloc((void**)&dual.qkv, sizeof(float) * RR2 * 3 * Hd));
        ICZ_TRY(alloc((void**)&dual.z, sizeof(float) * RR2 * 2 * Hd));
        ICZ_TRY(alloc((void**)&dual.meanf, sizeof(float) * (size_t)2 * dims.max_rows * Hd));
        ICZ_CHECK_HIP(hipDeviceSynchronize());
    }
    // (host state, outside any captured graph: a replayed rollout pair leaves the banks where this call's batch size puts them)
    if (!lens && pair_refine) point_banks_at_pair(B); else { point_bank_at_own(0); point_bank_at_own(1); }
    if (!use_graphs || lens || aoa_explicit_rng(rng)) return rollouts_impl(feats, B, T, ids_out, seq_out, logp_out, st);
    const std::vector<uintptr_t> key = {1, (uintptr_t)feats, (uintptr_t)B, (uintptr_t)T, (uintptr_t)cur_R, (uintptr_t)ids_out, (uintptr_t)seq_out,
                                        (uintptr_t)logp_out};
    return gc.run(key, st, [&](hipStream_t s) { return rollouts_impl(feats, B, T, ids_out, seq_out, logp_out, s); });
}

int Aoa::rollouts_impl(const float* feats, int B, int T, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st) {
    const float* proj = nullptr;
    if (!lens) {                        // (an 'adaptive' batch packs its rows per bank: each pass projects its own)
        ICZ_TRY(project(feats, B, proj_shared, st));
        proj = proj_shared;
    }
    const bool pair = proj && pair_refine && dual.xa;
    if (pair) ICZ_TRY(refine_pair(B, st, proj));          // both refiner passes in one, in front of the fork
    ICZ_CHECK_HIP(hipEventRecord(ev_fork, st));
    ICZ_CHECK_HIP(hipStreamWaitEvent(side_st, ev_fork, 0));
    const int sg = greedy(feats, B, T, ids_out, side_st, proj, true, pair);
    const int ss = sg == ICZ_OK ? sample_impl(feats, B, T, seq_out, logp_out, st, proj, pair) : sg;
    ICZ_CHECK_HIP(hipEventRecord(ev_join, side_st));       // always join, also on error (a capture must be closed)
    ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join, 0));
    return ss;
}

int Aoa::sample_backward(const float* reward, const icz_aoa_params* G, float* loss_out, float* msum_out, float msum_global, hipStream_t st) {
    ICZ_REQUIRE(mode == 1, "aoa: no rollout stored (call icz_aoa_sample first)");
    ICZ_REQUIRE(reward && G, "aoa sample_backward: null argument");
    if (msum_global >= 0.f)      // < 0: keep the device value handed over by icz_aoa_set_norm_global
        hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, (uint64_t*)nullptr, (uint64_t)0, d_msum, msum_global);
    mode = 0;
    bptt_early_out = true;
    if (!low_st) {          // the side stream of bptt: created here, outside any capture
        ICZ_CHECK_HIP(hipStreamCreateWithFlags(&low_st, hipStreamNonBlocking));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork2, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork3, hipEventDisableTiming));
        ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join3, hipEventDisableTiming));
    }
    // with a DP callback the hook must fire on every call: eager launches (a replayed graph would not call it)
    if (!use_graphs || lens || grad_cb || aoa_explicit_rng(rng)) return sample_backward_impl(reward, *G, loss_out, msum_out, st);
    std::vector<uintptr_t> key = {2, (uintptr_t)reward, (uintptr_t)loss_out, (uintptr_t)msum_out, (uintptr_t)cur_B, (uintptr_t)cur_T, (uintptr_t)cur_R,
                                  (uintptr_t)cur_seq, (uintptr_t)cur_logp,
                                  (uintptr_t)bank[0].refined, (uintptr_t)bank[0].Kd, (uintptr_t)bank[1].refined, (uintptr_t)bank[1].Kd, (uintptr_t)bank[1].Vd, (uintptr_t)cur_bank};      // paired-refine banks (rollouts) or the handle's own (sample)
    const float* const* gp = reinterpret_cast<const float* const*>(G);
    for (size_t i = 0; i < sizeof(icz_aoa_params) / sizeof(float*); ++i) key.push_back((uintptr_t)gp[i]);
    const icz_aoa_params Gc = *G;
    return gc.run(key, st, [&](hipStream_t s) { return sample_backward_impl(reward, Gc, loss_out, msum_out, s); });
}

int Aoa::sample_backward_impl(const float* reward, const icz_aoa_params& G, float* loss_out, float* msum_out, hipStream_t st) {
    const int B = cur_B, T = cur_T;
    hipLaunchKernelGGL(reinforce_loss_kernel, dim3(1), dim3(256), 0, st, cur_logp, cur_seq, reward, B, T, (const float*)d_msum, coef, loss_out, msum_out);
    hipLaunchKernelGGL(reinforce_dlogits_kernel, dim3(cdiv(Vp, 256), T * B), dim3(256), 0, st, tlogit, dims.V, Vp, draw, lse, coef, B, T);
    return bptt(G, st);
}

int Aoa::xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_aoa_rng* r, int train,
                    float* packed_out, hipStream_t st) {
    ICZ_REQUIRE(feats && captions && lengths && B > 0 && B <= dims.max_rows && L > 1, "aoa xe_forward: bad arguments");
    ICZ_REQUIRE(fresh, "aoa: call icz_aoa_refresh_weights after binding/updating parameters");
    ICZ_REQUIRE(!train || r, "aoa xe_forward: training mode needs an icz_aoa_rng");
    int T = 0;
    for (int b = 0; b < B; ++b) {
        ICZ_REQUIRE(lengths[b] >= 1 && lengths[b] <= L - 1, "aoa xe_forward: length %d out of range 1..%d", lengths[b], L - 1);
        ICZ_REQUIRE(b == 0 || lengths[b] <= lengths[b - 1], "aoa xe_forward: lengths must be sorted in decreasing order");
        if (lengths[b] > T) T = lengths[b];
    }
    ICZ_TRY(ensure_train(B, T));
    use_bank(1);
    if (r) rng = *r; else rng = {};
    hipLaunchKernelGGL(set_scalars_kernel, dim3(1), dim3(1), 0, st, d_seed, rng.seed, (float*)nullptr, 0.f);
    mode = 2; cur_B = B; cur_T = T; cur_L = L; cur_train = train != 0; cur_captions = captions;
    rows_t.assign(T, 0);
    n_tokens = 0;
    for (int t = 0; t < T; ++t) {
        int cnt = 0;
        for (int b = 0; b < B; ++b) cnt += lengths[b] > t;
        rows_t[t] = cnt;
        n_tokens += cnt;
    }
    ICZ_TRY(refine(feats, B, cur_train, st));
    const size_t sH = (size_t)B * dims.Hd;
    ICZ_CHECK_HIP(hipMemsetAsync(th, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tm, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tctx, 0, sizeof(float) * sH, st));
    ICZ_CHECK_HIP(hipMemsetAsync(tlogit, 0, sizeof(float) * (size_t)T * B * Vp, st));
    hipLaunchKernelGGL(aoa_captions_to_tok_kernel, dim3(cdiv(T * B, 256)), dim3(256), 0, st, captions, B, L, T, tok);
    // teacher forcing: one vocabulary projection over all (t, b) rows after the loop unless scheduled sampling needs the previous
    // step's logits (butd_train.hip: xe_forward); rows b >= rows_t[t] are zeroed by xe_loss_dlogits_kernel / the scatter kernel
    const bool batched_predict = ss_prob <= 0.f && T * B >= 128;
    for (int t = 0; t < T; ++t) {
        if (t >= 2 && ss_prob > 0.f)          // scheduled sampling (AoA_Model.py:258-270): this step's tokens, mixed with draws from the previous logits
            ICZ_TRY(ss_select_launch(st, rows_t[t], tlogit + (size_t)(t - 1) * B * Vp, (int)Vp, dims.V, t, B, ss_prob, ss_gate, ss_draw, d_seed,
                                     tok + (size_t)t * B));
        AoaStepIO io = train_io(rows_t[t], t, cur_train);
        io.u_ready = t > 0;             // the next step's rows are a prefix of this step's: its u comes from this step's GLU kernel
        if (t + 1 < T) { const AoaStepIO nx = train_io(rows_t[t + 1], t + 1, cur_train); io.u_next = nx.u; io.d_ctx_next = nx.d_ctx; }
        io.skip_predict = batched_predict;
        ICZ_TRY(step(io, st));
    }
    if (batched_predict) {
        GemmArgs g = {};
        g.nseg = 1;
        g.seg[0] = {tcd, w_pred, dims.Hd, dims.Hd, dims.Hd, nullptr};
        g.M = T * B; g.N = dims.V; g.out = tlogit; g.ldo = Vp; g.bias = P.predict_b; g.nsplit = 1;
        ICZ_TRY(gemm_f32(GEMM_NT, g, st));
    }
    if (packed_out) {
        std::vector<int> hostv(2 * T);
        int acc = 0;
        for (int t = 0; t < T; ++t) { hostv[t] = acc; hostv[T + t] = rows_t[t]; acc += rows_t[t]; }
        ICZ_CHECK_HIP(hipMemcpyAsync(pack_idx, hostv.data(), sizeof(int) * 2 * T, hipMemcpyHostToDevice, st));
        ICZ_CHECK_HIP(hipStreamSynchronize(st));
        hipLaunchKernelGGL(aoa_gather_packed_kernel, dim3(cdiv(dims.V, 256), T * B), dim3(256), 0, st, tlogit, dims.V, Vp, B, pack_idx,
                           pack_idx + T, packed_out);
    }
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

int Aoa::xe_backward(float smoothing, const icz_aoa_params* G, float* loss_out, float n_tokens_global, hipStream_t st) {
    ICZ_REQUIRE(mode == 2, "aoa: no XE forward stored (call icz_aoa_xe_forward first)");
    ICZ_REQUIRE(G, "aoa xe_backward: null grads");
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
    return bptt(*G, st);
}

int Aoa::colsum(const float* Xm, int K, int N, int ldx, float* out, hipStream_t st) {
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(N, 32)), dim3(256), 0, st, Xm, K, N, ldx, out);
    return ICZ_OK;
}

// slabs [ns][M][N] = A[M,K] . B[K,N]  (ns == 1: the dense product); `cap` = capacity of slab_out in floats
int Aoa::nn(const float* A, int lda, int M, int K, const float* Bm, int ldb, int N, float* slab_out, size_t cap, int* ns_out, int target,
            hipStream_t st, const int* live, const int* rows_live) {
    GemmArgs g = {};
    g.nseg = 1;
    g.live = live; g.rows_live = rows_live;
    g.seg[0] = {A, Bm, lda, ldb, K, nullptr};
    g.M = M; g.N = N; g.out = slab_out; g.ldo = N;
    g.nsplit = M <= 64 ? gemm_pick_split(g, target, GEMM_NN) : gemm_pick_split_balanced(g, GEMM_NN, cap);
    ICZ_REQUIRE(gemm_slab_floats(M, N, g.nsplit) <= cap, "aoa: slab buffer too small");
    ICZ_TRY(gemm_f32(GEMM_NN, g, st));
    *ns_out = g.nsplit;
    return ICZ_OK;
}

int Aoa::tn(const float* dY, int ldy, int M, const float* Xm, int ldx, int N, int K, float* out, int ldo, int accumulate, hipStream_t st,
            const int* rows_live) {
    GemmArgs g = {};
    g.nseg = 1;
    g.rows_live = rows_live;
    g.seg[0] = {dY, Xm, ldy, ldx, K, nullptr};
    g.M = M; g.N = N; g.out = out; g.ldo = ldo; g.nsplit = 1; g.accumulate = accumulate;
    return gemm_f32(GEMM_TN, g, st);
}

int Aoa::bptt(const icz_aoa_params& G, hipStream_t st) {
    use_bank(1);
    const int B = cur_B, T = cur_T, Hd = dims.Hd, E = dims.E, V = dims.V, NH = dims.NH, R = cur_R, dh = Hd / NH;
    const int TB = T * B;
    const size_t sH = (size_t)B * Hd;
    int ns = 1;
    // backward of a sampled rollout: the steps behind the reference's break (AoA_Model.py:400) never ran -- their kernels return at
    // entry (the buffers the batched GEMMs read are zeroed in front of the loop) and the GEMMs over all (t, b) stop behind the last live step
    const bool eo = bptt_early_out && early_out;
    const int* const rl = eo ? live_rows : nullptr;
    // ---- predict layer over all time steps: d(dropped ctx), weight-norm gradients
    ICZ_TRY(nn(tlogit, Vp, TB, Vp, w_pred, Hd, Hd, X, xfloats, &ns, TARGET_WGS, st, nullptr, rl));
    {
        const size_t MN = (size_t)TB * Hd;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, X, ns, MN, Hd, (const float*)nullptr, dCd);
    }
    // The predict layer's weight / bias gradients need dlogits only.  Without a DP callback they go to a side stream,
    // ISSUED behind the d(ctx) product above (the head of the critical chain, as in Butd::bptt) and joined behind the loop; with a
    // callback they stay in line, because stage 0 reports them complete in stream order right here.
    const bool side = !grad_cb;
    hipStream_t ps = st;
    if (side) {
        if (!low_st) {
            // a plain stream: eager launches on a PRIORITISED stream beside other streams' work can serialise on this runtime
            // (EXPERIMENTS.md, round 4, section 8), and the AoA paths are eager
            ICZ_CHECK_HIP(hipStreamCreateWithFlags(&low_st, hipStreamNonBlocking));
            ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork2, hipEventDisableTiming));
            ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join2, hipEventDisableTiming));
            ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_fork3, hipEventDisableTiming));
            ICZ_CHECK_HIP(hipEventCreateWithFlags(&ev_join3, hipEventDisableTiming));
        }
        ICZ_CHECK_HIP(hipEventRecord(ev_fork2, st));
        ICZ_CHECK_HIP(hipStreamWaitEvent(low_st, ev_fork2, 0));
        ps = low_st;
    }
    const int s_tn = tn(tlogit, Vp, Vp, tcd, Hd, Hd, TB, dWp, Hd, 0, ps, rl);
    if (s_tn == ICZ_OK) {
        (void)colsum(tlogit, TB, V, Vp, G.predict_b, ps);
        hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(cdiv(V, 4)), dim3(256), 0, ps, dWp, Hd, P.predict_v, P.predict_g, n_pred, G.predict_v,
                           G.predict_g, V, Hd);
    }
    if (side) ICZ_CHECK_HIP(hipEventRecord(ev_join2, low_st));
    if (s_tn != ICZ_OK) { if (side) (void)hipStreamWaitEvent(st, ev_join2, 0); return s_tn; }
    if (grad_cb) grad_cb(grad_cb_user, 0);      // predict.* complete in stream order: reduced beside the reverse-time loop
    if (rows_t[T - 1] < B || eo) {      // ragged batch / steps that never ran: those rows contribute exact zeros to the batched GEMMs
        ICZ_CHECK_HIP(hipMemsetAsync(dZ, 0, sizeof(float) * (size_t)TB * 2 * Hd, st));
        ICZ_CHECK_HIP(hipMemsetAsync(dQp, 0, sizeof(float) * (size_t)TB * Hd, st));
        ICZ_CHECK_HIP(hipMemsetAsync(dQn, 0, sizeof(float) * (size_t)TB * Hd, st));
        ICZ_CHECK_HIP(hipMemsetAsync(dG, 0, sizeof(float) * (size_t)TB * 4 * Hd, st));
        ICZ_CHECK_HIP(hipMemsetAsync(tdS, 0, sizeof(float) * (size_t)TB * NH * R, st));
        ICZ_CHECK_HIP(hipMemsetAsync(tdX, 0, sizeof(float) * (size_t)TB * Hd, st));
    }
    const size_t lds = sizeof(float) * (2 * R * (dh + 1) + 2 * dh + 128 + 4);
    DropCfg off = {0, nullptr, nullptr, 0, 0};
    // the reverse-time loop in a lambda: whatever it returns, the side stream is joined behind it (a capture must be closed)
    auto loop = [&]() -> int {
    int cur = 0, nsx = 1, bnext = 0;
    for (int t = T - 1; t >= 0; --t) {
        const int bt = rows_t[t];
        const size_t s0 = (size_t)t * B;
        const AoaStepIO io = train_io(bt, t, cur_train);
        const AoaStepIO io_next = train_io(bnext, t + 1 < T ? t + 1 : t, cur_train);
        const unsigned eb = (unsigned)(((size_t)bt * Hd + 255) / 256);
        const int* const live = (eo && t > 0) ? nunf + (t - 1) : nullptr;
        const int* const carry_live = (eo && t + 1 < T) ? nunf + t : nullptr;
        hipLaunchKernelGGL(aoa_glu_bwd_kernel, dim3(eb), dim3(256), 0, st, dCd + s0 * Hd, io.d_out, bnext ? (const float*)X : nullptr, nsx, bnext,
                           io_next.d_ctx, tz + s0 * 2 * Hd, dZ + s0 * 2 * Hd, bt, Hd, live, carry_live);
        int ns2 = 1, nsq = 1;
        ICZ_TRY(nn(dZ + s0 * 2 * Hd, 2 * Hd, bt, 2 * Hd, P.dec.aoa_w, 2 * Hd, 2 * Hd, X2, xfloats, &ns2, STEP_WGS, st, live));
        hipLaunchKernelGGL(aoa_dec_attn_bwd_kernel, dim3(bt, NH), dim3(256), lds, st, X2, ns2, bt, tP + s0 * NH * R, tPd + s0 * NH * R,
                           tQp + s0 * Hd, Kd, Vd, dQp + s0 * Hd, tdS + s0 * NH * R, tdX + s0 * Hd, R, Hd, NH, region_rows(), io.d_att.mode ? io.d_att.scale : 1.0f, live);
        ICZ_TRY(nn(dQp + s0 * Hd, Hd, bt, Hd, P.dec.q_w, Hd, Hd, ws, ws_floats, &nsq, STEP_WGS, st, live));
        hipLaunchKernelGGL(aoa_ln_bwd_kernel, dim3(bt), dim3(256), 0, st, ws, nsq, X2, ns2, bt, th + (s0 + B) * Hd, tstats + s0 * 2,
                           P.dec.ln_g, dQn + s0 * Hd, dHln, Hd, live);
        LstmBwdArgs a = {};
        a.dh_a = bnext ? X + Hd : nullptr; a.ns_a = nsx; a.lda_a = 2 * Hd; a.rows_a = bnext;
        a.dh_b = dHln; a.ns_b = 1; a.lda_b = Hd; a.rows_b = bt;
        a.dc_in = bnext ? dcb[cur] : nullptr; a.dc_in_rows = bnext;
        a.gates = tg + s0 * 4 * Hd;
        a.c_prev = tm + s0 * Hd; a.c_cur = tm + (s0 + B) * Hd;
        a.dgates = dG + s0 * 4 * Hd; a.dc_prev = dcb[cur ^ 1];
        a.rows = bt; a.H = Hd; a.live = live; a.carry_live = carry_live;
        hipLaunchKernelGGL(lstm_bwd_point_kernel, dim3(cdiv(Hd, 256), bt), dim3(256), 0, st, a, off);
        // [du_t | dh_{t-1}] = dgates_t . [W_ih[:, E:] | W_hh]
        if (t > 0) ICZ_TRY(nn(dG + s0 * 4 * Hd, 4 * Hd, bt, 4 * Hd, w_rec, 2 * Hd, 2 * Hd, X, xfloats, &nsx, STEP_WGS, st, live));
        bnext = bt;
        cur ^= 1;
    }
    return ICZ_OK;
    };
    const int s_loop = loop();
    if (side) ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join2, 0));      // the predict branch has long finished beside the loop
    if (s_loop != ICZ_OK) return s_loop;
    static const bool tail_side_on = [] { const char* e = getenv("ICZ_AOA_TAIL_SIDE"); return e ? atoi(e) != 0 : true; }();      // A/B switch (read once)
    const bool tail_side = side && tail_side_on;
    if (tail_side) ICZ_CHECK_HIP(hipEventRecord(ev_fork3, st));
    // ---- embedding gradient
    ICZ_TRY(nn(dG, 4 * Hd, TB, 4 * Hd, P.lstm_w_ih, E + Hd, E, X, xfloats, &ns, TARGET_WGS, st, nullptr, rl));
    {
        const size_t MN = (size_t)TB * E;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv((int)(MN / 4), 256)), dim3(256), 0, st, X, ns, MN, E, (const float*)nullptr, dEmb);
    }
    ICZ_CHECK_HIP(embed_grad_launch(st, tok, TB, dEmb, 1, (size_t)0, temb, cur_train ? 2.0f : 1.0f, E, G.embed_weight, V, 1, rl));
    // ---- weight gradients: one TN GEMM each over all (t, b) rows
    // (round 5) products that share d y go as one launch over column groups where the shape is taken (gemm_big_x3.hip)
    const GemmColGroup lstm_groups[3] = {{temb, E, E, G.lstm_w_ih, E + Hd}, {tu, Hd, Hd, G.lstm_w_ih + E, E + Hd}, {th, Hd, Hd, G.lstm_w_hh, Hd}};
    if (gemm_tn_grouped_fits(4 * Hd, TB, lstm_groups, 3)) {
        ICZ_TRY(gemm_tn_grouped(dG, 4 * Hd, 4 * Hd, TB, lstm_groups, 3, rl, st));
    } else {
        ICZ_TRY(tn(dG, 4 * Hd, 4 * Hd, temb, E, E, TB, G.lstm_w_ih, E + Hd, 0, st, rl));
        ICZ_TRY(tn(dG, 4 * Hd, 4 * Hd, tu, Hd, Hd, TB, G.lstm_w_ih + E, E + Hd, 0, st, rl));
        ICZ_TRY(tn(dG, 4 * Hd, 4 * Hd, th, Hd, Hd, TB, G.lstm_w_hh, Hd, 0, st, rl));
    }
    ICZ_TRY(colsum(dG, TB, 4 * Hd, 4 * Hd, G.lstm_b_ih, st));
    ICZ_CHECK_HIP(hipMemcpyAsync(G.lstm_b_hh, G.lstm_b_ih, sizeof(float) * 4 * Hd, hipMemcpyDeviceToDevice, st));
    if (grad_cb) grad_cb(grad_cb_user, 1);      // embed + lstm.*: reduced beside the attention block's weight gradients
    const GemmColGroup aoa_groups[2] = {{txatt, Hd, Hd, G.dec.aoa_w, 2 * Hd}, {tqn, Hd, Hd, G.dec.aoa_w + Hd, 2 * Hd}};
    if (gemm_tn_grouped_fits(2 * Hd, TB, aoa_groups, 2)) {
        ICZ_TRY(gemm_tn_grouped(dZ, 2 * Hd, 2 * Hd, TB, aoa_groups, 2, rl, st));
    } else {
        ICZ_TRY(tn(dZ, 2 * Hd, 2 * Hd, txatt, Hd, Hd, TB, G.dec.aoa_w, 2 * Hd, 0, st, rl));
        ICZ_TRY(tn(dZ, 2 * Hd, 2 * Hd, tqn, Hd, Hd, TB, G.dec.aoa_w + Hd, 2 * Hd, 0, st, rl));
    }
    ICZ_TRY(colsum(dZ, TB, 2 * Hd, 2 * Hd, G.dec.aoa_b, st));
    // ---- the attention block's small products (three Hd x Hd weight gradients, d K / d V, seven column sums: 240 us of kernels that fill a
    //      fraction of the chip, eager timeline round 6) depend on the loop only: without a DP callback they run on the side stream beside
    //      the embedding / LSTM / AoA-linear gradients above -- forked at the loop's end, ISSUED last, joined here (as Butd::bptt's tail)
    hipStream_t ts = tail_side ? low_st : st;
    auto tail = [&]() -> int {
    ICZ_TRY(tn(dQp, Hd, Hd, tqn, Hd, Hd, TB, G.dec.q_w, Hd, 0, ts, rl));
    ICZ_TRY(colsum(dQp, TB, Hd, Hd, G.dec.q_b, ts));
    {
        const int tc_fit = (int)(48 * 1024 / (sizeof(float) * 2 * (R + dh))), tc = T < tc_fit ? T : tc_fit;
        hipLaunchKernelGGL(aoa_dkv_kernel, dim3(B, NH), dim3(256), sizeof(float) * (size_t)tc * 2 * (R + dh), ts, tdS, tPd, tQp, tdX, dKd, dVd, B, T,
                           tc, R, Hd, NH, region_rows());
    }
    const int rrows = (int)region_row_count(B);      // region rows of the batch (packed valid rows with per-image counts)
    ICZ_TRY(tn(dKd, Hd, Hd, refined, Hd, Hd, rrows, G.dec.k_w, Hd, 0, ts));
    ICZ_TRY(colsum(dKd, rrows, Hd, Hd, G.dec.k_b, ts));
    ICZ_TRY(tn(dVd, Hd, Hd, refined, Hd, Hd, rrows, G.dec.v_w, Hd, 0, ts));
    ICZ_TRY(colsum(dVd, rrows, Hd, Hd, G.dec.v_b, ts));
    // ---- h_norm gain / bias
    {
        const size_t n = (size_t)TB * Hd;
        hipLaunchKernelGGL(aoa_ln_prod_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ts, dQn, th + sH, tstats, prod, (size_t)TB, Hd);
        ICZ_TRY(colsum(prod, TB, Hd, Hd, G.dec.ln_g, ts));
        ICZ_TRY(colsum(dQn, TB, Hd, Hd, G.dec.ln_b, ts));
    }
    return ICZ_OK;
    };
    if (tail_side) ICZ_CHECK_HIP(hipStreamWaitEvent(low_st, ev_fork3, 0));
    const int s_tail = tail();
    if (tail_side) {       // joined also on an error (inside a capture an unjoined side stream would hide it behind a capture failure)
        ICZ_CHECK_HIP(hipEventRecord(ev_join3, low_st));
        ICZ_CHECK_HIP(hipStreamWaitEvent(st, ev_join3, 0));
    }
    if (s_tail != ICZ_OK) return s_tail;
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}

}  // namespace icz

// ================================================================================================
using namespace icz;
extern "C" {

int icz_aoa_sample(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, const icz_aoa_rng* rng, int64_t* seq_out,
                   float* logprobs_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->sample(feats, B, max_len, rng, seq_out, logprobs_out, (hipStream_t)stream);
}
int icz_aoa_scst_rollouts(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, const icz_aoa_rng* rng, int64_t* ids_out,
                          int64_t* seq_out, float* logprobs_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->rollouts(feats, B, max_len, rng, ids_out, seq_out, logprobs_out, (hipStream_t)stream);
}
int icz_aoa_sample_backward(icz_aoa_t* h, const float* reward, const icz_aoa_params* grads, float* loss_out, float* mask_sum_out,
                            float mask_sum_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->sample_backward(reward, grads, loss_out, mask_sum_out, mask_sum_global, (hipStream_t)stream);
}
int icz_aoa_set_scheduled_sampling(icz_aoa_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms) {
    ICZ_REQUIRE(h, "null handle");
    ICZ_REQUIRE(ss_prob >= 0.f && ss_prob <= 1.f, "icz_aoa_set_scheduled_sampling: ss_prob %g outside [0, 1]", (double)ss_prob);
    Aoa* a = reinterpret_cast<Aoa*>(h);
    a->ss_prob = ss_prob; a->ss_gate = gate_uniforms; a->ss_draw = draw_uniforms;
    return ICZ_OK;
}
int icz_aoa_xe_forward(icz_aoa_t* h, const float* feats, const int64_t* captions, int32_t B, int32_t L, const int32_t* lengths_host,
                       const icz_aoa_rng* rng, int32_t train, float* packed_logits_out, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->xe_forward(feats, captions, B, L, lengths_host, rng, train, packed_logits_out, (hipStream_t)stream);
}
int icz_aoa_saved_alphas(icz_aoa_t* h, float* alphas_out, void* stream) {
    ICZ_REQUIRE(h && alphas_out, "icz_aoa_saved_alphas: null argument");
    Aoa* a = reinterpret_cast<Aoa*>(h);
    ICZ_REQUIRE(a->mode != 0 && a->tP, "icz_aoa_saved_alphas: no forward pass stored");
    const int n = a->cur_B * a->cur_T * a->cur_R;
    hipLaunchKernelGGL(saved_alphas_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, a->tP, a->cur_T, a->cur_B, a->dims.NH, a->cur_R, alphas_out);
    ICZ_CHECK_HIP(hipGetLastError());
    return ICZ_OK;
}
int icz_aoa_set_grad_callback(icz_aoa_t* h, icz_grad_ready_cb cb, void* user) {
    ICZ_REQUIRE(h, "null handle");
    Aoa* a = reinterpret_cast<Aoa*>(h);
    a->grad_cb = cb; a->grad_cb_user = user;
    return ICZ_OK;
}
int icz_aoa_set_norm_global(icz_aoa_t* h, const float* norm_dev, void* stream) {
    ICZ_REQUIRE(h && norm_dev, "icz_aoa_set_norm_global: null argument");
    ICZ_CHECK_HIP(hipMemcpyAsync(reinterpret_cast<Aoa*>(h)->d_msum, norm_dev, sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ICZ_OK;
}
int icz_aoa_xe_backward(icz_aoa_t* h, float smoothing, const icz_aoa_params* grads, float* loss_out, float n_tokens_global, void* stream) {
    ICZ_REQUIRE(h, "null handle");
    return reinterpret_cast<Aoa*>(h)->xe_backward(smoothing, grads, loss_out, n_tokens_global, (hipStream_t)stream);
}

}  // extern "C"
