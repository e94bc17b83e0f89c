// AoADetection captioner (Models/AoA_Model.py:657-753): feature projection + 6-layer AoA refiner (once per image) and
// the AoA decoder (LSTM + LayerNorm + 8-head attention over the 36 refined regions + GLU gate) -- greedy / sampled /
// beam decoding and teacher-forced XE, with BPTT for the decoder parameters (the only ones the reference optimises,
// AoA_Model.py:669-674).  linear_K / linear_V of the decoder block are hoisted out of the time loop (the reference
// recomputes them every step, :114-115).
#pragma once
#include <math.h>

#include <vector>

#include "aoa_kernels.h"
#include "beam_kernels.h"
#include "butd_impl.h"

namespace icz {

enum AoaRngStream : uint32_t { AOA_RNG_PROJ = 10, AOA_RNG_REF_ATT = 11, AOA_RNG_REF_AOA = 12, AOA_RNG_REF_SC = 13,
                               AOA_RNG_CTX = 20, AOA_RNG_ATT = 21, AOA_RNG_OUT = 22 };

struct AoaStepIO {
    int rows;
    const int32_t* img_of_row;      // null = identity (row b decodes image b)
    const int64_t* it;
    bool emb_ready;
    const float *h_in, *m_in, *ctx_in;
    float *h_out, *m_out, *ctx_out;
    // per-step tensors; training passes slots of the saved [T,B,...] buffers, inference the scratch ones
    float *emb, *u, *gates_out, *ln_stats, *qn, *Qp, *P_out, *Pd_out, *xatt, *z_out, *ctxdrop, *logits;
    DropCfg d_emb;                   // embedding dropout (p = 0.5) through the BUTD embedding kernel
    DropP d_ctx, d_att, d_out;
    bool u_ready;                    // s.u was written by the previous step's GLU kernel (skip aoa_u_kernel)
    float* u_next;                   // where this step's GLU kernel leaves the next step's u (null: it does not)
    DropP d_ctx_next;                // the next step's ctx dropout
    int* pred_nsplit;                // non-null: the caller's consumer sums split-K slabs of the predict GEMM (gemm_predict)
    bool skip_predict;               // teacher-forced XE forward: the vocabulary projection of all time steps is one GEMM after the loop
    const int* live;                 // rollouts: the count of unfinished rows after the previous step; 0 = every kernel of this step returns at
                                     // entry (step_dead, icz_common.h: the reference's break, AoA_Model.py:400)
};

struct Aoa {
    static constexpr int STEP_WGS = 256, TARGET_WGS = 512, ARGMAX_PARTS = 8, NL = 6;
    icz_aoa_dims dims;
    icz_aoa_params P;
    bool bound = false, fresh = false;
    std::vector<void*> allocs;
    int Vp = 0;
    float *w_pred = nullptr, *n_pred = nullptr, *zeros = nullptr;
    float* w_qkv[NL] = {}; float* b_qkv[NL] = {};       // per refiner layer [3Hd, Hd] / [3Hd]: linear_Q | linear_K | linear_V (refresh)
    float* w_rec = nullptr;          // [4Hd, 2Hd] = [W_ih[:, E:] | W_hh]: one dgrad GEMM per BPTT step for (du, dh_prev)
    // refiner scratch / per-image tensors (rows = max_rows * R).  Two banks: bank 0 serves the evaluation-mode paths (greedy,
    // beam), bank 1 the training-mode ones (sample, XE, backward), so that the greedy baseline and the sampled rollout of
    // one SCST step can be in flight together; use_bank() points the members below at a bank before a chain is enqueued.
    struct Bank { float *xa, *xb, *ln, *qkv, *o, *od, *nd, *z, *refined, *meanf, *Kd, *Vd, *ws; int32_t *off, *rowmap; float* featp; };
    Bank bank[2] = {};
    int cur_bank = 0;
    // (round 5) the two refiner passes of an SCST step as ONE pass over [evaluation rows; training rows] (refine_pair): buffers of twice the
    // rows; while a pair is current, the banks' result pointers (refined, meanf, Kd, Vd) point at its halves, own[] keeps the banks' own
    Bank dual = {};
    Bank own[2] = {};
    bool pair_refine = true;             // icz_aoa_set_option("refine_pair")
    void point_banks_at_pair(int n_img) {
        const size_t nel = (size_t)n_img * cur_R * dims.Hd, nm = (size_t)n_img * dims.Hd;
        bank[0].refined = dual.refined; bank[0].meanf = dual.meanf; bank[0].Kd = dual.Kd; bank[0].Vd = dual.Vd;
        bank[1].refined = dual.refined + nel; bank[1].meanf = dual.meanf + nm; bank[1].Kd = dual.Kd + nel; bank[1].Vd = dual.Vd + nel;
        use_bank(cur_bank);
    }
    void point_bank_at_own(int b) {
        bank[b].refined = own[b].refined; bank[b].meanf = own[b].meanf; bank[b].Kd = own[b].Kd; bank[b].Vd = own[b].Vd;
        use_bank(cur_bank);
    }
    void use_bank(int b) {
        cur_bank = b;
        const Bank& s = bank[b];
        xa = s.xa; xb = s.xb; ln = s.ln; qkv = s.qkv; o = s.o; od = s.od; nd = s.nd; z = s.z;
        refined = s.refined; meanf = s.meanf; Kd = s.Kd; Vd = s.Vd; ws = s.ws; off = s.off; rowmap = s.rowmap;
    }
    // hipGraph replay of the SCST rollout pair and of the REINFORCE backward pass (option "graphs"; fixed region counts and Philox
    // randomness only: an 'adaptive' batch changes its grid sizes, explicit mask arrays their addresses)
    GraphCache gc;
    bool use_graphs = false;
    float* proj_shared = nullptr;        // rollouts: img_feats_porjection(feats) before ReLU / dropout, computed ONCE for the evaluation-
                                         // mode pass of the greedy baseline and the training-mode pass of the sampled rollout
    hipStream_t side_st = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t low_st = nullptr;                         // backward: the predict layer's weight gradient beside the reverse-time loop (plain priority)
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    hipEvent_t ev_fork3 = nullptr, ev_join3 = nullptr;      // (round 6) the attention block's small products beside the weight-gradient GEMMs of bptt's tail
    float *xa = nullptr, *xb = nullptr, *ln = nullptr, *qkv = nullptr, *o = nullptr, *od = nullptr, *nd = nullptr,
          *z = nullptr, *refined = nullptr, *meanf = nullptr, *Kd = nullptr, *Vd = nullptr;
    // decoder state + scratch
    float *h[2], *m[2], *ctx[2];
    float *emb = nullptr, *u = nullptr, *qn = nullptr, *Qp = nullptr, *xatt = nullptr, *ctxdrop = nullptr, *logits = nullptr, *ws = nullptr;
    size_t ws_floats = 0;
    int64_t* it = nullptr;
    float* amax_val = nullptr; int* amax_idx = nullptr;
    uint64_t* d_seed = nullptr; float* d_msum = nullptr;
    icz_grad_ready_cb grad_cb = nullptr; void* grad_cb_user = nullptr;      // DP overlap hook (icz_aoa_set_grad_callback)
    float ss_prob = 0.f; const float* ss_gate = nullptr; const float* ss_draw = nullptr;      // scheduled sampling in xe_forward
    BeamBuf bm;
    // training buffers (aoa_train.hip), slot stride = max_rows: th/tm/tctx slot 0 = zeros, slot t+1 = after step t
    int tcap_B = 0, tcap_T = 0;          // capacity of the training buffers (grown on demand by ensure_train)
    std::vector<void*> tallocs; bool alloc_train = false;
    int64_t* tok = nullptr;
    float *th = nullptr, *tm = nullptr, *tctx = nullptr, *temb = nullptr, *tu = nullptr, *tg = nullptr, *tstats = nullptr, *tqn = nullptr,
          *tQp = nullptr, *tP = nullptr, *tPd = nullptr, *tdS = nullptr, *tdX = nullptr, *txatt = nullptr, *tz = nullptr, *tcd = nullptr, *tlogit = nullptr;
    float *dCd = nullptr, *dZ = nullptr, *dQp = nullptr, *dQn = nullptr, *dHln = nullptr, *dG = nullptr, *dEmb = nullptr, *dKd = nullptr,
          *dVd = nullptr, *dcb[2] = {nullptr, nullptr}, *X = nullptr, *X2 = nullptr, *dWp = nullptr, *prod = nullptr;
    float *coef = nullptr, *lse = nullptr, *loss_rows = nullptr;
    int32_t* draw = nullptr; uint8_t* unf = nullptr; int* nunf = nullptr; int* pack_idx = nullptr;
    uint8_t* gunf = nullptr; int* gnunf = nullptr; int* live_rows = nullptr;      // SCST baseline's counters; (steps the sampled rollout ran) x B
    bool early_out = true, bptt_early_out = false;
    size_t xfloats = 0;
    icz_aoa_rng rng = {};
    int mode = 0, cur_B = 0, cur_T = 0, cur_L = 0, n_tokens = 0;
    // regions per image of the current batch: row stride cur_R <= dims.R and, for the 'adaptive' bottom-up features
    // (10..100 boxes, AoA_Engine.py:37-44), the valid count per image (device, caller-owned; null = all cur_R)
    int cur_R = 0, lens_n = 0, cur_total = 0;      // cur_total = sum of the counts = rows of the packed refiner tensors
    const int32_t* lens = nullptr;
    int32_t *off = nullptr, *rowmap = nullptr;     // per bank: row offsets / padded indices of the packed rows (RegionRows)
    RegionRows region_rows() const { return lens ? RegionRows{off, rowmap, lens, cur_R} : RegionRows{nullptr, nullptr, nullptr, cur_R}; }
    size_t region_row_count(int n_img) const { return lens ? (size_t)cur_total : (size_t)n_img * cur_R; }
    static constexpr size_t LDS_BUDGET = 156 * 1024;
    size_t self_lds(int R, int qc) const {       // mha_self_kernel: K, V [R4][ld] + Q chunk [qc4][ld] + P [qc4][lp]
        const size_t ld = aoa_pitch(dims.Hd / dims.NH), lp = aoa_pitch(R), R4 = (R + 3) & ~3, q4 = (qc + 3) & ~3;
        return sizeof(float) * (2 * R4 * ld + q4 * ld + q4 * lp);
    }
    int self_qc(int R) const {       // query rows per pass of mha_self_kernel: all of them when the tiles fit
        if (self_lds(R, R) <= LDS_BUDGET) return R;
        int qc = R & ~3;
        while (qc >= 4 && self_lds(R, qc) > LDS_BUDGET) qc -= 4;
        return qc >= 4 ? qc : 0;
    }
    bool cur_train = false;
    const int64_t* cur_seq = nullptr; const float* cur_logp = nullptr;
    const int64_t* cur_captions = nullptr;
    std::vector<int> rows_t;

    ~Aoa() {
        if (side_st) (void)hipStreamDestroy(side_st);
        if (low_st) (void)hipStreamDestroy(low_st);
        if (ev_fork2) (void)hipEventDestroy(ev_fork2);
        if (ev_join2) (void)hipEventDestroy(ev_join2);
        if (ev_fork3) (void)hipEventDestroy(ev_fork3);
        if (ev_join3) (void)hipEventDestroy(ev_join3);
        if (ev_fork) (void)hipEventDestroy(ev_fork);
        if (ev_join) (void)hipEventDestroy(ev_join);
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
    int init(const icz_aoa_dims& d);
    int refresh(hipStream_t st);
    // out[M,N] = A[M,K] W[N,K]^T + bias  (split-K through `ws` when the launch would be too small)
    int lin(const float* A, int M, int K, const float* W, const float* bias, int N, float* out, hipStream_t st);
    int refine(const float* feats, int n_img, bool train, hipStream_t st, const float* proj = nullptr);
    int refine_pair(int n_img, hipStream_t st, const float* proj);
    bool mha_mfma = true;                // icz_aoa_set_option("mha_mfma"): refiner self-attention on the fp32 matrix pipe (<= 64 regions)
    void launch_mha_self(int n_img, int R, int qc, size_t lds, const RegionRows& rr, const float* qkv_, float* o_, const DropP& dp, hipStream_t st);
    int project(const float* feats, int n_img, float* out, hipStream_t st);
    int step(const AoaStepIO& s, hipStream_t st);
    int greedy(const float* feats, int B, int T, int64_t* ids_out, hipStream_t st, const float* proj = nullptr, bool scst = false, bool refined_ready = false);
    int rollouts_impl(const float* feats, int B, int T, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st);
    int rollouts(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* ids_out, int64_t* seq_out, float* logp_out, hipStream_t st);
    int beam_search(const float* feats, int n_img, int kb, int max_steps, float* seqs_out, int32_t* lens_out, hipStream_t st);
    DropP dropp(bool train, const uint8_t* mask, size_t off, uint32_t stream, int step, float p) const {
        DropP d = {0, nullptr, d_seed, stream, (uint32_t)step, (uint32_t)((double)p * 4294967296.0), 1.0f / (1.0f - p)};
        if (!train) return d;
        if (mask) { d.mode = 1; d.mask = mask + off; } else d.mode = 2;
        return d;
    }
    DropCfg dropbits(bool train, const uint8_t* mask, size_t off, uint32_t stream, int step) const {
        DropCfg d = {0, nullptr, d_seed, stream, (uint32_t)step};
        if (!train) return d;
        if (mask) { d.mode = 1; d.mask = mask + off; } else d.mode = 2;
        return d;
    }
    // training paths (aoa_train.hip)
    int ensure_train(int B, int T);
    AoaStepIO train_io(int rows, int t, bool train);
    int sample(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st);
    int sample_prelude(const float* feats, int B, int T, const icz_aoa_rng* r, int64_t* seq_out, float* logp_out, hipStream_t st);
    int sample_impl(const float* feats, int B, int T, int64_t* seq_out, float* logp_out, hipStream_t st, const float* proj = nullptr, bool refined_ready = false);
    int sample_backward_impl(const float* reward, const icz_aoa_params& G, float* loss_out, float* msum_out, hipStream_t st);
    int sample_backward(const float* reward, const icz_aoa_params* G, float* loss_out, float* msum_out, float msum_global, hipStream_t st);
    int xe_forward(const float* feats, const int64_t* captions, int B, int L, const int32_t* lengths, const icz_aoa_rng* r, int train,
                   float* packed_out, hipStream_t st);
    int xe_backward(float smoothing, const icz_aoa_params* G, float* loss_out, float n_tokens_global, hipStream_t st);
    int bptt(const icz_aoa_params& G, hipStream_t st);
    int colsum(const float* Xm, int K, int N, int ldx, float* out, hipStream_t st);
    int nn(const float* A, int lda, int M, int K, const float* Bm, int ldb, int N, float* slab_out, size_t cap, int* ns, int target, hipStream_t st,
           const int* live = nullptr, const int* rows_live = nullptr);
    int tn(const float* dY, int ldy, int M, const float* Xm, int ldx, int N, int K, float* out, int ldo, int accumulate, hipStream_t st,
           const int* rows_live = nullptr);
};

}  // namespace icz
