// Internal declarations of the BUTD decoder handle.
#pragma once
#include <vector>

#include "butd_kernels.h"
#include "gemm_f32.h"

namespace icz {

// Inputs / outputs of one decoder step.  Null outputs fall back to the handle's scratch buffers.
struct StepIO {
    int rows;
    const float* feats;              // [n_img, R, D]
    const int32_t* img_of_row;       // beam search: image of each decoder row; null = identity
    const int64_t* it;               // [rows] input token ids
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
    DropCfg drop_emb, drop_att, drop_out;
};

struct Butd {
    static constexpr int TARGET_WGS = 512;   // ~2 workgroups per CU on 256 CUs
    icz_butd_dims dims;
    icz_butd_params P;
    bool bound = false, fresh = false;
    std::vector<void*> allocs;

    // weight-normed weights (w = g v / ||v||) and the row norms ||v||
    float *w_enc = nullptr, *w_dec = nullptr, *w_aff = nullptr, *w_pred = nullptr;
    float *n_enc = nullptr, *n_dec = nullptr, *n_aff = nullptr, *n_pred = nullptr;
    // per-image hoisted tensors
    float *mean = nullptr, *premean = nullptr, *enc_ctx = nullptr;
    // recurrent state (double buffered) and per-step scratch
    float *h1[2], *c1[2], *h2[2], *c2[2];
    float *emb = nullptr, *ctx = nullptr, *scores = nullptr, *alpha = nullptr, *h2drop = nullptr, *logits = nullptr;
    int64_t* it = nullptr;
    float* ws = nullptr;
    size_t ws_floats = 0;

    ~Butd();
    int alloc(void** p, size_t bytes);
    int init(const icz_butd_dims& d);
    int refresh(hipStream_t st);
    int gemm_nt(GemmArgs& g, int* nsplit_out, hipStream_t st);
    int prologue(const float* feats, int n_img, hipStream_t st);
    int step(const StepIO& s, hipStream_t st);
    int zero_state(int rows, int which, hipStream_t st);
    int greedy(const float* feats, int B, int max_len, int64_t* ids_out, float* alphas_out, hipStream_t st);
};

}  // namespace icz
