/* libicz -- C ABI of the MI355X-native caption-decoding hot path (BUTD / SCST).
 *
 * The reference (zyj0021200/simpleImageCaptionZoo) is pure Python on PyTorch and has no FFI of its own; its
 * extension point for this path is the duck-typed Captioner nn.Module that Engine calls
 * (Engine.py:179-182, 258-262, 284-286) plus the loss / reward helpers in Utils.py.  Each entry point below
 * names the reference function it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions: every pointer is a DEVICE pointer unless the name ends in _host; tensors are dense row-major
 * fp32 unless stated; ids are int64 (torch.LongTensor); `stream` is a hipStream_t passed as void* (0 = null
 * stream).  Calls are asynchronous on `stream`; nothing synchronises the device unless documented.  The
 * library never frees caller memory.  Return value: ICZ_OK (0) or a negative icz_status; icz_last_error()
 * gives the text (thread-local).  A handle is not thread-safe; distinct handles are independent.
 */
#ifndef ICZ_H_
#define ICZ_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ICZ_OK = 0,
    ICZ_ERR_INVALID = -1,   /* bad argument / shape / alignment */
    ICZ_ERR_HIP = -2,       /* a HIP runtime call failed        */
    ICZ_ERR_STATE = -3,     /* call order violated (e.g. backward before sample) */
    ICZ_ERR_NOMEM = -4
} icz_status;

const char* icz_last_error(void);
/* library version and the gfx target it was built for ("gfx950") */
const char* icz_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * BUTD top-down attention decoder (Models/BUTD_Model.py:64-318, DecoderRNN + SoftAttention)
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct icz_butd icz_butd_t;

typedef struct {
    int32_t R;        /* regions per image (36 bottom-up boxes, 49 for the 7x7 spatial grid) */
    int32_t D;        /* region feature size (2048)                                          */
    int32_t H;        /* LSTM hidden size (Configs/Models/BUTDDetection.json: 1024)          */
    int32_t E;        /* embedding size                                                      */
    int32_t A;        /* attention size                                                      */
    int32_t V;        /* vocabulary size                                                     */
    int32_t max_rows; /* capacity in decoder rows: batch for greedy/sample/XE, images*beam for beam search */
    int32_t max_len;  /* capacity in time steps kept for backward (>= max_len of sample / XE length)        */
} icz_butd_dims;

/* Parameter block in the reference's state_dict layout (keys "decoder.<name>", BUTD_Model.py:75-84;
 * weight_norm'd Linear layers are the old-style (weight_g, weight_v) pairs).  The same struct type carries
 * gradients (one buffer per parameter, same shapes). */
typedef struct {
    float* embed_weight;                          /* embed.0.weight          [V, E]            */
    float *td_w_ih, *td_w_hh, *td_b_ih, *td_b_hh; /* TD_atten.*              [4H, H+D+E] [4H, H] [4H] [4H] */
    float *lm_w_ih, *lm_w_hh, *lm_b_ih, *lm_b_hh; /* language_model.*        [4H, D+H]   [4H, H] [4H] [4H] */
    float *enc_att_v, *enc_att_g, *enc_att_b;     /* atten.enc_att.weight_v [A, D], weight_g [A,1], bias [A] */
    float *dec_att_v, *dec_att_g, *dec_att_b;     /* atten.dec_att.*        [A, H], [A,1], [A]               */
    float *affine_v, *affine_g, *affine_b;        /* atten.affine.*         [1, A], [1,1], [1]               */
    float *predict_v, *predict_g, *predict_b;     /* predict.*              [V, H], [V,1], [V]               */
} icz_butd_params;

int icz_butd_create(const icz_butd_dims* dims, icz_butd_t** out);
int icz_butd_destroy(icz_butd_t* h);
/* Bind the (caller-owned, device-resident) parameters; pointers must stay valid while the handle is used. */
int icz_butd_bind_params(icz_butd_t* h, const icz_butd_params* params);
/* Options.  "graphs" = 1: greedy / sample / sample_backward are captured into hipGraphs on first use and replayed
 * afterwards; the cache is keyed by every pointer and size in the call, so enable it only when buffers are reused.
 * "concurrent" = 0: independent chains (greedy vs sampled rollout, predict gradients vs BPTT) run on ONE stream instead
 * of side streams (default 1); used by bench.py to time single kernels with events.
 * "early_out" (default 1, ICZ_EARLY_OUT): DecoderRNN.sample_rl breaks out of its loop once no row is unfinished
 * (BUTD_Model.py:233); on the device every kernel of a rollout step behind that point -- and of its BPTT step -- returns at entry
 * (a per-step count of unfinished rows in device memory, written by the token-choice kernel), the GEMMs over all (t, b) rows stop
 * there, and the greedy baseline of icz_butd_scst_rollouts stops once EVERY row has emitted <end>.  0 = run every step as rounds
 * 1 - 4 did: the same results, an A/B switch.
 * "merge_small" = n (default 8, 0..32, ICZ_MERGE_SMALL): icz_butd_scst_rollouts of <= n images runs the greedy baseline and the
 * sampled rollout as ONE chain of 2 B decoder rows (evaluation-mode rows in front: per-row dropout / argmax-vs-multinomial in the
 * kernels), so that the weights are streamed once per step pair; same tokens, log-probs, loss as the two chains.
 * "small_nt" (default 1): BPTT steps of <= 32 rows (small batches, the short tail of an XE batch) take their per-step dgrad products on
 * the transposed LSTM weight copies through the fp32 NT kernel instead of NN products on the weights (which stream at half the rate
 * at so few rows); 0 = the NN products of rounds 1 - 4.  Gradients agree to fp32 rounding. */
int icz_butd_set_option(icz_butd_t* h, const char* name, int32_t value);
/* Data-parallel overlap hook (no reference counterpart: the reference is single-process).  While a backward call is
 * being enqueued, `cb(user, stage)` is invoked each time a group of gradient tensors is complete in stream order:
 *   stage 0: predict.{weight_v, weight_g, bias};  stage 1: embed.0.weight, TD_atten.weight_{ih,hh};
 *   stage 2: language_model.weight_{ih,hh};  everything else is complete when the call returns.
 * The host side starts the all-reduce of that group on the same stream (it then runs beside the remaining weight-gradient
 * GEMMs).  NULL removes the hook. */
/* Data-parallel: the global (all-reduced) loss normaliser as a DEVICE scalar, so that no host round trip separates the
 * forward pass from the backward pass: the mask sum of the rollout (then call icz_butd_sample_backward with
 * mask_sum_global < 0, "use the device value") or the token count of the XE batch (icz_butd_xe_backward with
 * n_tokens_global < 0).  icz_nic_set_norm_global / icz_aoa_set_norm_global are the same for the other two decoders. */
int icz_butd_set_mask_sum_global(icz_butd_t* h, const float* mask_sum_global_dev, void* stream);
typedef void (*icz_grad_ready_cb)(void* user, int32_t stage);
int icz_butd_set_grad_callback(icz_butd_t* h, icz_grad_ready_cb cb, void* user);
/* Re-materialise w = g * v / ||v|| for the four weight-normed layers; call after every parameter update.  The transposed copies of
 * the LSTM weights that the per-step dgrad products of BPTT read (117 MB at the BASELINE sizes, 33 us) are NOT rebuilt here but by the
 * first backward call after it (icz_butd_sample_backward / icz_butd_xe_backward*, in front of their captured graph): evaluation
 * loops refresh per batch and never need them. */
int icz_butd_refresh_weights(icz_butd_t* h, void* stream);

/* DecoderRNN.sample (BUTD_Model.py:153-189): greedy decode.
 * feats [B,R,D]; ids_out [B,max_len] int64; alphas_out [B,max_len,R] or NULL. */
int icz_butd_greedy(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, int64_t* ids_out,
                    float* alphas_out, void* stream);

/* Randomness for the training-mode paths.  Either explicit arrays (parity tests; layouts below) or, for any
 * NULL pointer, an in-kernel counter-based generator (Philox4x32-10) keyed by `seed` -- reproduced
 * bit-for-bit by the matching backward call, so nothing is stored. */
typedef struct {
    uint64_t seed;
    const float* uniforms;     /* [T, B]        one draw per row per step (multinomial), in [0,1) */
    const uint8_t* emb_mask;   /* [T, B, E]     keep-masks (1 = keep) of nn.Dropout(0.5) on the embedding, */
    const uint8_t* att_mask;   /* [T, B, R, A]  on relu(enc_ctx + dec_ctx) in SoftAttention,               */
    const uint8_t* out_mask;   /* [T, B, H]     and on h2 before `predict`                                 */
} icz_rng;

/* DecoderRNN.sample_rl (BUTD_Model.py:191-234), training mode (dropout active).
 * seq_out [B,max_len] int64 (0 at and after a sampled <end>), logprobs_out [B,max_len].  Activations needed by
 * icz_butd_sample_backward are kept inside the handle until the next sample/XE call. */
int icz_butd_sample(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, const icz_rng* rng,
                    int64_t* seq_out, float* logprobs_out, void* stream);

/* Both rollouts of one SCST step (Engine.py:258-262: greedy baseline in eval mode + sampled rollout in train mode)
 * in one call; the two decode chains share the per-image prologue and run concurrently on two streams.  Results are
 * identical to icz_butd_greedy + icz_butd_sample. */
/* (greedy_ids_out: the reference's greedy ids up to and including every row's first <end> -- all the reward reads, Utils.py:354;
 * columns behind the step at which the LAST row emitted it are 0, see option "early_out".) */
int icz_butd_scst_rollouts(icz_butd_t* h, const float* feats, int32_t B, int32_t max_len, const icz_rng* rng,
                           int64_t* greedy_ids_out, int64_t* seq_out, float* logprobs_out, void* stream);

/* RewardCriterion.forward + loss.backward() for the rollout kept by the last icz_butd_sample
 * (Utils.py:295-317, Engine.py:266-270).  reward [B,max_len]; grads receives d loss / d parameter (overwritten,
 * not accumulated); loss_out (1 float, device) receives the loss.
 * Data-parallel use: pass mask_sum_global > 0 to normalise by the all-rank sum of the mask instead of the
 * local one; mask_sum_out (1 float, device, may be NULL) always receives the local mask sum. */
int icz_butd_sample_backward(icz_butd_t* h, const float* reward, const icz_butd_params* grads, float* loss_out,
                             float* mask_sum_out, float mask_sum_global, void* stream);
/* Same backward, driven by an arbitrary upstream gradient d loss / d logprobs [B,max_len] (what autograd hands to
 * the sampler_rl output when the reference's own RewardCriterion + loss.backward() are used, Engine.py:266-270). */
int icz_butd_sample_backward_dlogp(icz_butd_t* h, const float* dlogp, const icz_butd_params* grads, void* stream);
/* local sum of the REINFORCE mask of the last rollout (1 float, device) without running backward */
int icz_butd_sample_mask_sum(icz_butd_t* h, float* mask_sum_out, void* stream);

/* DecoderRNN.forward + LabelSmoothingLoss + backward (BUTD_Model.py:97-151, Utils.py:268-286,
 * Engine.py:178-186).  captions [B,L] int64, rows sorted by length descending; lengths_host[b] = len-1 as in
 * Engine.py:178 (host array).  packed_logits_out [sum(lengths), V] in pack_padded_sequence order or NULL.
 * n_tokens_global > 0 replaces the local token count as the loss normaliser (data-parallel); < 0: the device scalar set by
 * icz_butd_set_mask_sum_global. */
int icz_butd_xe_forward(icz_butd_t* h, const float* feats, const int64_t* captions, int32_t B, int32_t L,
                        const int32_t* lengths_host, const icz_rng* rng, int32_t train,
                        float* packed_logits_out, void* stream);
int icz_butd_xe_backward(icz_butd_t* h, float smoothing, const icz_butd_params* grads, float* loss_out,
                         float n_tokens_global, void* stream);

/* Scheduled sampling in DecoderRNN.forward (BUTD_Model.py:120-132; the decoder attribute `ss_prob`, which
 * Engine.py:140-144 means to raise per epoch but sets on the Captioner, where nothing reads it): from time step 2 on,
 * row b of the following icz_butd_xe_forward calls feeds a draw from softmax(logits of step t-1) instead of its caption
 * token when gate[t,b] < ss_prob.  gate_uniforms / draw_uniforms: device arrays [T,B] in [0,1) (T = max(lengths), B of
 * the forward call; parity tests) or NULL for the Philox generator keyed by the call's icz_rng seed.  The draw is the
 * inverse-CDF draw of the sampler contract (smallest v with cumsum(p)[v] > u sum(p), float64 sums); no gradient
 * flows through it and the embedding gradient goes to the tokens actually fed.  ss_prob = 0 (default) switches it off. */
int icz_butd_set_scheduled_sampling(icz_butd_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms);

/* Attention maps of the forward pass the handle holds (the last icz_butd_xe_forward or icz_butd_sample of B rows and T
 * steps): alphas_out [B, T, R].  A teacher-forced evaluation-mode forward over a decoded sentence yields the alphas the
 * reference's sample / beam_search_sample return beside the ids (BUTD_Model.py:178-189, :309-317), which
 * eval_test_image hands to show_additional_rlt (Engine.py:325,339). */
int icz_butd_saved_alphas(icz_butd_t* h, float* alphas_out, void* stream);

/* XE backward driven by an upstream gradient w.r.t. the packed logits [sum(lengths), V] (autograd path:
 * criterion(predictions[0], targets[0]).backward(), Engine.py:182-186). */
int icz_butd_xe_backward_dlogits(icz_butd_t* h, const float* dpacked, const icz_butd_params* grads, void* stream);

/* DecoderRNN.beam_search_sample (BUTD_Model.py:236-318) for n_img images at once (the reference runs one image
 * per call, Utils.py:72-73).  feats [n_img,R,D]; seqs_out [n_img, max_steps+1] float32 incl. the leading <sta>
 * and, if finished, the trailing <end>; lens_out [n_img] int32 = valid length of each row. */
int icz_butd_beam_search(icz_butd_t* h, const float* feats, int32_t n_img, int32_t beam, int32_t max_steps,
                         float* seqs_out, int32_t* lens_out, void* stream);

/* One decoder step from an explicit state (BUTD_Model.py:172-182) -- exposed for the per-kernel parity tests.
 * it [B] int64; state tensors [B,H] are updated in place; ctx_out [B,D], alpha_out [B,R], logits_out [B,V]. */
int icz_butd_step(icz_butd_t* h, const float* feats, int32_t B, const int64_t* it, float* h1, float* c1,
                  float* h2, float* c2, float* ctx_out, float* alpha_out, float* logits_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * NIC decoder (Models/NIC_Model.py:39-212, DecoderRNN): plain embedding -> one LSTMCell -> predict; the image
 * embedding `features` [B,E] (output of the CNN encoder, outside the hot path) enters through one LSTM step from the
 * zero state (:52-56).  Same conventions as the BUTD entry points; of icz_rng only `uniforms` and `out_mask` are used.
 * dfeatures_out [B,E] (may be NULL) receives d loss / d features for an upstream encoder.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct icz_nic icz_nic_t;
typedef struct { int32_t E, H, V, max_rows, max_len; } icz_nic_dims;
typedef struct {
    float* embed_weight;                  /* embed.weight   [V, E]                         */
    float *w_ih, *w_hh, *b_ih, *b_hh;     /* lstm.*         [4H, E] [4H, H] [4H] [4H]      */
    float *predict_v, *predict_g, *predict_b;   /* predict.weight_v [V,H], weight_g [V,1], bias [V] */
} icz_nic_params;
int icz_nic_create(const icz_nic_dims* dims, icz_nic_t** out);
int icz_nic_destroy(icz_nic_t* h);
int icz_nic_bind_params(icz_nic_t* h, const icz_nic_params* params);
int icz_nic_refresh_weights(icz_nic_t* h, void* stream);
/* DecoderRNN.sample, NIC_Model.py:100-119 */
int icz_nic_greedy(icz_nic_t* h, const float* features, int32_t B, int32_t max_len, int64_t* ids_out, void* stream);
/* DecoderRNN.sample_rl, NIC_Model.py:121-151, and RewardCriterion + backward (Utils.py:295-317) */
int icz_nic_sample(icz_nic_t* h, const float* features, int32_t B, int32_t max_len, const icz_rng* rng, int64_t* seq_out,
                   float* logprobs_out, void* stream);
int icz_nic_sample_backward(icz_nic_t* h, const float* reward, const icz_nic_params* grads, float* dfeatures_out, float* loss_out,
                            float* mask_sum_out, float mask_sum_global, void* stream);
/* DecoderRNN.forward, NIC_Model.py:58-98, and LabelSmoothingLoss + backward (Utils.py:268-286) */
int icz_nic_xe_forward(icz_nic_t* h, const float* features, const int64_t* captions, int32_t B, int32_t L, const int32_t* lengths_host,
                       const icz_rng* rng, int32_t train, float* packed_logits_out, void* stream);
int icz_nic_xe_backward(icz_nic_t* h, float smoothing, const icz_nic_params* grads, float* dfeatures_out, float* loss_out,
                        float n_tokens_global, void* stream);
int icz_nic_set_norm_global(icz_nic_t* h, const float* norm_dev, void* stream);
/* Option "early_out" (default 1; also icz_butd_set_option / icz_aoa_set_option): the kernels of the rollout / BPTT steps behind
 * sample_rl's break (NIC_Model.py:150: every row has finished) return at entry; 0 = run them as rounds 1 - 4 did (A/B switch: the
 * results are the same). */
int icz_nic_set_option(icz_nic_t* h, const char* name, int32_t value);
/* Scheduled sampling for the following icz_nic_xe_forward calls (NIC_Model.py:77-89): see icz_butd_set_scheduled_sampling. */
int icz_nic_set_scheduled_sampling(icz_nic_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms);
/* DecoderRNN.beam_search_sample, NIC_Model.py:153-212 (batched over images) */
int icz_nic_beam_search(icz_nic_t* h, const float* features, int32_t n_img, int32_t beam, int32_t max_steps, float* seqs_out,
                        int32_t* lens_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * AoADetection captioner (Models/AoA_Model.py:657-753): img_feats_porjection (Linear 2048->Hd + ReLU + Dropout) ->
 * AoA_Refine_Core (6 x {LayerNorm -> 8-head self-attention over the R regions -> GLU gate -> residual}, final LayerNorm;
 * :122-162) once per image, then AoA_Decoder (:197-502) per step: embed -> LSTMCell([emb, mean + drop(ctx)]) -> LayerNorm
 * -> 8-head attention over the refined regions + GLU -> predict.  Fixed region sets (36 boxes or the 7x7 grid,
 * bu_masks = None) and the 'adaptive' ones (10..100 boxes per image with prefix bu_masks, AoA_Engine.py:37-44) through
 * icz_aoa_set_regions.
 * Only decoder.* parameters receive gradients: they are the only ones in the reference's optimizer (:669-674).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct icz_aoa icz_aoa_t;
typedef struct { int32_t R, D, Hd, E, V, NH, max_rows, max_len; } icz_aoa_dims;
typedef struct {   /* one AoABlock + its LayerNorm: linear_Q/K/V [Hd,Hd]+[Hd], aoa_module.0 [2Hd,2Hd]+[2Hd], norm gain/bias [Hd] */
    float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *aoa_w, *aoa_b, *ln_g, *ln_b;
} icz_aoa_block;
typedef struct {
    float *proj_w, *proj_b;                        /* img_feats_porjection.0.{weight [Hd,D], bias [Hd]}              */
    icz_aoa_block layer[6];                        /* aoa_refine.aoa_layers.<l>.{aoa_block.*, sublayer.norm.*}        */
    float *ref_ln_g, *ref_ln_b;                    /* aoa_refine.norm.{gain, bias}                                    */
    float *lstm_w_ih, *lstm_w_hh, *lstm_b_ih, *lstm_b_hh;   /* decoder.lstm.*  [4Hd, E+Hd] [4Hd, Hd] [4Hd] [4Hd]      */
    icz_aoa_block dec;                             /* decoder.aoa_block.* and decoder.h_norm.{gain,bias} (ln_g, ln_b) */
    float* embed_weight;                           /* decoder.embed.0.weight [V, E]                                   */
    float *predict_v, *predict_g, *predict_b;      /* decoder.predict.*                                               */
} icz_aoa_params;
/* Randomness of the training-mode paths (explicit arrays for parity tests, Philox from `seed` for NULL pointers).
 * Keep-masks (1 = keep) with the reference's drop probabilities: proj 0.5 [B,R,Hd]; per refiner layer: ref_att 0.1
 * [6,B,NH,R,R], ref_aoa 0.3 [6,B,R,2Hd], ref_sc 0.1 [6,B,R,Hd]; per step: emb 0.5 [T,B,E], ctx 0.5 [T,B,Hd], att 0.1
 * [T,B,NH,R], out 0.5 [T,B,Hd]; uniforms [T,B]. */
typedef struct {
    uint64_t seed;
    const float* uniforms;
    const uint8_t *proj_mask, *ref_att_mask, *ref_aoa_mask, *ref_sc_mask, *emb_mask, *ctx_mask, *att_mask, *out_mask;
} icz_aoa_rng;
int icz_aoa_create(const icz_aoa_dims* dims, icz_aoa_t** out);
int icz_aoa_destroy(icz_aoa_t* h);
int icz_aoa_bind_params(icz_aoa_t* h, const icz_aoa_params* params);
int icz_aoa_refresh_weights(icz_aoa_t* h, void* stream);
/* Region layout of the batches that follow (sticky; a new handle has regions = dims.R and no counts): `feats` of every
 * later call is [B, regions, D] with regions <= dims.R (the handle's capacity).  counts = the number of valid leading
 * regions of each image, i.e. bu_masks.sum(1) of the reference's prefix masks (AoA_Engine.py:37-40), in device memory
 * (read by the kernels of the following calls -- keep it alive until they have run) and in host memory (validated here:
 * 1 <= counts[i] <= regions); both NULL = every region valid (bu_masks = None).  Masked keys get attention weight 0
 * (masked_fill(-1e9) before the softmax, AoA_Model.py:63-64), the projection runs on the valid rows only (pack_wrapper,
 * :650-653 -- here the whole refiner does: the padded rows the reference carries along never reach a result) and
 * mean_features averages the valid rows (:253). */
int icz_aoa_set_regions(icz_aoa_t* h, int32_t regions, const int32_t* counts_dev, const int32_t* counts_host, int32_t n_img);
/* Option "graphs" = 1 (round 5): icz_aoa_scst_rollouts and icz_aoa_sample_backward are captured into hipGraphs on first use and
 * replayed afterwards, as icz_butd_set_option("graphs") does for the BUTD decoder; keyed by every pointer and size of the call, so
 * enable it only when buffers are reused.  Batches with per-image region counts (icz_aoa_set_regions), explicit randomness arrays
 * and a backward pass under a gradient-ready callback stay eager.
 * Option "refine_pair" (default 1, round 5): icz_aoa_scst_rollouts runs the evaluation-mode refiner pass of the greedy baseline and the
 * training-mode pass of the sampled rollout (AoA_Model.py:698-714 under Engine.py:256-262) as ONE pass over twice the rows -- same
 * bits in both halves as the two passes when the GEMMs take the same split-K decomposition (they do at the BASELINE batch of 64;
 * otherwise within fp32 rounding); 0 = two passes.  Fixed region counts only (a batch under icz_aoa_set_regions runs two passes).
 * Option "mha_mfma" (default 1, round 5): the refiner's self-attention (AoA_Model.py:41-69) on the fp32 matrix pipe for up to 64
 * regions; 0 = the register-blocked kernel of rounds 1 - 4 (which larger region sets use anyway). */
int icz_aoa_set_option(icz_aoa_t* h, const char* name, int32_t value);
/* eval-mode refined features [B,regions,Hd] (AoADetection_Captioner.sampler's first two lines, :712-713) -- for tests.  With
 * region counts the refiner runs on the packed valid rows; rows past an image's count come back as zeros. */
int icz_aoa_refine(icz_aoa_t* h, const float* feats, int32_t B, float* refined_out, void* stream);
/* AoADetection_Captioner.sampler / beam_search_sampler / sampler_rl / forward (:698-753, :676-696) */
int icz_aoa_greedy(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, int64_t* ids_out, void* stream);
int icz_aoa_beam_search(icz_aoa_t* h, const float* feats, int32_t n_img, int32_t beam, int32_t max_steps, float* seqs_out,
                        int32_t* lens_out, void* stream);
int icz_aoa_sample(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, const icz_aoa_rng* rng, int64_t* seq_out,
                   float* logprobs_out, void* stream);
/* The two decodes of one SCST step (Engine.py:256-261: greedy in eval mode, sampler_rl in train mode) as concurrent chains. */
int icz_aoa_scst_rollouts(icz_aoa_t* h, const float* feats, int32_t B, int32_t max_len, const icz_aoa_rng* rng, int64_t* ids_out,
                          int64_t* seq_out, float* logprobs_out, void* stream);
int icz_aoa_sample_backward(icz_aoa_t* h, const float* reward, const icz_aoa_params* grads, float* loss_out, float* mask_sum_out,
                            float mask_sum_global, void* stream);
int icz_aoa_xe_forward(icz_aoa_t* h, const float* feats, const int64_t* captions, int32_t B, int32_t L, const int32_t* lengths_host,
                       const icz_aoa_rng* rng, int32_t train, float* packed_logits_out, void* stream);
int icz_aoa_xe_backward(icz_aoa_t* h, float smoothing, const icz_aoa_params* grads, float* loss_out, float n_tokens_global,
                        void* stream);
int icz_aoa_set_norm_global(icz_aoa_t* h, const float* norm_dev, void* stream);
/* Data-parallel overlap hook of the AoA decoder's backward passes (icz_aoa_sample_backward, icz_aoa_xe_backward), as
 * icz_butd_set_grad_callback: cb(user, stage) is called on the calling thread while the backward pass is being enqueued, each time
 * a group of the decoder's gradients (the only parameters in the reference's optimizer, AoA_Model.py:669-674) is complete in
 * stream order:
 *   stage 0: decoder.predict.{weight_v, weight_g, bias} -- before the reverse-time loop, so its all-reduce runs beside all of BPTT;
 *   stage 1: decoder.embed.0.weight, decoder.lstm.{weight_ih, weight_hh, bias_ih, bias_hh};
 * the attention block's and h_norm's gradients are complete when the call returns.  NULL removes the hook. */
int icz_aoa_set_grad_callback(icz_aoa_t* h, icz_grad_ready_cb cb, void* user);
/* As icz_butd_saved_alphas: the decoder block's attention weights averaged over the heads (AoA_Model.py:118), [B, T, regions]. */
int icz_aoa_saved_alphas(icz_aoa_t* h, float* alphas_out, void* stream);
/* Scheduled sampling for the following icz_aoa_xe_forward calls (AoA_Model.py:258-270): see icz_butd_set_scheduled_sampling. */
int icz_aoa_set_scheduled_sampling(icz_aoa_t* h, float ss_prob, const float* gate_uniforms, const float* draw_uniforms);

/* ------------------------------------------------------------------------------------------------------------
 * Optimiser step: clip_gradient (Utils.py:241-250, value clamp) + torch.optim.Adam(betas=(0.9,0.999),
 * eps=1e-8, weight_decay=0) (Utils.py:219-220) fused, one call per parameter tensor.
 * ---------------------------------------------------------------------------------------------------------- */
int icz_adam_clamp_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                        float lr, float clip, int32_t step, void* stream);
/* The same update for `count` (<= 32) tensors in one launch; the pointer arrays are HOST arrays of device pointers. */
int icz_adam_clamp_multi(int32_t count, float* const* params, const float* const* grads, float* const* exp_avg,
                         float* const* exp_avg_sq, const int64_t* numel, float lr, float clip, int32_t step, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * CIDEr-D reward (Utils.py:319-367 get_self_critical_reward -> ciderD.py:30-55 -> ciderD_scorer.py:127-206)
 * The document-frequency table and the cooked references are built once on the host by the Python side
 * (they depend on strings) and uploaded as flat arrays; scoring runs on the device in float64.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct icz_ciderd icz_ciderd_t;

/* df table: open-addressing hash of n-grams of token ids.  keys [cap,4] int32 (unused slots -1-filled, n-gram
 * right-padded with -1), idf [cap] float64 = log(ref_len) - log(max(1, df)); cap is a power of two.
 * default_idf = log(ref_len) (n-gram absent from the table).  penalty [64] float64: exp(-d^2/(2 sigma^2)) for
 * |delta| = 0..63 computed on the host exactly as the reference does.  The three arrays are DEVICE arrays owned
 * by the caller and must outlive the handle. */
int icz_ciderd_create(const int32_t* df_keys, const double* df_idf, int64_t cap, double default_idf,
                      const double* penalty, icz_ciderd_t** out);
int icz_ciderd_destroy(icz_ciderd_t* h);

/* Cooked references of a batch, CSR over images -> refs -> n-gram entries:
 *   img_ref_ptr [B+1]; ref_ent_ptr [n_refs+1]; ent_key [n_ent,4] int32; ent_order [n_ent] int32 (1..4);
 *   ent_w [n_ent] float64 tf-idf weight; ref_norm [n_refs,4] float64; ref_len [n_refs] int32 (bigram count).
 * gen [B,T] int64 = sampled ids (sample_rl semantics), greedy [B,T] int64.  reward_out [B,T] float32 =
 * (CIDEr-D(sample) - CIDEr-D(greedy)) broadcast over T.  scores_out [2B] float64 or NULL. */
int icz_ciderd_reward(icz_ciderd_t* h, const int64_t* gen, const int64_t* greedy, int32_t B, int32_t T,
                      const int32_t* img_ref_ptr, const int32_t* ref_ent_ptr, const int32_t* ent_key,
                      const int32_t* ent_order, const double* ent_w, const double* ref_norm,
                      const int32_t* ref_len, float* reward_out, double* scores_out, void* stream);
/* Host-side cooking of references (ciderD_scorer.py:17-32, 128-153) from token ids into the flat arrays above -- pure host code,
 * HOST pointers throughout.  df_keys_host / df_idf_host: the table of icz_ciderd_create in host memory (it must then also hold
 * the n-grams that contain out-of-vocabulary words, under the private ids the caller gives those words).  tokens: the token
 * ids of n_refs references back to back, reference r = tokens[ref_tok_ptr[r] .. ref_tok_ptr[r+1]).  Outputs sized by the
 * caller (max_ent entries; at most 4 * tokens in all): ref_ent_ptr_out [n_refs+1] counts from 0. */
int icz_ciderd_cook_host(const int32_t* df_keys_host, const double* df_idf_host, int64_t cap, double default_idf,
                         const int32_t* tokens, const int32_t* ref_tok_ptr, int32_t n_refs, int64_t max_ent,
                         int32_t* ent_key_out, int32_t* ent_order_out, double* ent_w_out, int32_t* ref_ent_ptr_out,
                         double* ref_norm_out, int32_t* ref_len_out, int64_t* n_ent_out);
/* Host-side word -> id map of the scorer: the caption vocabulary (Caption_Vocabulary.word2ix, ClassRepository/CaptionVocabClass.py:1-19)
 * plus private ids >= V, in order of first appearance, for words outside it (references and the document-frequency table contain
 * them; ciderD_scorer.py keys n-grams by the word strings themselves).  words: n_words strings back to back, word i =
 * words[word_off[i] .. word_off[i+1]) with id word_id[i].  Thread-safe. */
typedef struct icz_ciderd_vocab icz_ciderd_vocab_t;
int icz_ciderd_vocab_create(const char* words, const int64_t* word_off, const int32_t* word_id, int32_t n_words, int32_t V,
                            icz_ciderd_vocab_t** out);
int icz_ciderd_vocab_destroy(icz_ciderd_vocab_t* v);
int icz_ciderd_vocab_oov_id(icz_ciderd_vocab_t* v, const char* word, int32_t len, int32_t* id_out);
/* icz_ciderd_cook_host with `precook`'s tokenisation (ciderD_scorer.py:17-32: s.split()) in front of it: text = n_refs ASCII
 * references separated by '\n', words by runs of blanks.  Outputs as icz_ciderd_cook_host (at most 4 * words entries). */
int icz_ciderd_cook_text(icz_ciderd_vocab_t* vocab, const int32_t* df_keys_host, const double* df_idf_host, int64_t cap, double default_idf,
                         const char* text, int64_t text_len, int32_t n_refs, int64_t max_ent,
                         int32_t* ent_key_out, int32_t* ent_order_out, double* ent_w_out, int32_t* ref_ent_ptr_out,
                         double* ref_norm_out, int32_t* ref_len_out, int64_t* n_ent_out);
/* The same against a device-resident STORE of cooked references (every image of the dataset cooked once, instead of the
 * reference's re-cooking of the batch's references on every call, ciderD.py:41-52): the seven arrays are the store's, laid
 * out as above with one CSR row per stored image, and img_slot [B] int32 (device) names the store row of image b of the
 * batch -- the only per-batch upload. */
int icz_ciderd_reward_indexed(icz_ciderd_t* h, const int64_t* gen, const int64_t* greedy, int32_t B, int32_t T,
                              const int32_t* img_slot, const int32_t* img_ref_ptr, const int32_t* ref_ent_ptr,
                              const int32_t* ent_key, const int32_t* ent_order, const double* ent_w, const double* ref_norm,
                              const int32_t* ref_len, float* reward_out, double* scores_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------
 * Building blocks exported for tests and benchmarks
 * ---------------------------------------------------------------------------------------------------------- */
/* C[M,N] = X[M,K] W[N,K]^T (+ bias[N]) on the fp32 MFMA; layout 0 = NT (y = x W^T), 1 = NN (C = X[M,K] W[K,N]),
 * 2 = TN (C = X[K,M]^T W[K,N]).  nsplit = 0 lets the library choose (then icz_gemm_workspace_floats(M,N) floats
 * of workspace suffice); an explicit nsplit needs nsplit*M*N floats. */
int icz_gemm_f32(int32_t layout, const float* X, int32_t ldx, const float* W, int32_t ldw, const float* bias,
                 float* C, int32_t ldc, int32_t M, int32_t N, int32_t K, int32_t nsplit, float* workspace,
                 size_t workspace_floats, void* stream);
size_t icz_gemm_workspace_floats(int32_t M, int32_t N);
/* Which kernel the many-row split-precision products (>= 128 rows: weight gradients, dgrad over all steps, beam / refiner / XE forward
 * GEMMs) run on: -1 = chosen per shape (default; the environment's ICZ_GEMM_BIG sets the same switch), 0 = the 128 x 128 two-barrier
 * kernel everywhere, 1..5 = one large-tile configuration of gemm_big_x3.hip everywhere, -2 = back to the environment's value.
 * Process-wide; meant for tests and A/B measurements, not for use while other threads launch. */
int icz_gemm_set_big_cfg(int32_t cfg);
/* The tile configuration a many-row product of that shape is given (host logic only, no GPU needed): 0 = the 128 x 128 two-barrier
 * kernel, 1 = 256 x 256 / eight waves, 4 = 128 x 128 three workgroups per CU; -1 = bad arguments.  Only meaningful for shapes that
 * reach those kernels at all (>= 128 rows and columns, K in whole 128-deep chunks; TN: >= 256 tiles of 128 x 128). */
int icz_gemm_big_cfg_for(int32_t layout, int32_t M, int32_t N, int32_t K, int32_t nsplit);
/* Weight gradients that share d y as ONE launch: out_j[M, cols_j] (row stride ldo_j) = dY[K, M]^T X_j[K, cols_j], j < ngroups <= 4,
 * cols_j % 256 == 0, M * sum(cols_j) >= 256 tiles of 128 x 128 (else ICZ_ERR: issue them one by one through icz_gemm_f32).
 * rows_live: optional device count of leading rows of K that matter (rounded up to 32).  What Butd::bptt does for the
 * LSTM weight gradients (BUTD_Model.py:137-145 under loss.backward(), Engine.py:186,270). */
int icz_gemm_tn_grouped(const float* dY, int32_t ldy, int32_t M, int32_t K, int32_t ngroups, const float* const* X, const int32_t* ldx,
                        const int32_t* cols, float* const* out, const int32_t* ldo, const int32_t* rows_live, void* stream);
/* Live timing of the dominant kernel (the forward GEMMs of a decoder step at 33..64 rows: gemm_resident_x3_kernel, or
 * gemm_nt_kernel<4, ...> with ICZ_GEMM_RESIDENT_X3=0; see icz_prof_select) with HIP events on its launch stream, for bench.py's
 * roofline line.  Between begin and end every launch is bracketed by an event
 * pair; end synchronises on them and reports the average duration [us] and the ALGORITHMIC bytes / flops per launch
 * (A read once + W read once + C written once; 2MNK). */
int icz_prof_begin(void);
/* Which launches icz_prof_begin brackets: 0 (default) = every forward GEMM of a decoder step at 33..64 rows, whatever kernel
 * takes it; 1 = only the launches of the resident-activation split-precision kernel (gemm_resident_x3_kernel: the LSTM-gate
 * and vocabulary-projection GEMMs, the dominant kernel of the SCST step since round 2). */
int icz_prof_select(int32_t which);
/* Average time [us] of an event pair around an EMPTY kernel on `stream` (n pairs): what the pair itself and the dispatch
 * of the bracketed kernel add to every duration icz_prof_end reports. */
int icz_prof_pair_overhead(void* stream, int32_t n, double* avg_us);
int icz_prof_end(double* avg_us, double* bytes_per_launch, double* flops_per_launch, long long* launches);
/* The same for the small kernels of a BUTD decoder step (bench.py's roofline entries of the attention trio and the select kernels):
 * between icz_kprof_begin and icz_kprof_end every EAGER launch (never inside a captured graph) of a group is bracketed by an event
 * pair on its stream.  Groups: 0 = SoftAttention.forward (BUTD_Model.py:49-62: dec_att product + scores + softmax / weighted sum,
 * three launches), 1 = greedy token choice + next embedding (:183), 2 = multinomial draw + log-prob + next embedding (:221-233),
 * 3 = one LSTMCell's pointwise part (:82-83).  icz_kprof_end synchronises and reports the average pair time [us] and the count. */
int icz_kprof_begin(void);
int icz_kprof_end(int32_t group, double* avg_us, long long* pairs);
/* What THIS box streams from HBM (bench.py's roofline.peak_measured, beside the 8 TB/s specification): a read-only kernel over
 * `buf` (`bytes` a multiple of 65536, at least 64 MiB; take it well above the 256 MB Infinity Cache), 64 KB chunks dealt to workgroups, sixteen
 * 16-byte loads in flight per lane, `reps` launches timed with one HIP event pair on `stream`; *gbs = bytes * reps / time. */
int icz_prof_stream_rate(const void* buf, size_t bytes, int32_t reps, void* stream, double* gbs);

#ifdef __cplusplus
}
#endif
#endif /* ICZ_H_ */
