// Batched beam search (DecoderRNN.beam_search_sample, Models/BUTD_Model.py:236-318) for many images at once.
// The reference decodes one image per call with a Python list comprehension over tensor elements every step
// (k host syncs per step, :282-283).  Here every image owns k consecutive decoder rows; after the shared decoder
// step a per-image workgroup does log-softmax + running score + top-k over (active beams x V), retires beams that
// emitted <end> (k shrinks exactly as in the reference, no length normalisation), and emits the row permutation
// that re-gathers the LSTM state.  No host synchronisation inside a step.
#include "beam_kernels.h"
#include "butd_impl.h"

namespace icz {

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
        ICZ_TRY(alloc((void**)&bm.cand_val, sizeof(float) * R_ * BEAM_MAX_K));
        ICZ_TRY(alloc((void**)&bm.cand_idx, sizeof(int) * R_ * BEAM_MAX_K));
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
        // Step 1: the k rows of an image are identical (<sta>, zero state) and only row 0 is scored (:273-274), so the decoder runs
        // ONE row per image (row img of the buffers); the top-k kernel reads image img's logits from row img and the state gather
        // fans row img out to the image's k rows.
        const bool compact = step == 1 && k > 1;
        StepIO s = {};
        s.rows = compact ? n_img : rows; s.feats = feats; s.img_of_row = compact ? nullptr : bm.img_of_row; s.it = it;
        s.rows_per_img = compact ? 1 : k;
        s.h1_in = h1[0]; s.c1_in = c1[0]; s.h2_in = h2[0]; s.c2_in = c2[0];
        s.h1_out = h1[1]; s.c1_out = c1[1]; s.h2_out = h2[1]; s.c2_out = c2[1];
        ICZ_TRY(this->step(s, st));
        BeamArgs a = {logits, dims.V, pad_vocab(dims.V), k, step, L, bm.n_act, bm.run, bm.seqs[sb], bm.seqs[sb ^ 1], bm.src_row, it,
                      bm.best_score, bm.best_len, bm.best_seq, bm.has_complete, bm.n_live + step};
        launch_beam_rowtopk(st, rows, a.logits, a.V, a.ldl, a.k, a.step, (const int*)bm.n_act, (const float*)bm.run, bm.cand_val, bm.cand_idx,
                            compact ? 1 : 0);
        hipLaunchKernelGGL(beam_merge_kernel, dim3(n_img), dim3(64), 0, st, a, (const float*)bm.cand_val, (const int*)bm.cand_idx);
        hipLaunchKernelGGL(beam_gather_kernel, dim3(cdiv(H, 1024), rows), dim3(256), 0, st, bm.src_row, H, h1[1], c1[1], h2[1], c2[1],
                           h1[0], c1[0], h2[0], c2[0], compact ? k : 1);
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
