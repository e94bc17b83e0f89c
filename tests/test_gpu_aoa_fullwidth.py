"""AoA at the benchmark width (BASELINE config 5): the reference's own captioner over two steps, one whole 64 x 20 SCST
step with every dropout site injected, beam 5 over 320 decoder rows, AoASpatial at 49 regions -- against the CPU oracle / committed
reference vectors.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (D, E, R, V, _aoa_captioner, _first_divergence, check_grads_against_float64, feats_from_seed, probe_indices)  # noqa: E402

pytestmark = pytest.mark.gpu


def test_aoa_two_steps_match_reference_at_full_width(golden_dir):
    g = dict(np.load(os.path.join(golden_dir, "aoa_fullwidth_step.npz")))
    B, seed = int(g["dims"][0]), int(g["seed"])
    m = _aoa_captioner(seed).to("cuda")
    m.eval()
    rs = np.random.RandomState(seed)
    feats = torch.tensor(feats_from_seed(seed + 1, B, R, D), device="cuda")
    caps = np.zeros((B, 3), dtype=np.int64)
    caps[:, 0] = 1
    caps[:, 1:] = rs.randint(4, V, size=(B, 2))
    h = m._handle()
    refined = h.refine(feats).cpu().numpy().reshape(-1)
    np.testing.assert_allclose(refined[probe_indices(refined.size, 4096)], g["refined_probe"], atol=1e-4, rtol=1e-4)
    assert abs(refined.astype(np.float64).sum() - float(g["refined_sum"])) < 1e-4 * refined.size ** 0.5 * 10
    assert abs((refined.astype(np.float64) ** 2).sum() / float(g["refined_sumsq"]) - 1.0) < 1e-5
    packed = m({"bu_feats": feats, "bu_bboxes": None, "bu_masks": None}, torch.tensor(caps, device="cuda"), [2] * B)[0]
    torch.cuda.synchronize()
    np.testing.assert_allclose(packed.cpu().numpy(), g["packed_logits"], atol=1e-4, rtol=1e-4)
    clear = g["margin"] > 1e-3
    assert np.array_equal(packed.argmax(1).cpu().numpy()[clear], g["argmax"][clear]) and clear.any()


def test_fullsize_aoa_scst_step_64x20_matches_oracle():
    """BASELINE config 5 at the size it is timed on (`secondary.aoa_scst_step`): one whole AoADetection SCST step -- refiner in
    evaluation and in training mode (all four dropout sites of the six layers injected), greedy and sampled rollout of 64 images x
    20 steps as the two concurrent chains (`icz_aoa_scst_rollouts`), CIDEr-D reward, REINFORCE loss and the decoder gradients --
    against the CPU oracle on the same features / parameters / uniforms / keep-masks.  Same excuse rule as the BUTD step above."""
    from oracle import aoa as oa
    from oracle import butd as ob
    from oracle import ciderd as oc
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    from simpleimagecaptionzoo_amd.ciderd import CiderDReward
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    B, T, Hd, NH = 64, 20, 1024, 8
    vocab = synthetic_vocab(V)
    words = [vocab.ix2word[i] for i in range(V)]
    dfd = document_frequency(synthetic_references(2000, words, seed=0))
    torch.manual_seed(17)
    cap = AoADetection_Captioner(V, max_batch=B, max_beam=1).cuda()
    with torch.no_grad():
        cap.decoder.predict.weight_g.mul_(6.0)
        for l in cap.aoa_refine.aoa_layers:          # clones() starts the six layers identical; make them differ
            for p_ in l.parameters():
                p_.add_(torch.randn_like(p_) * 0.01)
    h = cap._handle()
    g = torch.Generator(device="cpu")
    g.manual_seed(4321)
    feats_c = torch.relu(torch.randn(B, R, D, generator=g))
    feats = feats_c.cuda()
    rs = np.random.RandomState(5)
    keep = lambda shape, p: (rs.rand(*shape) >= p)
    masks = {"proj": keep((B, R, Hd), 0.5), "ref_att": keep((6, B, NH, R, R), 0.1), "ref_aoa": keep((6, B, R, 2 * Hd), 0.3),
             "ref_sc": keep((6, B, R, Hd), 0.1), "emb": keep((T, B, E), 0.5), "ctx": keep((T, B, Hd), 0.5),
             "att": keep((T, B, NH, R), 0.1), "out": keep((T, B, Hd), 0.5)}
    u = rs.rand(T, B).astype(np.float32)
    dev = "cuda"
    rng = make_aoa_rng(0, torch.tensor(u, device=dev), {k: torch.tensor(v.astype(np.uint8), device=dev) for k, v in masks.items()})
    greedy, seq, lp = h.rollouts(feats, T, rng)
    greedy, seq, lp = greedy.cpu().numpy(), seq.cpu().numpy(), lp.cpu().numpy()

    # ---- oracle (gradients only for the decoder: the only parameters in the reference's optimizer, AoA_Model.py:669-674), once in
    #      fp32 (the reference's arithmetic: ids, log-probs, loss) and once in float64 (the truth the gradients are held to)
    runs = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dt)
        try:
            pp = {k: v.detach().cpu().to(dt).requires_grad_(k.startswith("decoder.")) for k, v in cap.state_dict().items()}
            runs[name] = (pp,) + tuple(oa.sample_rl(feats_c.to(dt), pp, u.astype(np.float64), masks, T, early_exit=False, hoisted=True))
        finally:
            torch.set_default_dtype(torch.float32)
    p, w_seq, w_lp = runs["f32"]
    with torch.no_grad():
        w_greedy, w_glog = oa.greedy(feats_c, p, T, hoisted=True)
    div = _first_divergence(greedy, w_greedy.numpy())
    excused = 0
    for b in np.nonzero(div >= 0)[0]:
        top2 = torch.topk(w_glog[b, div[b]], 2).values
        assert float(top2[0] - top2[1]) < 1e-4, ("greedy row %d differs at step %d with margin %g" % (b, div[b], float(top2[0] - top2[1])))
        excused += 1
    assert excused <= 2, "greedy: %d rows excused" % excused
    sdiv = _first_divergence(seq, w_seq.numpy())
    assert (sdiv >= 0).sum() <= 2, "sampled: %d rows differ" % int((sdiv >= 0).sum())
    for b in np.nonzero(sdiv >= 0)[0]:              # a draw within fp32 rounding of a CDF boundary lands on the neighbouring token
        assert abs(int(seq[b, sdiv[b]]) - int(w_seq[b, sdiv[b]])) <= 2, (b, seq[b], w_seq[b])
    ok = (sdiv < 0) & (runs["f64"][1].numpy() == seq).all(1)
    assert ok.sum() >= B - 4
    np.testing.assert_allclose(lp[ok], w_lp.detach().numpy()[ok], atol=1e-4)

    # ---- reward: bit-exact on the ids the device produced
    refs = synthetic_references(B, words, seed=9)
    gts = {i: refs[i] for i in range(B)}
    scorer = CiderDReward(dfd["document_frequency"], dfd["ref_len"], vocab.word2ix, dev)
    reward = scorer.reward(torch.tensor(seq, device=dev), torch.tensor(greedy, device=dev), gts, list(range(B)))
    w_reward = oc.self_critical_reward(seq, greedy, gts, list(range(B)), dict(enumerate(words)),
                                       oc.DocFreq(dfd["document_frequency"], dfd["ref_len"]))
    assert np.array_equal(reward.cpu().numpy(), w_reward)

    # ---- REINFORCE loss and decoder gradients: |HIP - f64| <= 2 |torch32 - f64| + 2e-4 max per unit (tests/_fullwidth.py)
    from _fullwidth import check_grads_against_float64
    rw = w_reward.copy()
    rw[~ok] = 0.0
    rw = rw + rs.randn(B, 1).astype(np.float32) * ok[:, None].astype(np.float32)
    grads = h.new_grads()
    loss, msum = h.sample_backward(torch.tensor(rw, device=dev), grads)
    gsets = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dt)
        try:
            pp, ws, wl = runs[name]
            w_seq_m = torch.from_numpy(np.where(ok[:, None], ws.numpy(), seq))
            w_loss = ob.reward_criterion(wl, w_seq_m, torch.from_numpy(rw).to(dt))
            w_loss.backward()
            gsets[name] = {k: v.grad.numpy() for k, v in pp.items() if v.grad is not None}
            if name == "f32":
                assert abs(loss.item() - w_loss.item()) < 1e-4, (loss.item(), w_loss.item())
        finally:
            torch.set_default_dtype(torch.float32)
    check_grads_against_float64(grads, gsets["f32"], gsets["f64"], None, skip=("decoder.aoa_block.linear_K.bias",))


@pytest.mark.parametrize("regime", ["nat", "end_biased"])
def test_fullsize_aoa_beam5_matches_oracle(regime):
    """BASELINE config 5 decodes with beam 5: AoA beam search at full width over 64 images = 320 decoder rows (many-row
    split-precision GEMMs with split-K over the LSTM's three K segments, register top-k at V = 10102); 4 of the images against
    the oracle's one-image beam search (AoA_Model.py:403-502), in the natural regime and with <end> competing in mid-sentence."""
    from oracle import aoa as oa
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner
    n_img, k, steps = 64, 5, 20
    torch.manual_seed(23)
    cap = AoADetection_Captioner(V, max_batch=n_img, max_beam=k).cuda()
    with torch.no_grad():
        cap.decoder.predict.weight_g.mul_(6.0)
    h = cap._handle()
    feats = torch.relu(torch.randn(n_img, R, D, device="cuda"))
    imgs = [0, 21, 40, 63]
    if regime == "end_biased":
        ids = h.greedy(feats, steps).cpu().numpy()
        tok = int(np.bincount(ids[ids > 3].ravel()).argmax())
        with torch.no_grad():
            cap.decoder.predict.weight_v[2] = cap.decoder.predict.weight_v[tok]
            cap.decoder.predict.weight_g[2] = cap.decoder.predict.weight_g[tok]
            cap.decoder.predict.bias[2] = cap.decoder.predict.bias[tok] - 0.2
        h = cap._handle()                  # refreshes the weight-normed copies
        early = [i for i in range(n_img) if tok in ids[i, :6]]
        imgs = (early + imgs)[:4]
    seqs, lens = h.beam_search(feats, k, steps)
    seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
    p = {kk: v.detach().cpu().clone() for kk, v in cap.state_dict().items()}
    finished = 0
    for i in imgs:
        want = oa.beam_search(feats[i:i + 1].cpu(), p, k, steps).numpy().ravel()
        got = seqs[i, :lens[i]]
        assert got.shape == want.shape and np.array_equal(got, want), (regime, i, got.tolist(), want.tolist())
        finished += int(want[-1] == 2)
    if regime == "end_biased":
        assert finished >= 1


def test_aoaspatial_49_regions_gradients_match_oracle():
    """AoASpatial (7 x 7 grid = 49 regions, AoA_Model.py:638-655) at full width: XE (evaluation mode) and REINFORCE (every dropout
    site injected) decoder gradients for 4 images against torch autograd through the oracle."""
    from oracle import aoa as oa
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    R49, B, T, Hd, NH = 49, 4, 6, 1024, 8
    torch.manual_seed(31)
    cap = AoADetection_Captioner(V, max_batch=B, max_beam=1, num_regions=R49).cuda()
    with torch.no_grad():
        cap.decoder.predict.weight_g.mul_(6.0)
        for l in cap.aoa_refine.aoa_layers:
            for p_ in l.parameters():
                p_.add_(torch.randn_like(p_) * 0.01)
    h = cap._handle()
    feats = torch.relu(torch.randn(B, R49, D, device="cuda"))
    feats_c = feats.cpu()

    def fresh():
        return {k: v.detach().cpu().clone().requires_grad_(k.startswith("decoder.")) for k, v in cap.state_dict().items()}

    def compare(grads, p, tol):
        for k, gt in grads.items():
            if k == "decoder.aoa_block.linear_K.bias":
                continue                                   # identically zero (softmax shift invariance)
            want = p[k].grad.numpy()
            scale = max(1e-6, float(np.abs(want).max()))
            err = float(np.abs(gt.cpu().numpy() - want).max())
            assert err <= tol * scale + 1e-7, (k, err, scale)
    # ---- XE
    lengths = [5, 4, 3, 2]
    caps = torch.tensor([[1, 17, 230, 4001, 9, 2], [1, 9, 77, 51, 2, 0], [1, 5000, 8, 2, 0, 0], [1, 44, 2, 0, 0, 0]])
    logits = h.xe_forward(feats, caps.cuda(), lengths, None, train=False, want_logits=True)
    p = fresh()
    want_logits = oa.forward_xe(feats_c, caps, lengths, p)
    np.testing.assert_allclose(logits.cpu().numpy(), want_logits.detach().numpy(), atol=5e-4, rtol=1e-4)
    tgt = torch.tensor([caps[b, t + 1] for b, t in ob.packed_order(lengths)])
    loss = ob.label_smoothing_loss(want_logits, tgt, 0.1)
    loss.backward()
    grads = h.new_grads()
    got = h.xe_backward(grads, 0.1)
    assert abs(got.item() - loss.item()) < 1e-4
    compare(grads, p, 3e-4)
    # ---- REINFORCE with every dropout site injected
    rs = np.random.RandomState(9)
    keep = lambda shape, pr: (rs.rand(*shape) >= pr)
    masks = {"proj": keep((B, R49, Hd), 0.5), "ref_att": keep((6, B, NH, R49, R49), 0.1), "ref_aoa": keep((6, B, R49, 2 * Hd), 0.3),
             "ref_sc": keep((6, B, R49, Hd), 0.1), "emb": keep((T, B, E), 0.5), "ctx": keep((T, B, Hd), 0.5),
             "att": keep((T, B, NH, R49), 0.1), "out": keep((T, B, Hd), 0.5)}
    u = rs.rand(T, B).astype(np.float32)
    rng = make_aoa_rng(0, torch.tensor(u, device="cuda"), {k: torch.tensor(v.astype(np.uint8), device="cuda") for k, v in masks.items()})
    seq, lp = h.sample(feats, T, rng)
    p = fresh()
    w_seq, w_lp = oa.sample_rl(feats_c, p, u.astype(np.float64), masks, T, early_exit=False, hoisted=True)
    assert np.array_equal(seq.cpu().numpy(), w_seq.numpy())
    np.testing.assert_allclose(lp.cpu().numpy(), w_lp.detach().numpy(), atol=1e-4)
    rw = rs.randn(B, 1).astype(np.float32).repeat(T, 1)
    grads = h.new_grads()
    loss, _ = h.sample_backward(torch.tensor(rw, device="cuda"), grads)
    w_loss = ob.reward_criterion(w_lp, w_seq, torch.from_numpy(rw))
    w_loss.backward()
    assert abs(loss.item() - w_loss.item()) < 1e-4
    compare(grads, p, 3e-4)
