"""BUTD at the benchmark width (BASELINE configs 2, 3 and 4): the reference's own decoder step at 3 / 64 / 128 rows,
one whole 64 x 20 SCST step (ids, log-probs, reward, loss, gradients held to a float64 oracle), 8- and 16-row shards, beam 5 over 640
and 625 decoder rows, greedy at 128 rows, BUTDSpatial XE at 49 regions -- all against the CPU oracle / committed reference vectors.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (A, D, E, H, R, V, _butd_inputs, _butd_scst_case, _cpu, _first_divergence, _full_params, _sharp_params, attention_kink_units, check_grads_against_float64, probe_indices)  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows", [3, 64, 128])
def test_butd_step_matches_reference_at_full_width(golden_dir, rows):
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    g = dict(np.load(os.path.join(golden_dir, "butd_fullwidth_step.npz")))
    B = int(g["dims"][0])
    assert [int(x) for x in g["dims"][1:]] == [R, D, H, E, A, V]
    seed = int(g["seed"])
    h = ButdHandle(R, D, H, E, A, V, rows, 20)
    h.bind(random_butd_params(R, D, H, E, A, V, "cuda", seed=seed))
    feats_np, st_np, it_np = _butd_inputs(seed, B)
    idx = np.arange(rows) % B                   # golden row of each decoder row
    feats = torch.tensor(feats_np[idx], device="cuda")
    st = [torch.tensor(x[idx], device="cuda") for x in st_np]
    it = torch.tensor(it_np[idx], device="cuda")
    ctx, alpha, logits = h.step(feats, it, *st)
    torch.cuda.synchronize()
    for got, key in ((st[0], "nh1"), (st[1], "nc1"), (st[2], "nh2"), (st[3], "nc2"), (ctx, "ctx"), (alpha, "alpha"), (logits, "logits")):
        np.testing.assert_allclose(got.cpu().numpy(), g["s1_" + key][idx], atol=1e-4, rtol=1e-4, err_msg=key)
    tok = logits.argmax(1)
    clear = g["s1_margin"][idx] > 1e-3          # rows whose top-2 logits are further apart than any fp32 reordering
    assert np.array_equal(tok.cpu().numpy()[clear], g["s1_argmax"][idx][clear]) and clear.any()
    # the second step chained on the first one's outputs with the reference's argmax tokens
    ctx2, alpha2, logits2 = h.step(feats, torch.tensor(g["s1_argmax"][idx], device="cuda"), *st)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ctx2.cpu().numpy(), g["s2_ctx"][idx], atol=1e-4, rtol=1e-4)
    np.testing.assert_allclose(alpha2.cpu().numpy(), g["s2_alpha"][idx], atol=1e-4, rtol=1e-4)
    probe = probe_indices(B * V, 2048)
    want = np.zeros(B * V, dtype=np.float32)
    want[probe] = g["s2_logits_probe"]
    got2 = logits2.cpu().numpy()
    for r in range(rows):
        sel = probe[(probe >= idx[r] * V) & (probe < (idx[r] + 1) * V)]
        np.testing.assert_allclose(got2[r, sel - idx[r] * V], want[sel], atol=1e-4, rtol=1e-4)
    clear2 = g["s2_margin"][idx] > 1e-3
    assert np.array_equal(got2.argmax(1)[clear2], g["s2_argmax"][idx][clear2])
    h.close()


def test_fullsize_scst_step_64x20_matches_oracle():
    """BASELINE config 4's step itself, once (64 rows x 20 steps, every dropout site injected): greedy ids token-exact up to near-ties of
    the two largest logits (margin < 1e-4, at most 2 rows), sampled ids exact up to draws within 1e-6 of a CDF edge,
    log-probs 1e-4, the CIDEr-D reward bit-exact on the ids the device produced, the REINFORCE loss 1e-4, and the gradients against a
    float64 oracle with the fp32 oracle as the yardstick (see _fullwidth.check_grads_against_float64)."""
    rep, kink = _butd_scst_case(64, 20, seed=77, with_reward=True)
    assert kink.sum() < 200           # a few dozen of the 1024 units, not a blanket excuse
    worst = max(v[0] for v in rep.values())
    assert worst < 2e-2, rep


@pytest.mark.parametrize("B,T", [(8, 20), (16, 12)])
def test_fullsize_small_row_counts_8_and_16_rows(B, T):
    """8 rows x 20 steps (the shard of a 64-image batch on 8 GPUs under strong scaling) and 16 rows x 12 steps (BASELINE config 1's
    batch) at full width take the fp32-MFMA gemm_nt tiles (<= 32 rows) through every decoder-step GEMM: greedy ids, sampled ids and
    log-probs, and the REINFORCE gradients of an SCST rollout pair against the oracle (float64 criterion)."""
    rep, _ = _butd_scst_case(B, T, seed=100 + B)
    assert max(v[0] for v in rep.values()) < 2e-2, rep


@pytest.mark.parametrize("regime", ["nat", "end_biased"])
def test_fullsize_beam5_640_rows_matches_oracle(regime):
    """BASELINE config 3: beam 5 over 128 images = 640 decoder rows (split-precision many-row GEMM, grouped attention
    context, per-row top-k at V = 10102); 4 of the images against the oracle's one-image beam search (BUTD_Model.py:236-318)."""
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    params = _full_params(seed=78)
    n_img, k, steps = 128, 5, 20
    h = ButdHandle(R, D, H, E, A, V, n_img * k, 20)
    h.bind(params)
    torch.manual_seed(6)
    feats = torch.relu(torch.randn(n_img, R, D, device="cuda"))
    if regime == "end_biased":
        # <end> enters the top-k in mid-sentence (shrinking k, best-complete selection): its output row becomes a copy of the
        # most frequent greedy token's, 0.2 below it (the 'track' regime of tests/golden/make_goldens.py)
        ids = h.greedy(feats, steps).cpu().numpy()
        tok = int(np.bincount(ids[ids > 3].ravel()).argmax())
        params["predict.weight_v"][2] = params["predict.weight_v"][tok]
        params["predict.weight_g"][2] = params["predict.weight_g"][tok]
        params["predict.bias"][2] = params["predict.bias"][tok] - 0.2
        h.refresh()
    seqs, lens = h.beam_search(feats, k, steps)
    seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
    p = _cpu(params)
    finished = 0
    imgs = [0, 37, 90, 127]
    if regime == "end_biased":          # images whose greedy decode emits the tracked token early: <end> competes there
        early = [i for i in range(n_img) if tok in ids[i, :6]]
        imgs = (early + imgs)[:4]
    for i in imgs:
        want = ob.beam_search(feats[i:i + 1].cpu(), p, k, steps).numpy().ravel()
        got = seqs[i, :lens[i]]
        assert got.shape == want.shape and np.array_equal(got, want), (regime, i, got.tolist(), want.tolist())
        finished += int(want[-1] == 2)
    if regime == "end_biased":
        assert finished >= 1            # the regime does what it is for
    h.close()


@pytest.mark.parametrize("regime,sharpen,n_check", [("nat", 6.0, 12), ("end_biased", 6.0, 12), ("nat", 1.0, 8)])
def test_fullsize_beam5_128_images_checked_against_the_oracle(regime, sharpen, n_check):
    """BASELINE config 3 (beam 5 x 128 images = 640 decoder rows): 12 images per regime against the oracle's one-image beam search
    (BUTD_Model.py:236-318), and one run on un-sharpened random-init weights (sharpen = 1: margins as narrow as they get; 8 images)."""
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=178)
    params["predict.weight_g"].mul_(sharpen)
    n_img, k, steps = 128, 5, 20
    h = ButdHandle(R, D, H, E, A, V, n_img * k, 20)
    h.bind(params)
    torch.manual_seed(16)
    feats = torch.relu(torch.randn(n_img, R, D, device="cuda"))
    imgs = list(range(0, 128, 8))[:n_check]
    if regime == "end_biased":
        ids = h.greedy(feats, steps).cpu().numpy()
        tok = int(np.bincount(ids[ids > 3].ravel()).argmax())
        params["predict.weight_v"][2] = params["predict.weight_v"][tok]
        params["predict.weight_g"][2] = params["predict.weight_g"][tok]
        params["predict.bias"][2] = params["predict.bias"][tok] - 0.2
        h.refresh()
        early = [i for i in range(n_img) if tok in ids[i, :6]]
        imgs = (early + imgs)[:n_check]
    seqs, lens = h.beam_search(feats, k, steps)
    seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
    p = {k_: v.detach().cpu().clone() for k_, v in params.items()}
    finished, differ = 0, []
    for i in imgs:
        want = ob.beam_search(feats[i:i + 1].cpu(), p, k, steps).numpy().ravel()
        got = seqs[i, :lens[i]]
        if got.shape != want.shape or not np.array_equal(got, want):
            differ.append((i, got.tolist(), want.tolist()))
        finished += int(want[-1] == 2)
    # un-sharpened weights: candidate scores of different beams can tie within fp32 rounding; one of the images may take the other branch
    assert len(differ) <= (1 if sharpen == 1.0 else 0), differ
    if regime == "end_biased":
        assert finished >= 1
    h.close()


def test_beam5_at_125_rows_matches_oracle():
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    params = _sharp_params(92)
    n_img, k, steps = 25, 5, 20
    h = ButdHandle(R, D, H, E, A, V, n_img * k, 20)
    h.bind(params)
    torch.manual_seed(9)
    feats = torch.relu(torch.randn(n_img, R, D, device="cuda"))
    seqs, lens = h.beam_search(feats, k, steps)
    seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
    p = {k_: v.detach().cpu() for k_, v in params.items()}
    for i in (0, 12, 24):
        want = ob.beam_search(feats[i:i + 1].cpu(), p, k, steps).numpy().ravel()
        got = seqs[i, :lens[i]]
        assert got.shape == want.shape and np.array_equal(got, want), (i, got.tolist(), want.tolist())
    h.close()


def test_greedy_at_128_rows_matches_oracle_and_the_64_row_path():
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    params = _sharp_params(91)
    h = ButdHandle(R, D, H, E, A, V, 128, 20)
    h.bind(params)
    torch.manual_seed(8)
    feats = torch.relu(torch.randn(128, R, D, device="cuda"))
    ids128 = h.greedy(feats, 20).cpu().numpy()
    ids_a = h.greedy(feats[:64].contiguous(), 20).cpu().numpy()           # 33..64 rows: the 64-row resident kernel
    ids_b = h.greedy(feats[64:].contiguous(), 20).cpu().numpy()
    ids100 = h.greedy(feats[:100].contiguous(), 20).cpu().numpy()         # ragged: 100 of the kernel's 128 rows
    p = {k: v.detach().cpu() for k, v in params.items()}
    rows = [0, 63, 64, 99, 127]
    with torch.no_grad():
        want, _, logits = ob.greedy(feats[rows].cpu(), p, 20, hoisted=True)
    top2 = torch.topk(logits, 2, dim=2).values
    clear = ((top2[..., 0] - top2[..., 1]) > 1e-3).numpy()                # steps whose argmax no fp32 reordering can flip
    for j, r in enumerate(rows):
        n = int(np.argmin(clear[j])) if not clear[j].all() else 20        # compare up to the first unclear step
        assert n >= 10 and np.array_equal(ids128[r, :n], want[j, :n].numpy()), (r, n)
    same = (ids128 == np.concatenate([ids_a, ids_b])).all(1)
    assert same.sum() >= 126, int(same.sum())                             # the two kernels sum in different orders: near-ties may differ
    assert (ids100 == ids128[:100]).all(1).sum() >= 99
    h.close()


def test_butdspatial_xe_batch64_49_regions_full_width():
    """BASELINE config 2 (BUTDSpatial XE, 7 x 7 x 2048 grid features, batch 64): the decoder of BUTD_Model.py:321-440 over 49
    regions.  Evaluation-mode XE forward + label-smoothed loss + backward at full width against the oracle: packed logits of
    the rows it computes, loss 1e-4, gradients 2e-4.  (The reference DecoderRNN itself pins the same path at 49 regions and
    small width through tests/golden/butd_dec_spatial.npz.)  The oracle runs 8 of the 64 rows (they are independent); the
    device runs all 64 and the 8-row sub-batch separately."""
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    R49 = 49
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R49, D, H, E, A, V, "cuda", seed=79)
    params["predict.weight_g"].mul_(6.0)
    h = ButdHandle(R49, D, H, E, A, V, 64, 20)
    h.bind(params)
    torch.manual_seed(8)
    feats = torch.relu(torch.randn(64, R49, D, device="cuda"))
    rs = np.random.RandomState(4)
    lengths = sorted(rs.randint(5, 15, size=64).tolist(), reverse=True)
    L = max(lengths) + 1
    caps = torch.zeros(64, L, dtype=torch.int64)
    for b, n in enumerate(lengths):
        caps[b, 0] = 1
        caps[b, 1:n] = torch.from_numpy(rs.randint(4, V, size=n - 1))
        caps[b, n] = 2
    # full batch: finite, and row-for-row equal to the sub-batch run below (rows are independent)
    full_logits = h.xe_forward(feats, caps.cuda(), lengths, None, train=False, want_logits=True).clone()
    assert torch.isfinite(full_logits).all()
    g64 = h.new_grads()
    loss64 = h.xe_backward(g64, smoothing=0.1)
    assert np.isfinite(loss64.item())
    sub = [0, 9, 17, 26, 35, 44, 53, 63]
    sl = [lengths[i] for i in sub]
    sc = caps[sub]
    sf = feats[sub].contiguous()
    logits = h.xe_forward(sf, sc.cuda(), sl, None, train=False, want_logits=True)
    from _fullwidth import attention_kink_units, check_grads_against_float64
    order = ob.packed_order(sl)
    tgt = torch.tensor([int(sc[b, t + 1]) for b, t in order])
    gsets, trace64, p64 = {}, None, None
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):      # fp32 oracle = the yardstick, float64 oracle = the truth
        torch.set_default_dtype(dt)
        try:
            p = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in params.items()}
            trace = {}
            w_logits = ob.forward_xe(sf.cpu().to(dt), sc, sl, p, trace=trace)
            w_loss = ob.label_smoothing_loss(w_logits, tgt, 0.1)
            w_loss.backward()
            gsets[name] = {k: v.grad.numpy() for k, v in p.items()}
            if name == "f32":
                np.testing.assert_allclose(logits.cpu().numpy(), w_logits.detach().numpy(), atol=2e-4, rtol=1e-4)
                w_loss32 = float(w_loss.item())
            else:
                trace64, p64 = trace, {k: v.detach() for k, v in p.items()}
        finally:
            torch.set_default_dtype(torch.float32)
    grads = h.new_grads()
    loss = h.xe_backward(grads, smoothing=0.1)
    assert abs(loss.item() - w_loss32) < 1e-4
    kink = attention_kink_units(sf.cpu().double(), p64, trace64["h1"], None, active_rows=[sum(l > t for l in sl) for t in range(max(sl))])
    check_grads_against_float64(grads, gsets["f32"], gsets["f64"], {"atten.enc_att": kink, "atten.dec_att": kink})
    # the sub-batch rows of the full run: packed position of (b, t) in the 64-row batch
    pos = {bt: i for i, bt in enumerate(ob.packed_order(lengths))}
    fl = full_logits.cpu().numpy()
    sub_l = logits.cpu().numpy()
    for i, (b, t) in enumerate(order):
        np.testing.assert_allclose(fl[pos[(sub[b], t)]], sub_l[i], atol=1e-5, rtol=1e-5)
    h.close()
