"""Behaviour of the BUTD handle around the kernels (8b): device-side early-out of the steps behind sample_rl's break,
graph replay = eager launches, the merged greedy + sampled chain of small batches, BPTT on the transposed weight copies, training
buffers that grow, graph caches against re-bound parameters and mixed call sequences, top-k under massive ties.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _fullwidth import (A, D, E, H, R, V, _butd_scst_case, _end_biased_params, _small_case)  # noqa: E402

pytestmark = pytest.mark.gpu


def test_fullsize_rollout_with_early_break_matches_oracle_and_autograd():
    """64 rows x 20 steps at full width, <end> likely enough that every row has finished around step 10: ids, log-probs and the
    REINFORCE gradients against the oracle WITH the reference's break (oracle/butd.py: early_exit=True) -- the steps behind the
    break are zeros in the outputs and contribute nothing to any gradient.  The handle's step slots are first filled by a
    rollout that never ends (stale activations in every slot the short rollout leaves untouched)."""
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    B, T = 64, 20
    params, feats = _end_biased_params(91, 0.35)
    h = ButdHandle(R, D, H, E, A, V, B, T)
    long_params = random_butd_params(R, D, H, E, A, V, "cuda", seed=92)
    h.bind(long_params)
    seq0, _ = h.sample(feats * 3.0, T, make_rng(5))          # fills all 20 slots; random weights never emit <end>
    assert int((seq0[:, -1] != 0).sum()) > B // 2
    g0 = h.new_grads()
    h.sample_backward(torch.ones(B, T, device="cuda"), g0)
    h.bind(params)
    rs = np.random.RandomState(7)
    em, am, om = rs.rand(T, B, E) < 0.5, rs.rand(T, B, R, A) < 0.5, rs.rand(T, B, H) < 0.5
    u = rs.rand(T, B).astype(np.float32)
    dev = "cuda"
    rng = make_rng(0, torch.tensor(u, device=dev), torch.tensor(em.astype(np.uint8), device=dev),
                   torch.tensor(am.astype(np.uint8), device=dev), torch.tensor(om.astype(np.uint8), device=dev))
    greedy, seq, lp = h.rollouts(feats, T, rng)
    seq_h, lp_h = seq.cpu().numpy(), lp.cpu().numpy()
    # the SCST baseline: the reference's greedy ids (icz_butd_greedy: no break, BUTD_Model.py:171-186) up to and including every row's
    # first <end> -- all the reward reads (Utils.py:354) -- and zeros behind the step at which the last row emitted it
    g_roll, g_full = greedy.cpu().numpy(), h.greedy(feats, T).cpu().numpy()
    ends = [(np.nonzero(r == 2)[0][0] if (r == 2).any() else T - 1) for r in g_full]
    for b in range(B):
        assert np.array_equal(g_roll[b, :ends[b] + 1], g_full[b, :ends[b] + 1]), b
    if all((r == 2).any() for r in g_full):
        assert max(ends) < T - 1 and (g_roll[:, max(ends) + 1:] == 0).all()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params.items()}
    w_seq, w_lp, w_logits = ob.sample_rl(feats.cpu(), p, u.astype(np.float64), em, am, om, T, early_exit=True, hoisted=True)
    steps_run = w_logits.shape[1]
    assert 4 <= steps_run <= 16, steps_run                    # the regime does what it is for: the reference broke out early
    same = (w_seq.numpy() == seq_h).all(1)
    assert same.sum() >= B - 2, (int(same.sum()), steps_run)  # a draw within fp32 rounding of a CDF edge may differ (test_gpu_butd_fullwidth)
    assert (seq_h[:, steps_run:] == 0).all() and (lp_h[:, steps_run:] == 0).all()
    np.testing.assert_allclose(lp_h[same], w_lp.detach().numpy()[same], atol=1e-4)
    rw = (rs.randn(B, 1).astype(np.float32) * same[:, None]).repeat(T, 1)
    grads = h.new_grads()
    for v in grads.values():
        v.fill_(float("nan"))
    loss, _ = h.sample_backward(torch.tensor(rw, device=dev), grads)
    w_seq_m = torch.from_numpy(np.where(same[:, None], w_seq.numpy(), seq_h))
    w_loss = ob.reward_criterion(w_lp, w_seq_m, torch.from_numpy(rw))
    w_loss.backward()
    assert abs(loss.item() - w_loss.item()) < 1e-4
    for k, gt in grads.items():
        want = p[k].grad.numpy()
        got = gt.cpu().numpy()
        assert np.isfinite(got).all(), k
        scale = max(1e-6, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 5e-4 * scale + 1e-7, (k, float(np.abs(got - want).max()), scale)
    h.close()


def test_early_break_under_graph_replay_equals_eager_launches():
    """The Engine's form: Philox randomness, whole rollouts and the backward pass replayed as hipGraphs.  Same seed, same
    inputs: replayed and eager launches give the same tokens, log-probs and gradients bit for bit, break or no break."""
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    B, T = 64, 20
    params, feats = _end_biased_params(93, 0.4)
    out = {}
    for graphs in (False, True):
        h = ButdHandle(R, D, H, E, A, V, B, T)
        h.bind(params)
        h.enable_graphs(graphs)
        res = []
        for rep in range(3):                                   # the first call captures, the others replay
            greedy, seq, lp = h.rollouts(feats, T, make_rng(1000 + rep))
            grads = getattr(h, "_test_grads", None) or h.new_grads()
            h._test_grads = grads
            rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous() if rep == 0 else res[0][3]
            loss, _ = h.sample_backward(rew, grads)
            res.append((seq.clone(), lp.clone(), {k: v.clone() for k, v in grads.items()}, rew, loss.clone()))
        out[graphs] = res
        h.close()
    broke = 0
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[4], b[4])
        for k in a[2]:
            assert torch.equal(a[2][k], b[2][k]), k
        broke += int((a[0][:, -1] == 0).all())
    assert broke == 3                                          # every rollout ended before the last step


@pytest.mark.parametrize("B,end_bias", [(8, None), (16, None), (5, None), (8, 9.5), (16, 9.0)])
def test_merged_small_row_chain_equals_the_two_separate_chains(B, end_bias):
    """Engine.py:256-262 at <= 16 images: greedy baseline (eval mode) and sampled rollout (train mode) as ONE chain of 2 B decoder rows
    against the two chains of rounds 1 - 4, same injected randomness.  Up to 32 rows both forms run the same fp32-MFMA GEMM tiles
    with the same split: greedy ids, sampled ids, log-probs, loss and mask sum are EQUAL, with and without the reference's break
    (end_bias: every row finishes early; greedy ids then agree up to each row's <end>); the gradients agree to fp32 rounding (the
    sums over (t, b) of the batched weight-gradient GEMMs run over 2 B rows per step, half of them zero: another blocking of
    the same sum)."""
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=300 + B)
    g = torch.Generator(device="cpu")
    g.manual_seed(B)
    feats = torch.relu(torch.randn(B, R, D, generator=g)).cuda()
    a = _small_case(B, True, params, feats, 11, end_bias)
    b = _small_case(B, False, params, feats, 11, end_bias)
    if end_bias is None:
        assert np.array_equal(a[0], b[0])
    else:
        assert (b[1][:, -1] == 0).all()                           # the regime does what it is for: every sampled row has ended
        for r in range(B):
            e = np.nonzero(b[0][r] == 2)[0]
            n = e[0] + 1 if e.size else b[0].shape[1]
            assert np.array_equal(a[0][r, :n], b[0][r, :n]), r
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]
    for k in a[5]:
        assert np.isfinite(a[5][k]).all(), k
        scale = float(np.abs(b[5][k]).max()) + 1e-12
        assert float(np.abs(a[5][k] - b[5][k]).max()) <= 2e-5 * scale, (k, float(np.abs(a[5][k] - b[5][k]).max()), scale)


def test_merged_chain_at_32_images_rides_the_64_row_kernel_and_matches_the_oracle():
    """2 x 32 rows take the resident split-precision kernel (the separate 32-row chains take the fp32-MFMA tiles): not bit-equal to
    them, so this size is held to the oracle like every full-width case (_fullwidth._butd_scst_case: ids, log-probs 1e-4,
    gradients against float64).  12 steps: the 20-step forms of this size class are the 8- and 64-row cases of test_gpu_butd_fullwidth.py."""
    from _fullwidth import _butd_scst_case
    rep, _ = _butd_scst_case(32, 12, seed=132, options={"merge_small": 32})
    assert max(v[0] for v in rep.values()) < 2e-2, rep


@pytest.mark.parametrize("B,merged,end_bias", [(8, True, None), (16, False, None), (24, False, None), (16, True, 9.0)])
def test_small_row_bptt_on_transposed_weights_equals_the_nn_kernel(B, merged, end_bias):
    """BPTT steps of <= 32 rows take d[ctx | h1], d h2 and d h1 (BUTD_Model.py:137-145 under loss.backward()) as NT products on the
    transposed weight copies (option small_nt, default) instead of NN products on the weights themselves: same rollout, loss
    and mask sum; every gradient within fp32 rounding (another kernel, another blocking of the same sums)."""
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=700 + B)
    g = torch.Generator(device="cpu")
    g.manual_seed(B + 1)
    feats = torch.relu(torch.randn(B, R, D, generator=g)).cuda()
    a = _small_case(B, merged, params, feats, 13, end_bias, small_nt=1)
    b = _small_case(B, merged, params, feats, 13, end_bias, small_nt=0)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert a[3] == b[3] and a[4] == b[4]
    for k in a[5]:
        assert np.isfinite(a[5][k]).all(), k
        scale = float(np.abs(b[5][k]).max()) + 1e-12
        assert float(np.abs(a[5][k] - b[5][k]).max()) <= 2e-5 * scale, (k, float(np.abs(a[5][k] - b[5][k]).max()), scale)


def test_xe_long_captions_grow_the_training_buffers(golden_dir):
    """captions longer than the handle's initial 20 steps.  The handle starts at max_len 20, runs the
    ordinary 12-step golden, then the 41-step one from the reference (buffers re-allocated), then the short one again."""
    from test_gpu_butd import _check_grads, _masks, load, make_handle
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, "butd_dec_long")
    B, R_, D_, H_, E_, A_, V_ = [int(x) for x in g["dims"]]
    lengths = g["xe_lengths"].tolist()
    assert max(lengths) > 25
    h, _ = make_handle(g, max_len=20)
    h.enable_graphs(True)
    feats = torch.tensor(g["feats"], device="cuda")
    ids_before = h.greedy(feats, 20).clone()           # a captured graph that must survive the re-allocation
    caps = torch.tensor(g["xe_captions"], device="cuda")
    em, am, om = _masks(g, "xe_", A_)
    for rep in range(2):
        short = [min(l, 7) for l in lengths]
        h.xe_forward(feats, caps[:, :8].contiguous(), short, make_rng(1), train=True)
        h.xe_backward(h.new_grads(), 0.1)
        logits = h.xe_forward(feats, caps, lengths, make_rng(0, None, em, am, om), train=True, want_logits=True)
        np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
        grads = h.new_grads()
        loss = h.xe_backward(grads, smoothing=0.1)
        assert abs(loss.item() - float(g["xe_loss"])) < 1e-4
        _check_grads(grads, g, "xe_grad.")
        assert torch.equal(h.greedy(feats, 20), ids_before)
    # a sampled rollout longer than the initial capacity as well
    seq, lp = h.sample(feats, 30, make_rng(5))
    assert seq.shape == (B, 30) and torch.isfinite(lp).all()


def test_rebinding_parameters_invalidates_captured_graphs(golden_dir):
    """a handle with graphs enabled that is re-bound to other parameter tensors must not replay graphs
    that carry the old pointers."""
    from test_gpu_butd import load, make_handle
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, "butd_dec_tiny")
    h, params = make_handle(g)
    h.enable_graphs(True)
    feats = torch.tensor(g["feats"], device="cuda")
    ids0 = h.greedy(feats, 20).clone()
    seq0, lp0 = [x.clone() for x in h.sample(feats, 20, make_rng(3))]
    assert np.array_equal(ids0.cpu().numpy(), g["greedy_ids"])
    # new tensors, different values: the embedding table rolled by one row, LSTM biases perturbed
    p2 = {k: v.clone() for k, v in params.items()}
    p2["embed.0.weight"] = torch.roll(p2["embed.0.weight"], 1, 0).contiguous()
    p2["language_model.bias_ih"] = p2["language_model.bias_ih"] + 0.3
    for v in params.values():
        v.fill_(float("nan"))            # anything still reading the old tensors is caught
    h.bind(p2)
    ids1 = h.greedy(feats, 20).clone()
    seq1, lp1 = [x.clone() for x in h.sample(feats, 20, make_rng(3))]
    he, _ = make_handle(g)
    he.bind(p2)                          # eager handle on the same tensors
    assert torch.equal(ids1, he.greedy(feats, 20))
    se, le = he.sample(feats, 20, make_rng(3))
    assert torch.equal(seq1, se) and torch.equal(lp1, le)
    assert torch.isfinite(lp1).all() and not torch.equal(ids1, ids0)


@pytest.mark.parametrize("V", [600, 3000])
def test_beam_topk_with_massive_ties_takes_the_lowest_indices(V):
    """The per-row top-k of a beam step keeps a candidate list of the scores >= the n-th largest thread maximum; a row full of
    equal scores overflows the list and takes the insertion path instead (beam_kernels.h).  Ties must go to the lowest flat
    index (the order of a top-k over the flattened [k, V] scores, BUTD_Model.py:271-276).  Output layer with zero weights, so
    the logits are the bias: 200 tokens tie at the top, among them <pad>, <sta> and <end>."""
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    R_, D_, H_, E_, A_ = 4, 16, 8, 8, 8
    params = random_butd_params(R_, D_, H_, E_, A_, V, "cuda", seed=5)
    params["predict.weight_g"].zero_()
    rs = np.random.RandomState(V)
    top = np.concatenate([[0, 1, 2], 3 + rs.choice(V - 3, 197, replace=False)])
    bias = np.zeros(V, dtype=np.float32)
    bias[top] = 1.0
    params["predict.bias"].copy_(torch.from_numpy(bias))
    h = ButdHandle(R_, D_, H_, E_, A_, V, 3 * 5, 20)
    h.bind(params)
    feats = torch.relu(torch.randn(3, R_, D_, device="cuda"))
    # k = 5, one step: the five lowest tied indices are 0, 1, 2 and two more -> <end> is among them -> [<sta>, <end>]
    seqs, lens = h.beam_search(feats, 5, 1)
    assert lens.cpu().tolist() == [2, 2, 2] and (seqs[:, :2].cpu() == torch.tensor([1.0, 2.0])).all()
    # k = 1: the single best of a full tie is token 0 at every step, never <end>
    seqs, lens = h.beam_search(feats, 1, 4)
    assert lens.cpu().tolist() == [5, 5, 5] and (seqs[:, :5].cpu() == torch.tensor([1.0, 0, 0, 0, 0])).all()
    # k = 2: tokens 0 and 1 at step 1; at step 2 the two best flat indices are row 0's tokens 0 and 1
    seqs, lens = h.beam_search(feats, 2, 2)
    assert lens.cpu().tolist() == [3, 3, 3] and (seqs[:, :3].cpu() == torch.tensor([1.0, 0, 0])).all()


def test_sample_and_merged_rollouts_on_one_handle_keep_their_own_backward_graphs():
    """icz_butd_sample (B rows per step slot, offset 0) and icz_butd_scst_rollouts at B <= merge_small (2 B rows per slot, the
    sampled rows behind the greedy ones) write the caller's same persistent seq / log-prob / loss buffers, so the captured backward of
    one used to be replayed for the other (the key lacked the slot geometry).  Alternate the two on one handle under graph replay:
    every call's loss and gradients equal those of the same sequence launched eagerly, bit for bit."""
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    B, T = 8, 20
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=41)
    g = torch.Generator(device="cpu")
    g.manual_seed(41)
    feats = torch.relu(torch.randn(B, R, D, generator=g)).cuda()
    rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
    out = {}
    for graphs in (False, True):
        h = ButdHandle(R, D, H, E, A, V, B, T)
        h.bind(params)
        h.set_option("merge_small", 32)
        h.enable_graphs(graphs)
        res = []
        for rep in range(2):                        # graphs: capture on the first round, replay on the second
            for merged in (False, True):
                grads = h.new_grads()
                if merged:
                    _, seq, lp = h.rollouts(feats, T, make_rng(70 + rep))
                else:
                    seq, lp = h.sample(feats, T, make_rng(70 + rep))
                loss, msum = h.sample_backward(rew, grads)
                res.append((seq.clone(), lp.clone(), loss.clone(), msum.clone(), {k: v.clone() for k, v in grads.items()}))
        h.close()
        out[graphs] = res
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    # the two forms draw the same tokens from the same seed (the merged chain's sampled half = the separate sampled chain)
    assert torch.equal(out[True][0][0], out[True][1][0])
