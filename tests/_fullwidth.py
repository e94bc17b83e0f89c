"""Helpers shared by the full-width GPU parity tests (not collected by pytest): parameters / inputs at the benchmark width (36 x 2048
features, H = E = A = 1024, V = 10102), the excuse rules of SURVEY.md section 7 (near-ties of the two largest logits, draws within
fp32 rounding of a CDF edge), the float64 gradient yardstick, and the full-width SCST case every BUTD row-count test runs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from synth import feats_from_seed, probe_indices  # noqa: E402,F401

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102


def _cpu(params, grad=False):
    return {k: v.detach().cpu().clone().requires_grad_(grad) for k, v in params.items()}


def _full_params(seed=77, sharpen=6.0):
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=seed)
    params["predict.weight_g"].mul_(sharpen)      # trained decoders are far from uniform: well separated argmax / draws
    return params


def _first_divergence(got, want):
    """per row: index of the first differing step, or -1"""
    ne = got != want
    return np.where(ne.any(1), ne.argmax(1), -1)


def _excuse_greedy(greedy, w_greedy, w_glog, limit):
    """token-exact up to near-ties (< 1e-4) of the two largest logits at the first differing step; returns the excused rows"""
    div = _first_divergence(greedy, w_greedy.numpy())
    rows = np.nonzero(div >= 0)[0]
    for b in rows:
        top2 = torch.topk(w_glog[b, div[b]], 2).values
        assert float(top2[0] - top2[1]) < 1e-4, "greedy row %d differs at step %d with margin %g" % (b, div[b], float(top2[0] - top2[1]))
    assert len(rows) <= limit, "greedy: %d rows excused" % len(rows)
    return rows


def _excuse_sampled(seq, w_seq, w_slog, u, limit):
    """exact up to draws within 1e-6 of a CDF boundary (float64 softmax of the oracle's logits); returns the boolean mask of equal rows"""
    sdiv = _first_divergence(seq, w_seq.numpy())
    rows = np.nonzero(sdiv >= 0)[0]
    for b in rows:
        t = sdiv[b]
        c = torch.cumsum(torch.softmax(w_slog[b, t].detach().double(), 0), 0)
        tgt = float(u[t, b]) * float(c[-1])
        assert float((c - tgt).abs().min()) < 1e-6, "sampled row %d differs at step %d away from a CDF boundary" % (b, t)
    assert len(rows) <= limit, "sampled: %d rows excused" % len(rows)
    return sdiv < 0


def _units(x):
    """per-unit maxima: rows of a matrix (one output unit each), elements of a vector"""
    return x.reshape(x.shape[0], -1).max(1) if x.ndim >= 2 else x


def check_grads_against_float64(grads, g32, g64, kink_units=None, skip=("atten.affine.bias",)):
    """Every unit of every gradient tensor: |HIP - f64| <= 2 |torch32 - f64| + 2e-4 max|f64|.  `kink_units` ({tensor name prefix:
    boolean [A]}): attention units with a kept relu pre-activation within fp32 rounding of zero in the float64 pass; only those
    may leave the bound (a flipped relu element moves the unit's gradient by a finite amount in ANY fp32 evaluation), they are
    counted (at most 1 % of the units) and capped at 2e-2 of the maximum."""
    report = {}
    for k, gt in grads.items():
        if k in skip:
            continue
        got, w32, w64 = gt.cpu().double().numpy(), g32[k].astype(np.float64), g64[k]
        scale = max(1e-6, float(np.abs(w64).max()))
        e_hip, e_o32 = _units(np.abs(got - w64)), _units(np.abs(w32 - w64))
        bad = e_hip > 2.0 * e_o32 + 2e-4 * scale + 1e-7
        report[k] = (float(e_hip.max() / scale), float(e_o32.max() / scale), int(bad.sum()))
        if not bad.any():
            continue
        kink = None
        for pre, m in (kink_units or {}).items():
            if k.startswith(pre):
                kink = m
        assert kink is not None, (k, "units outside the float64 bound", np.nonzero(bad)[0][:8], report[k])
        unexplained = bad & ~kink
        assert not unexplained.any(), (k, "units outside the bound without a relu pre-activation at zero", np.nonzero(unexplained)[0][:8], report[k])
        assert bad.sum() <= max(1, bad.size // 100) and e_hip[bad].max() <= 2e-2 * scale, (k, int(bad.sum()), float(e_hip[bad].max()), scale)
    return report


def attention_kink_units(feats64, p64, h1_steps, att_masks, tol=3e-6, active_rows=None):
    """[A] boolean: attention unit a has an element z[t, b, r, a] = enc_ctx[b, r, a] + dec_ctx_t[b, a] (float64) that dropout keeps
    and that lies within `tol` x (|enc_ctx| + |dec_ctx| + 1) of zero -- the resolution at which two fp32 evaluations of the two
    dot products (2048 and 1024 terms) can disagree about the sign."""
    from oracle import butd as ob
    with torch.no_grad():
        enc = feats64 @ ob.wn_weight(p64, "atten.enc_att").t() + p64["atten.enc_att.bias"]          # [B, R, A]
        w_dec, b_dec = ob.wn_weight(p64, "atten.dec_att"), p64["atten.dec_att.bias"]
        hit = torch.zeros(enc.shape[2], dtype=torch.bool)
        for t, h1 in enumerate(h1_steps):
            dec = (h1 @ w_dec.t() + b_dec).unsqueeze(1)                                               # [B, 1, A]
            z = enc + dec
            near = z.abs() <= tol * (enc.abs() + dec.abs() + 1.0)
            if att_masks is not None:
                near &= torch.as_tensor(att_masks[t]).reshape(near.shape)
            if active_rows is not None:                                                               # XE: the batch shrinks with t
                near[active_rows[t]:] = False
            hit |= near.any(0).any(0)
    return hit.numpy()


def _butd_scst_case(B, T, seed, sharpen=6.0, options=None, with_reward=False):
    """device rollouts + REINFORCE gradients of B rows x T steps at full width, and the fp32 / float64 oracle passes on the same inputs
    (options: {handle option: value} set before the run).  with_reward: the step's CIDEr-D reward as well -- computed on the device from
    the ids the device produced, bit-exact against the oracle's (Utils.py:319-367) -- and used as the REINFORCE reward (plus a per-row
    signal: a random-init model scores ~0 against random references)"""
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=seed)
    params["predict.weight_g"].mul_(sharpen)
    h = ButdHandle(R, D, H, E, A, V, max(B, 8), T)
    h.bind(params)
    for name, value in (options or {}).items():
        h.set_option(name, value)
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    feats_c = torch.relu(torch.randn(B, R, D, generator=g))
    rs = np.random.RandomState(seed)
    em, am, om = rs.rand(T, B, E) < 0.5, rs.rand(T, B, R, A) < 0.5, rs.rand(T, B, H) < 0.5
    u = rs.rand(T, B).astype(np.float32)
    dev = "cuda"
    rng = make_rng(0, torch.tensor(u, device=dev), torch.tensor(em.astype(np.uint8), device=dev),
                   torch.tensor(am.astype(np.uint8), device=dev), torch.tensor(om.astype(np.uint8), device=dev))
    greedy, seq, lp = h.rollouts(feats_c.cuda(), T, rng)
    greedy, seq, lp = greedy.cpu().numpy(), seq.cpu().numpy(), lp.cpu().numpy()
    out = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dt)
        try:
            p = {k: v.detach().cpu().to(dt).requires_grad_(True) for k, v in params.items()}
            trace = {}
            w_seq, w_lp, w_slog = ob.sample_rl(feats_c.to(dt), p, u.astype(np.float64), em, am, om, T, early_exit=False, trace=trace, hoisted=True)
            out[name] = (p, w_seq, w_lp, w_slog, trace)
        finally:
            torch.set_default_dtype(torch.float32)
    p32 = out["f32"][0]
    with torch.no_grad():
        w_greedy, _, w_glog = ob.greedy(feats_c, {k: v.detach() for k, v in p32.items()}, T, hoisted=True)
    limit = max(1, B // 32)
    _excuse_greedy(greedy, w_greedy, w_glog, limit)
    ok = _excuse_sampled(seq, out["f32"][1], out["f32"][3], u, limit)
    ok &= (out["f64"][1].numpy() == seq).all(1)         # rows whose float64 draws agree as well take part in the gradient comparison
    assert ok.sum() >= B - 2 * limit
    np.testing.assert_allclose(lp[ok], out["f32"][2].detach().numpy()[ok], atol=1e-4)
    rw = (rs.randn(B, 1).astype(np.float32) * ok[:, None].astype(np.float32)).repeat(T, 1)
    if with_reward:
        from oracle import ciderd as oc
        from simpleimagecaptionzoo_amd.ciderd import CiderDReward
        from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
        from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
        vocab = synthetic_vocab(V)
        words = [vocab.ix2word[i] for i in range(V)]
        dfd = document_frequency(synthetic_references(2000, words, seed=0))
        refs = synthetic_references(B, words, seed=9)
        gts = {i: refs[i] for i in range(B)}
        scorer = CiderDReward(dfd["document_frequency"], dfd["ref_len"], vocab.word2ix, dev)
        reward = scorer.reward(torch.tensor(seq, device=dev), torch.tensor(greedy, device=dev), gts, list(range(B)))
        w_reward = oc.self_critical_reward(seq, greedy, gts, list(range(B)), dict(enumerate(words)),
                                           oc.DocFreq(dfd["document_frequency"], dfd["ref_len"]))
        assert reward.dtype == torch.float32 and reward.shape == (B, T) and np.array_equal(reward.cpu().numpy(), w_reward)
        rw = rw + np.where(ok[:, None], w_reward, 0.0).astype(np.float32)
    grads = h.new_grads()
    loss, _ = h.sample_backward(torch.tensor(rw, device=dev), grads)
    gsets = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        torch.set_default_dtype(dt)
        try:
            p, w_seq, w_lp, _, _ = out[name]
            w_seq_m = torch.from_numpy(np.where(ok[:, None], w_seq.numpy(), seq))
            w_loss = ob.reward_criterion(w_lp, w_seq_m, torch.from_numpy(rw).to(dt))
            w_loss.backward()
            gsets[name] = {k: v.grad.numpy() for k, v in p.items()}
            if name == "f32":
                assert abs(loss.item() - w_loss.item()) < 1e-4, (loss.item(), w_loss.item())
        finally:
            torch.set_default_dtype(torch.float32)
    p64, trace64 = out["f64"][0], out["f64"][4]
    kink = attention_kink_units(feats_c.double(), {k: v.detach() for k, v in p64.items()}, trace64["h1"], am)
    rep = check_grads_against_float64(grads, gsets["f32"], gsets["f64"], {"atten.enc_att": kink, "atten.dec_att": kink})
    h.close()
    return rep, kink


def _butd_inputs(seed, B):      # as tests/golden/make_fullwidth_goldens.py: butd_inputs
    rs = np.random.RandomState(seed)
    feats = feats_from_seed(seed + 1, B, R, D)
    st = [(rs.randn(B, H) * 0.5).astype(np.float32) for _ in range(4)]
    it = rs.randint(4, V, size=(B,)).astype(np.int64)
    return feats, st, it


def _aoa_captioner(seed):        # as make_fullwidth_goldens.py: aoa_captioner (the weights are a function of the seed)
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner
    torch.manual_seed(seed)
    m = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cpu")
    with torch.no_grad():
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed + 17)
        for name, prm in m.named_parameters():
            if name.endswith("norm.gain"):
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.2)
            if name.endswith("norm.bias"):
                prm.add_(torch.randn(prm.shape, generator=gen) * 0.1)
    return m


# The 128-row resident GEMM inside real decodes (65..128 decoder rows): greedy evaluation at batch 128 and beam 5 over 25 images
# (125 rows; its first step runs one row per image = 25 rows, the later ones 125) at full width against the CPU oracle.
def _sharp_params(seed):
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=seed)
    params["predict.weight_g"].mul_(6.0)        # trained decoders are far from uniform: well separated argmax (tests/test_gpu_butd_fullwidth.py)
    return params


def _end_biased_params(seed, p_end, B=64):
    """full-width parameters whose <end> logit is raised until a sampled step draws <end> with probability ~ p_end"""
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=seed)
    h = ButdHandle(R, D, H, E, A, V, B, 20)
    h.bind(params)
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    feats = torch.relu(torch.randn(B, R, D, generator=g)).cuda()
    params["predict.bias"][2] += float(np.log(p_end * V / (1.0 - p_end)))
    for _ in range(3):
        h.refresh()
        seq, _ = h.sample(feats, 20, make_rng(123))
        p = float((seq[:, 0] == 0).float().mean().clamp(1.0 / (4 * B), 1 - 1.0 / (4 * B)))
        params["predict.bias"][2] += float(np.log(p_end / (1 - p_end)) - np.log(p / (1 - p)))
    h.close()
    return params, feats


# ---- merged greedy + sampled chain of a small SCST batch (Butd::sample_chain with row0 = B) ---------------------------------------
def _small_case(B, merged, params, feats, seed, end_bias=None, small_nt=1):
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    T = 20
    p = {k: v.clone() for k, v in params.items()}
    if end_bias is not None:
        p["predict.weight_g"][2] = 0.0
        p["predict.bias"][2] = end_bias
    h = ButdHandle(R, D, H, E, A, V, B, T)
    h.bind(p)
    h.set_option("merge_small", 32 if merged else 0)
    h.set_option("small_nt", small_nt)
    rs = np.random.RandomState(seed)
    em, am, om = rs.rand(T, B, E) < 0.5, rs.rand(T, B, R, A) < 0.5, rs.rand(T, B, H) < 0.5
    u = rs.rand(T, B).astype(np.float32)
    dev = "cuda"
    rng = make_rng(0, torch.tensor(u, device=dev), torch.tensor(em.astype(np.uint8), device=dev),
                   torch.tensor(am.astype(np.uint8), device=dev), torch.tensor(om.astype(np.uint8), device=dev))
    greedy, seq, lp = h.rollouts(feats, T, rng)
    rw = torch.tensor(rs.randn(B, 1).astype(np.float32).repeat(T, 1), device=dev)
    grads = h.new_grads()
    for v in grads.values():
        v.fill_(float("nan"))
    loss, msum = h.sample_backward(rw, grads)
    out = (greedy.cpu().numpy(), seq.cpu().numpy(), lp.cpu().numpy(), loss.item(), msum.item(), {k: v.cpu().numpy() for k, v in grads.items()})
    h.close()
    return out


# Large-tile split-precision GEMM (csrc/gemm_big_x3.hip): every tile configuration against float64 and against the 128 x 128 kernel
def _gemm_operands(layout, M, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    if layout == "nt":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
    if layout == "nn":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
    return torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)


def _gemm_ref64(layout, X, W):
    X, W = X.double(), W.double()
    return X @ W.t() if layout == "nt" else (X @ W if layout == "nn" else X.t() @ W)


BIG_SHAPES = [("nt", 700, 4100, 1024, 1), ("nt", 700, 4100, 1024, 2), ("nt", 640, 1024, 1024, 4), ("nt", 2304, 1024, 1024, 1),
              ("nt", 129, 8200, 1152, 3), ("nt", 1280, 10102, 1024, 1),
              ("nn", 300, 260, 128, 1), ("nn", 1280, 1028, 2176, 4), ("nn", 130, 516, 2176, 1), ("nn", 1280, 1024, 4096, 2),
              ("tn", 2052, 2060, 96, 1), ("tn", 4096, 1024, 320, 1), ("tn", 2048, 2048, 64, 1), ("tn", 4100, 2044, 304, 1)]


def _check_dp_fields(j):
    """round 5: the N > 1 line explains itself -- per-phase times (max over ranks) and the same steps with the gradient exchange not
    overlapped with the backward pass"""
    ph = j["phases_ms"]
    assert set(("rollouts", "reward", "backward", "allreduce_exposed", "adam")) <= set(ph) and all(ph[k] >= 0 for k in ph)
    assert ph["rollouts"] > 0 and ph["backward"] > 0 and ph["adam"] > 0
    ov = j["dp_overlap"]
    assert ov["on_ms"] > 0 and ov["off_ms"] > 0 and ov["allreduce_exposed_off_ms"] > 0 and ov["allreduce_exposed_on_ms"] >= 0
    assert j["vs_baseline"] is None and j["vs_reference_in_container"] > 0
