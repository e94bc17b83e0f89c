"""Round 5: device-side early-out of the rollout / BPTT steps behind the reference's break (`if unfinished.sum() == 0: break`,
BUTD_Model.py:233, AoA_Model.py:400, NIC_Model.py:150): every kernel of such a step returns at entry."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102


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
    w_seq, w_lp, w_logits = ob.sample_rl(feats.cpu(), p, u.astype(np.float64), em, am, om, T, early_exit=True)
    steps_run = w_logits.shape[1]
    assert 4 <= steps_run <= 16, steps_run                    # the regime does what it is for: the reference broke out early
    same = (w_seq.numpy() == seq_h).all(1)
    assert same.sum() >= B - 2, (int(same.sum()), steps_run)  # a draw within fp32 rounding of a CDF edge may differ (test_gpu_round2)
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
    them, so this size is held to the oracle like every full-width case (test_gpu_round3._butd_scst_case: ids, log-probs 1e-4,
    gradients against float64)."""
    from test_gpu_round3 import _butd_scst_case
    rep, _ = _butd_scst_case(32, 20, seed=132, options={"merge_small": 32})
    assert max(v[0] for v in rep.values()) < 2e-2, rep


# ---- AoA: hipGraph replay of the SCST rollout pair and of the backward pass, shared feature projection -------------------------
def test_aoa_rollouts_and_backward_under_graph_replay_equal_eager_launches(golden_dir):
    """AoADetection SCST step (AoA_Model.py:698-753 behind Engine.py:256-270) at full width: the rollout pair and the REINFORCE
    backward as replayed hipGraphs against eager launches -- same Philox seeds, same inputs: tokens, log-probs, loss and every
    decoder gradient bit for bit; and the greedy ids / sampled rollout of rollouts() (ONE feature projection for both refiner
    passes) equal greedy() + sample() (each with its own)."""
    import os
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    B, T = 16, 20
    torch.manual_seed(7)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=36, enc_dim=D,
                                 max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, 36, D, device="cuda"))
    out = {}
    for graphs in (False, True):
        h = cap._handle()
        h.enable_graphs(graphs)
        res = []
        grads = h.new_grads()
        rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
        for rep in range(3):                                       # the first call captures, the others replay
            ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(500 + rep))
            loss, _ = h.sample_backward(rew, grads)
            res.append((ids.clone(), seq.clone(), lp.clone(), loss.clone(), {k: v.clone() for k, v in grads.items()}))
        out[graphs] = res
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    h = cap._handle()
    h.enable_graphs(False)
    ids0 = h.greedy(feats, T)
    seq0, lp0 = h.sample(feats, T, make_aoa_rng(500))
    # (since the paired refiner pass: at 16 images its GEMMs of 2 x 576 rows take another split-K decomposition than the single passes of
    # greedy() / sample() -- tokens equal, log-probs within fp32 rounding; test_aoa_paired_refiner_pass_equals_the_two_passes has the bits)
    assert torch.equal(ids0, out[False][0][0]) and torch.equal(seq0, out[False][0][1])
    assert (lp0 - out[False][0][2]).abs().max().item() < 1e-4


@pytest.mark.parametrize("B,bias", [(4, 4.0), (48, 7.0)])
def test_aoa_early_out_equals_running_every_step(golden_dir, B, bias):
    """AoA_Decoder.sample_rl's break (AoA_Model.py:400) on the device: with the steps behind it returning at entry (default) and with
    every step run as rounds 1 - 4 did (option early_out = 0; that form is pinned by the reference goldens in test_gpu_aoa.py):
    greedy prefix, sampled ids, log-probs, loss and the decoder gradients agree (the batched GEMMs stop behind the last live step:
    trailing all-zero rows dropped from the same sums)."""
    import os
    from simpleimagecaptionzoo_amd.aoa import AoaHandle, make_aoa_rng
    g = dict(np.load(os.path.join(golden_dir, "aoa_tiny.npz")))
    _, Rr, Dd, Hd, Ee, Vv, NH = [int(x) for x in g["dims"]]
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
    sd["decoder.predict.bias"][2] = bias
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda") for k, v in sd.items()}
    gen = torch.Generator(device="cpu")
    gen.manual_seed(B)
    feats = torch.relu(torch.randn(B, Rr, Dd, generator=gen)).cuda()
    T = 20
    out = {}
    for eo in (0, 1):
        h = AoaHandle(Rr, Dd, Hd, Ee, Vv, NH, B, T)
        h.bind(params)
        check = __import__("simpleimagecaptionzoo_amd._lib", fromlist=["check"])
        check.check(check.lib().icz_aoa_set_option(h._h, b"early_out", eo))
        ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(77))
        grads = h.new_grads()
        for v in grads.values():
            v.fill_(float("nan"))
        rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
        loss, _ = h.sample_backward(rew, grads)
        out[eo] = (ids.cpu().numpy(), seq.cpu().numpy(), lp.cpu().numpy(), loss.item(), {k: v.cpu().numpy() for k, v in grads.items()})
        h.close()
    a, b = out[1], out[0]
    assert (b[1][:, -1] == 0).all() and (b[1][:, 0] != 0).any()          # every sampled row ended before the last step
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
    for r in range(B):
        e = np.nonzero(b[0][r] == 2)[0]
        n = e[0] + 1 if e.size else T
        assert np.array_equal(a[0][r, :n], b[0][r, :n]), r
    for k in a[4]:
        assert np.isfinite(a[4][k]).all(), k
        scale = float(np.abs(b[4][k]).max()) + 1e-12
        assert float(np.abs(a[4][k] - b[4][k]).max()) <= 1e-5 * scale, (k, float(np.abs(a[4][k] - b[4][k]).max()), scale)


@pytest.mark.parametrize("B", [5, 40])
def test_nic_early_out_equals_running_every_step(B):
    """NIC DecoderRNN.sample_rl's break (NIC_Model.py:150) on the device, against every step run (option early_out = 0, the form the
    reference goldens in test_gpu_nic.py pin): sampled ids, log-probs, loss, decoder gradients and the gradient w.r.t. the image
    embedding agree."""
    from simpleimagecaptionzoo_amd._lib import check, lib
    from simpleimagecaptionzoo_amd.butd import make_rng
    from simpleimagecaptionzoo_amd.nic import NicHandle
    from simpleimagecaptionzoo_amd.synth import random_nic_params
    E_, H_, V_, T = 512, 512, 2543, 20
    params = random_nic_params(E_, H_, V_, "cuda", seed=9)
    params["predict.weight_g"][2] = 0.0
    params["predict.bias"][2] = 8.0
    gen = torch.Generator(device="cpu")
    gen.manual_seed(B)
    feats = torch.randn(B, E_, generator=gen).cuda()
    out = {}
    for eo in (0, 1):
        h = NicHandle(E_, H_, V_, B, T)
        h.bind(params)
        check(lib().icz_nic_set_option(h._h, b"early_out", eo))
        seq, lp = h.sample(feats, T, make_rng(31))
        grads = h.new_grads()
        rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
        res = h.sample_backward(rew, grads, want_dfeats=True)
        out[eo] = (seq.cpu().numpy(), lp.cpu().numpy(), res[0].item(), {k: v.cpu().numpy() for k, v in grads.items()},
                   [res[1].cpu().numpy(), res[2].cpu().numpy()])
        h.close()
    a, b = out[1], out[0]
    assert (b[0][:, -1] == 0).all() and (b[0][:, 0] != 0).any()          # every row ended before the last step
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2]
    for k in a[3]:
        scale = float(np.abs(b[3][k]).max()) + 1e-12
        assert np.isfinite(a[3][k]).all() and float(np.abs(a[3][k] - b[3][k]).max()) <= 1e-5 * scale, k
    for x, y in zip(a[4], b[4]):
        assert np.allclose(x, y, rtol=0, atol=1e-5 * (float(np.abs(y).max()) + 1e-12))


# ---------------------------------------------------------------------------------------------------------------------
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


@pytest.mark.parametrize("cfg", [1, 2, 3, 4, 5])
def test_large_tile_gemm_configurations_against_float64_and_the_128_tile_kernel(cfg):
    """Same arithmetic in the same order per accumulator: for one split-K decomposition every tile configuration returns the bits of
    gemm_tn128_x3_kernel; all of them within 3e-6 of max|C| of the float64 product (the bound of tests/test_gpu_butd.py)."""
    from simpleimagecaptionzoo_amd.butd import gemm, gemm_set_big_cfg
    try:
        for (lay, M, N, K, ns) in BIG_SHAPES:
            X, W = _gemm_operands(lay, M, N, K, 17 * M + N + K)
            gemm_set_big_cfg(0)
            base = gemm(lay, X, W, None, ns)
            gemm_set_big_cfg(cfg)
            out = gemm(lay, X, W, None, ns)
            ref = _gemm_ref64(lay, X, W)
            err = ((out.double() - ref).abs().max() / ref.abs().max()).item()
            assert err < 3e-6, (lay, M, N, K, ns, cfg, err)
            assert torch.equal(out, base), (lay, M, N, K, ns, cfg, (out - base).abs().max().item())
    finally:
        gemm_set_big_cfg(-2)


def test_large_tile_gemm_with_bias_and_per_shape_choice():
    """The routed default (cfg -1) with a bias on the direct path, the shapes of an XE step's vocabulary projection and of the refiner."""
    from simpleimagecaptionzoo_amd.butd import gemm
    for (M, N, K) in ((1280, 10102, 1024), (2304, 2048, 2048), (1088, 10102, 1024)):
        X, W = _gemm_operands("nt", M, N, K, M + N)
        b = torch.randn(N, device="cuda")
        out = gemm("nt", X, W, b, 1)
        ref = X.double() @ W.double().t() + b.double()
        assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6


@pytest.mark.parametrize("cfg", [-1, 1, 4])
def test_grouped_weight_gradients_equal_the_separate_products(cfg):
    """icz_gemm_tn_grouped (the LSTM weight gradients of Butd::bptt as one launch over column groups) = the products one by one, bit for
    bit, into strided outputs (W_ih column blocks and W_hh), also with the row limit of an early-ended rollout."""
    from simpleimagecaptionzoo_amd.butd import gemm, gemm_set_big_cfg, gemm_tn_grouped
    H, K = 1024, 640
    g = torch.Generator(device="cuda").manual_seed(5)
    dY = torch.randn(K, 4 * H, device="cuda", generator=g)
    Xs = [torch.randn(K, 2 * H, device="cuda", generator=g), torch.randn(K, H, device="cuda", generator=g), torch.randn(K, H, device="cuda", generator=g)]
    try:
        gemm_set_big_cfg(cfg)
        w_ih = torch.full((4 * H, 3 * H + 8), 7.0, device="cuda")        # [ctx | h1] and 8 columns nobody may touch
        w_hh = torch.zeros(4 * H, H, device="cuda")
        outs = [w_ih[:, :2 * H], w_ih[:, 2 * H:3 * H], w_hh]
        gemm_tn_grouped(dY, Xs, outs)
        assert torch.all(w_ih[:, 3 * H:] == 7.0)
        for x, o in zip(Xs, outs):
            ref = dY.double().t() @ x.double()
            assert ((o.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
        if cfg > 0:
            for x, o in zip(Xs, outs):
                # the same kernel on the product alone (4096 x 2048 and 4096 x 1024 both have >= 256 tiles of 128 x 128)
                assert torch.equal(o.contiguous(), gemm("tn", dY, x, None, 1))
        # rows behind the live count hold finite values whose products must not be read: the sum stops at 200 -> 224 rows
        live = torch.tensor([200], device="cuda", dtype=torch.int32)
        outs2 = gemm_tn_grouped(dY, Xs, None, live)
        for x, o in zip(Xs, outs2):
            ref = dY[:224].double().t() @ x[:224].double()
            assert ((o.double() - ref).abs().max() / ref.abs().max()).item() < 3e-6
    finally:
        gemm_set_big_cfg(-2)


@pytest.mark.parametrize("B", [64, 16])
def test_aoa_paired_refiner_pass_equals_the_two_passes(golden_dir, B):
    """icz_aoa_scst_rollouts with both refiner passes as ONE pass over [evaluation rows; training rows] (option refine_pair, default)
    against two passes: greedy ids and sampled ids equal; at the BASELINE batch (64 images: every GEMM of the pair takes the split-K
    decomposition of the single passes) log-probs, loss and every decoder gradient bit for bit, at 16 images within fp32 rounding."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, make_aoa_rng
    T = 20
    torch.manual_seed(11)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=36, enc_dim=D,
                                 max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, 36, D, device="cuda"))
    rew = torch.linspace(-1, 1, B, device="cuda").unsqueeze(1).repeat(1, T).contiguous()
    out = {}
    for pair in (0, 1):
        h = cap._handle()
        h.enable_graphs(False)
        h.set_option("refine_pair", pair)
        grads = h.new_grads()
        ids, seq, lp = h.rollouts(feats, T, make_aoa_rng(900))
        loss, _ = h.sample_backward(rew, grads)
        out[pair] = (ids.clone(), seq.clone(), lp.clone(), loss.clone(), {k: v.clone() for k, v in grads.items()})
    a, b = out[0], out[1]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    if B == 64:
        assert torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k
    else:
        assert (a[2] - b[2]).abs().max().item() < 1e-4 and abs(a[3].item() - b[3].item()) < 1e-5
        for k in a[4]:
            d = (a[4][k] - b[4][k]).abs().max().item()
            assert d <= 2e-5 * max(a[4][k].abs().max().item(), 1e-3), (k, d)
    cap._handle().set_option("refine_pair", 1)


@pytest.mark.parametrize("R,counts", [(36, None), (49, None), (36, "ragged"), (64, "ragged")])
def test_aoa_refiner_self_attention_on_the_matrix_pipe_equals_the_blocked_kernel(golden_dir, R, counts):
    """mha_self_mfma_kernel (option mha_mfma, default for <= 64 regions) against the register-blocked kernel that the reference goldens
    of test_gpu_aoa.py / test_gpu_aoa_adaptive.py pinned in rounds 1 - 4: refined regions of six layers within 2e-5 of max|x|, with fixed
    region sets (36 boxes, 7 x 7 grid) and with per-image counts (packed rows, masked keys)."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, RegionBatch
    B = 12
    torch.manual_seed(3)
    cap = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cuda:0", num_regions=R, enc_dim=D, max_batch=B)
    cap.to("cuda:0")
    feats = torch.relu(torch.randn(B, R, D, device="cuda"))
    batch = feats
    if counts == "ragged":
        c = [R, 10, R - 1, 17, 16, 15, 1, 33 if R > 33 else R, R, 12, 20, 2]
        m = (torch.arange(R, device="cuda").unsqueeze(0) < torch.tensor(c, device="cuda").unsqueeze(1))
        batch = RegionBatch(feats * m.unsqueeze(2), c)
    h = cap._handle()
    out = {}
    for on in (0, 1):
        h.set_option("mha_mfma", on)
        out[on] = h.refine(batch).clone()
    h.set_option("mha_mfma", 1)
    assert torch.isfinite(out[1]).all()
    assert (out[0] - out[1]).abs().max().item() <= 2e-5 * out[0].abs().max().item()


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
