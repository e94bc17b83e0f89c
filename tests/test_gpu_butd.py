"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the reference goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import butd as ob  # noqa: E402


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def sd_of(g, prefix="sd."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def make_handle(g, sd=None, max_rows=None, max_len=20):
    from simpleimagecaptionzoo_amd.butd import ButdHandle
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    sd = sd if sd is not None else sd_of(g)
    sd = ob.strip_prefix(sd)
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda") for k, v in sd.items()}
    h = ButdHandle(R, D, H, E, A, V, max_rows or max(B, 8), max_len)
    h.bind(params)
    return h, params


@pytest.mark.parametrize("layout,M,N,K", [
    ("nt", 64, 4096, 1024), ("nt", 5, 53, 96), ("nt", 64, 10102, 1024), ("nt", 130, 256, 2048),
    ("nt", 16, 128, 48), ("nt", 33, 70, 16), ("nt", 320, 4096, 1024), ("nt", 2304, 1024, 2048), ("nt", 130, 3000, 96),
    ("nn", 64, 1024, 4096), ("nn", 5, 96, 53 * 4), ("nn", 100, 64, 10104), ("nn", 20, 2048, 4096),
    ("tn", 4096, 1024, 1280), ("tn", 64, 52, 100), ("tn", 128, 2048, 37), ("tn", 1024, 10104, 64),
    # the split-precision 128 x 128-tile kernel in its three layouts (csrc/gemm_f32.hip: gemm_tn128_x3_kernel): more than one
    # round of tiles (one LDS buffer), one round (eight waves), ragged edges, batched-dgrad shapes with split-K
    ("tn", 4096, 3072, 1280), ("tn", 2100, 2052, 96), ("nn", 1280, 1024, 4096), ("nn", 1280, 1024, 10112), ("nn", 300, 132, 256),
    ("nt", 640, 10102, 1024), ("nt", 2304, 2048, 2048),
    # the skinny split-precision kernel (csrc/gemm_resident_x3.hip, 33..128 rows, whole 64-deep chunks): decoder-step shapes at 64
    # and 128 rows, ragged rows / columns, both column-tile widths, K = 64 (one stage) .. 4096
    ("nt", 64, 4096, 3072), ("nt", 128, 4096, 4096), ("nt", 128, 10102, 1024), ("nt", 50, 1024, 1024), ("nt", 100, 2100, 192),
    ("nt", 33, 70, 64), ("nt", 128, 1024, 1024), ("nt", 65, 4100, 448),
])
@pytest.mark.parametrize("nsplit", [0, 1, 3])
def test_gemm_against_float64(layout, M, N, K, nsplit):
    from simpleimagecaptionzoo_amd.butd import gemm
    rng = np.random.RandomState(M * 7 + N * 3 + K)
    f32 = lambda x: x.astype(np.float32).astype(np.float64)       # the operands as the device sees them
    if layout == "nt":
        X, W = f32(rng.randn(M, K)), f32(rng.randn(N, K))
        want = X @ W.T
    elif layout == "nn":
        X, W = f32(rng.randn(M, K)), f32(rng.randn(K, N))
        want = X @ W
    else:
        X, W = f32(rng.randn(K, M)), f32(rng.randn(K, N))
        want = X.T @ W
    bias = f32(rng.randn(N)) if (N % 4 == 0 or nsplit == 1) and layout != "tn" else None
    if nsplit == 3 and (K + 63) // 64 < 3:
        pytest.skip("fewer K chunks than splits")
    if nsplit != 1 and (M * N) % 4:
        pytest.skip("split-K reduce needs M*N % 4 == 0")
    Xd = torch.tensor(X, dtype=torch.float32, device="cuda")
    Wd = torch.tensor(W, dtype=torch.float32, device="cuda")
    bd = None if bias is None else torch.tensor(bias, dtype=torch.float32, device="cuda")
    got = gemm(layout, Xd, Wd, bd, nsplit).cpu().numpy().astype(np.float64)
    if bias is not None:
        want = want + bias
    # Against the exact (float64) product of the same fp32 operands: 3e-6 of the largest output for EVERY kernel -- fp32 MFMA
    # (k-ordered fp32 fma chains) and split precision (three bf16 pieces per operand, six bf16 MFMAs per product, fp32
    # accumulation: the dropped piece products are below 3 * 2^-24 |x w|) alike.  bf16 operands would miss it 1000-fold.
    # (fp32 accumulation itself grows like sqrt(K): the bound is scaled accordingly beyond K = 4096, the decoder's longest)
    err = np.abs(got - want).max() / max(1.0, np.abs(want).max())
    assert err < 3e-6 * max(1.0, np.sqrt(K / 4096.0)), err


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
def test_step_matches_reference(golden_dir, name):
    g = load(golden_dir, name)
    h, _ = make_handle(g)
    feats = torch.tensor(g["feats"], device="cuda")
    st = [torch.tensor(g["step_" + k], device="cuda") for k in ("h1", "c1", "h2", "c2")]
    it = torch.tensor(g["step_it"], device="cuda")
    ctx, alpha, logits = h.step(feats, it, *st)
    torch.cuda.synchronize()
    for got, key in ((st[0], "nh1"), (st[1], "nc1"), (st[2], "nh2"), (st[3], "nc2"), (ctx, "ctx"), (alpha, "alpha"),
                     (logits, "logits")):
        np.testing.assert_allclose(got.cpu().numpy(), g["step_" + key], atol=1e-4, rtol=1e-4, err_msg=key)


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
def test_greedy_token_exact(golden_dir, name):
    g = load(golden_dir, name)
    h, _ = make_handle(g)
    feats = torch.tensor(g["feats"], device="cuda")
    ids, alphas = h.greedy(feats, 20, want_alphas=True)
    torch.cuda.synchronize()
    assert np.array_equal(ids.cpu().numpy(), g["greedy_ids"])
    np.testing.assert_allclose(alphas.cpu().numpy(), g["greedy_alphas"], atol=1e-4)


def _masks(g, prefix, A):
    dev = "cuda"
    em = torch.tensor(g[prefix + "emb_mask"], device=dev)
    am = torch.tensor(np.unpackbits(g[prefix + "att_mask"], axis=-1)[..., :A], device=dev).contiguous()
    om = torch.tensor(g[prefix + "out_mask"], device=dev)
    return em, am, om


GRAD_ATOL = 5e-5


def _check_grads(grads, g, prefix):
    for k, v in grads.items():
        want = g[prefix + k]
        got = v.cpu().numpy()
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 2e-4 * scale + 2e-6, (k, np.abs(got - want).max(), scale)


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
def test_sample_rl_and_reinforce_backward(golden_dir, name):
    """sample_rl with injected uniforms/masks: ids exact, logprobs 1e-4; REINFORCE grads vs reference autograd."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, name)
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    sd = sd_of(g)
    sd["predict.bias"] = sd["predict.bias"].copy()
    sd["predict.bias"][2] = float(g["rl_end_bias"])
    h, params = make_handle(g, sd)
    feats = torch.tensor(g["feats"], device="cuda")
    em, am, om = _masks(g, "rl_", A)
    u = torch.tensor(g["rl_u"], dtype=torch.float32, device="cuda")
    rng = make_rng(0, u, em, am, om)
    seq, lp = h.sample(feats, 20, rng)
    torch.cuda.synchronize()
    assert np.array_equal(seq.cpu().numpy(), g["rl_seq"])
    np.testing.assert_allclose(lp.cpu().numpy(), g["rl_logprobs"], atol=1e-4)
    grads = h.new_grads()
    loss, msum = h.sample_backward(torch.tensor(g["rl_reward"], device="cuda"), grads)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["rl_loss"])) < 1e-4
    _check_grads(grads, g, "rl_grad.")


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
def test_xe_forward_backward(golden_dir, name):
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, name)
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    h, params = make_handle(g)
    feats = torch.tensor(g["feats"], device="cuda")
    em, am, om = _masks(g, "xe_", A)
    rng = make_rng(0, None, em, am, om)
    caps = torch.tensor(g["xe_captions"], device="cuda")
    lengths = g["xe_lengths"].tolist()
    logits = h.xe_forward(feats, caps, lengths, rng, train=True, want_logits=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss = h.xe_backward(grads, smoothing=0.1)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["xe_loss"])) < 1e-4
    _check_grads(grads, g, "xe_grad.")


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
def test_xe_with_scheduled_sampling(golden_dir, name):
    """DecoderRNN.forward with the decoder's ss_prob = 0.5 (BUTD_Model.py:120-132): gate and draw uniforms injected,
    packed logits / loss / gradients of the reference (the embedding gradient follows the tokens actually fed)."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, name)
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    h, params = make_handle(g)
    feats = torch.tensor(g["feats"], device="cuda")
    em, am, om = _masks(g, "xe_", A)
    rng = make_rng(0, None, em, am, om)
    caps = torch.tensor(g["xe_captions"], device="cuda")
    lengths = g["xe_lengths"].tolist()
    h.set_scheduled_sampling(float(g["ss_prob"]), g["ss_gate"], g["ss_draw"].astype(np.float32))
    logits = h.xe_forward(feats, caps, lengths, rng, train=True, want_logits=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), g["ss_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss = h.xe_backward(grads, smoothing=0.1)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(g["ss_loss"])) < 1e-4
    _check_grads(grads, g, "ss_grad.")
    # switched off again: the plain XE golden
    h.set_scheduled_sampling(0.0)
    logits = h.xe_forward(feats, caps, lengths, rng, train=True, want_logits=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)


def test_scheduled_sampling_generator_path_matches_oracle(golden_dir):
    """Library-generated gate / draw uniforms (Philox streams 5 and 6): regenerate them on the host, run the oracle with
    them and compare logits, loss and gradients; ss_prob = 1 replaces every token from step 2 on."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    from test_gpu_fullsize import _philox4x32_10

    def uniforms(seed, stream, T, B):       # csrc/rng.h rng_uniform: counter (row, 0, step, stream), top 24 bits of word 0
        rows = np.arange(B)
        return np.stack([(_philox4x32_10(rows, np.zeros(B), np.full(B, t), np.full(B, stream), seed & 0xFFFFFFFF, seed >> 32)[0]
                          >> np.uint64(8)).astype(np.float32) / np.float32(16777216.0) for t in range(T)])
    g = load(golden_dir, "butd_dec_tiny")
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    h, params = make_handle(g)
    feats = torch.tensor(g["feats"], device="cuda")
    em, am, om = _masks(g, "xe_", A)
    seed = 0xC0FFEE
    rng = make_rng(seed, None, em, am, om)
    caps = torch.tensor(g["xe_captions"], device="cuda")
    lengths = g["xe_lengths"].tolist()
    T = max(lengths)
    for prob in (0.3, 1.0):
        h.set_scheduled_sampling(prob)
        logits = h.xe_forward(feats, caps, lengths, rng, train=True, want_logits=True)
        grads = h.new_grads()
        loss = h.xe_backward(grads, smoothing=0.1)
        torch.cuda.synchronize()
        gate, draw = uniforms(seed, 5, T, B), uniforms(seed, 6, T, B)
        p = ob.to_params(sd_of(g), requires_grad=True)
        att = np.unpackbits(g["xe_att_mask"], axis=-1)[..., :A]
        toks = []
        ref = ob.forward_xe(torch.from_numpy(g["feats"]), torch.from_numpy(g["xe_captions"]), lengths, p, g["xe_emb_mask"], att,
                            g["xe_out_mask"], ss_prob=prob, ss_gate=gate, ss_draw=draw, tokens_out=toks)
        if prob == 1.0:
            assert all(not np.array_equal(it.numpy(), g["xe_captions"][: it.shape[0], t]) for t, it in enumerate(toks) if t >= 2)
        np.testing.assert_allclose(logits.cpu().numpy(), ref.detach().numpy(), atol=2e-4, rtol=1e-4)
        tgt = torch.tensor([g["xe_captions"][b, t + 1] for b, t in ob.packed_order(lengths)])
        rl = ob.label_smoothing_loss(ref, tgt, 0.1)
        assert abs(loss.item() - rl.item()) < 1e-4
        rl.backward()
        for k, v in p.items():
            np.testing.assert_allclose(grads[k].cpu().numpy(), v.grad.numpy(), atol=2e-4, rtol=2e-3, err_msg=k)
    h.set_scheduled_sampling(0.0)


def test_captioner_ss_prob_attribute_reaches_the_decoder(golden_dir):
    """Engine.py:143 sets `model.ss_prob`.  Default: ignored (the reference's live behaviour).  With `scheduled_sampling =
    True` the attribute drives the XE forward (autograd path of the reference's own training_epoch): packed logits and
    parameter gradients of the scheduled-sampling golden."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    from simpleimagecaptionzoo_amd.captioner import BUTDDetection_Captioner
    g = load(golden_dir, "butd_dec_tiny")
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    cap = BUTDDetection_Captioner(A, E, H, V, device="cuda:0", enc_dim=D, num_regions=R, max_batch=B)
    cap.load_state_dict({"decoder." + k: torch.tensor(v) for k, v in sd_of(g).items()}, strict=True)
    cap.to("cuda").train()
    em, am, om = _masks(g, "xe_", A)
    rng = make_rng(0, None, em, am, om)
    vi = {"bu_feats": torch.tensor(g["feats"], device="cuda"), "bu_bboxes": None, "bu_masks": None}
    caps = torch.tensor(g["xe_captions"], device="cuda")
    lengths = g["xe_lengths"].tolist()
    pred = cap(vi, caps, lengths, rng=rng)
    np.testing.assert_allclose(pred[0].detach().cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
    cap.ss_prob = float(g["ss_prob"])
    cap.set_scheduled_sampling_draws(g["ss_gate"], g["ss_draw"].astype(np.float32))
    pred = cap(vi, caps, lengths, rng=rng)          # default: the attribute is ignored, exactly like the reference's Captioner
    np.testing.assert_allclose(pred[0].detach().cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
    cap.scheduled_sampling = True
    cap.zero_grad()
    pred = cap(vi, caps, lengths, rng=rng)
    np.testing.assert_allclose(pred[0].detach().cpu().numpy(), g["ss_packed_logits"], atol=2e-4, rtol=1e-4)
    lp = torch.log_softmax(pred[0], dim=-1)
    tgt = torch.tensor(g["xe_packed_targets"], device="cuda")
    true = torch.full_like(lp, 0.1 / (V - 1)).scatter_(1, tgt.unsqueeze(1), 0.9)
    loss = torch.nn.functional.kl_div(lp, true, reduction="none").sum(1).sum() / lp.size(0)
    loss.backward()
    assert abs(loss.item() - float(g["ss_loss"])) < 1e-4
    _check_grads({k: p_.grad for k, p_ in cap.decoder.named_parameters()}, g, "ss_grad.")
    cap.ss_prob = 0.0
    pred = cap(vi, caps, lengths, rng=rng)
    np.testing.assert_allclose(pred[0].detach().cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)


def _beam_regime_sd(g, regime):
    sd = {k: v.copy() for k, v in sd_of(g).items()}
    if regime == "early":
        sd["predict.bias"][2] = 4.0
    elif regime == "never":
        sd["predict.bias"][2] = -1e4
    elif regime == "track":
        tok = int(g["beam_track_tok"])
        sd["predict.weight_v"][2] = sd["predict.weight_v"][tok]
        sd["predict.weight_g"][2] = sd["predict.weight_g"][tok]
        sd["predict.bias"][2] = sd["predict.bias"][tok] - 0.2
    return sd


@pytest.mark.parametrize("name", ["butd_dec_tiny", "butd_dec_odd", "butd_dec_spatial", "butd_dec_long"])
@pytest.mark.parametrize("regime", ["nat", "early", "never", "track"])
def test_beam_search_token_exact(golden_dir, name, regime):
    """Batched device beam search vs the reference's one-image-at-a-time beam search (k = 1, 3, 5)."""
    g = load(golden_dir, name)
    B = int(g["dims"][0])
    n = min(B, 3)
    h, _ = make_handle(g, _beam_regime_sd(g, regime), max_rows=n * 5)
    feats = torch.tensor(g["feats"][:n], device="cuda")
    for k in (1, 3, 5):
        seqs, lens = h.beam_search(feats, k, 50)
        torch.cuda.synchronize()
        seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
        for i in range(n):
            want = g["beam_%s_k%d_i%d" % (regime, k, i)].ravel()
            assert lens[i] == want.shape[0], (regime, k, i, lens[i], want.shape)
            assert np.array_equal(seqs[i, :lens[i]], want), (regime, k, i)


def test_graph_replay_equals_eager_and_reseeds(golden_dir):
    """hipGraph path: captured greedy / sample / backward replay bit-identically to eager launches, and a replay
    with a different Philox seed (device-resident seed) gives a different rollout."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, "butd_dec_tiny")
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    feats = torch.tensor(g["feats"], device="cuda")
    reward = torch.tensor(g["rl_reward"], device="cuda")
    outs = {}
    for mode in ("eager", "graph"):
        h, _ = make_handle(g)
        if mode == "graph":
            h.enable_graphs(True)
        res = []
        for rep, seed in enumerate((7, 7, 8)):
            ids = h.greedy(feats, 20).clone()
            seq, lp = h.sample(feats, 20, make_rng(seed))
            seq, lp = seq.clone(), lp.clone()
            grads = h.new_grads()
            loss, msum = h.sample_backward(reward, grads)
            torch.cuda.synchronize()
            res.append((ids.cpu(), seq.cpu(), lp.cpu(), loss.clone().cpu(), {k: v.clone().cpu() for k, v in grads.items()}))
        outs[mode] = res
    for (i0, s0, l0, L0, g0), (i1, s1, l1, L1, g1) in zip(outs["eager"], outs["graph"]):
        assert torch.equal(i0, i1) and torch.equal(s0, s1) and torch.equal(l0, l1) and torch.equal(L0, L1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k
    r = outs["graph"]
    assert torch.equal(r[0][1], r[1][1]) and torch.equal(r[0][2], r[1][2])      # same seed -> same rollout
    assert not torch.equal(r[0][2], r[2][2])                                    # new seed -> new rollout
    assert np.array_equal(r[0][0].numpy(), g["greedy_ids"])


def test_edge_cases_single_row_single_step_and_argument_errors(golden_dir):
    """Smallest shapes (one image, one step, beam of one) against the oracle, and the C ABI's argument checks."""
    from simpleimagecaptionzoo_amd._lib import IczError
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, "butd_dec_tiny")
    B, R, D, H, E, A, V = [int(x) for x in g["dims"]]
    h, params = make_handle(g, max_rows=8)
    p_cpu = {k: v.detach().cpu() for k, v in params.items()}
    feats = torch.tensor(g["feats"], device="cuda")
    one = feats[:1]
    # one image, one step
    ids = h.greedy(one, 1)
    want, _, _ = ob.greedy(one.cpu(), p_cpu, 1)
    assert ids.shape == (1, 1) and np.array_equal(ids.cpu().numpy(), want.numpy())
    # one image, full length == row 0 of the batched decode (rows are independent)
    assert np.array_equal(h.greedy(one, 20).cpu().numpy(), g["greedy_ids"][:1])
    # beam of one, one step: <sta> + the greedy token
    seqs, lens = h.beam_search(one, 1, 1)
    assert int(lens[0]) == 2 and seqs[0, :2].cpu().tolist() == [1.0, float(g["greedy_ids"][0, 0])]
    # teacher forcing with every caption of length 1 (a single time step, all rows active)
    caps = torch.tensor([[1, 7, 2]] * B, device="cuda")
    logits = h.xe_forward(feats, caps, [1] * B, None, train=False, want_logits=True)
    want = ob.forward_xe(feats.cpu(), caps.cpu(), [1] * B, p_cpu)
    np.testing.assert_allclose(logits.cpu().numpy(), want.numpy(), atol=1e-4)
    grads = h.new_grads()
    loss = h.xe_backward(grads, 0.1)
    assert np.isfinite(loss.item()) and all(torch.isfinite(v).all() for v in grads.values())
    # sampled rollout of a single step
    seq, lp = h.sample(one, 1, make_rng(3))
    assert seq.shape == (1, 1) and float(lp[0, 0]) <= 0.0
    # argument checks: nothing is clamped or silently truncated
    with pytest.raises(IczError):
        h.greedy(torch.zeros(9, R, D, device="cuda"), 20)                   # more rows than the handle's capacity
    with pytest.raises(IczError):
        h.greedy(torch.zeros(2, R + 1, D, device="cuda"), 20)               # wrong region count
    with pytest.raises(IczError):
        h.greedy(feats.double(), 20)                                        # wrong dtype
    with pytest.raises(IczError):
        h.beam_search(feats, 9, 20)                                         # beam wider than BEAM_MAX_K
    with pytest.raises(IczError):
        h.xe_forward(feats, torch.tensor([[1, 5, 6, 2]] * B, device="cuda"), [1, 2, 2, 1, 1][:B], None, train=False)   # lengths not sorted
    h.sample(one, 1, make_rng(3))
    h.sample_backward(torch.zeros(1, 1, device="cuda"), h.new_grads())
    with pytest.raises(IczError):
        h.sample_backward(torch.zeros(1, 1, device="cuda"), h.new_grads())     # the stored rollout was consumed


@pytest.mark.parametrize("cfg", [
    # B, R, D, H, E, A, V, T  -- sizes that exercise every GEMM variant: K multiples of 128 / 64 / neither, ragged N, tails
    (3, 36, 128, 128, 128, 128, 203, 5),
    (7, 36, 256, 64, 192, 320, 1001, 4),
    (2, 49, 100, 36, 20, 28, 57, 6),
    (9, 12, 64, 256, 64, 64, 130, 3),
    (64, 36, 32, 32, 32, 32, 41, 3),
    # 40 rows x 4 H = 2560 gate columns reach the resident GEMM; K segments of 10 | 4 | 10 (TD) and 4 | 10 | 10 (LM) 64-deep stages:
    # the 512-deep k ranges of round 6 (and their halves) cross segment boundaries
    (40, 12, 256, 640, 256, 64, 300, 3),
])
def test_random_shapes_match_oracle(cfg):
    """Randomly initialised decoders of assorted sizes: greedy ids / alphas, sampled log-probs and every REINFORCE gradient
    against the oracle (torch autograd), so that no code path depends on the benchmark's round sizes."""
    from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    B, R, D, H, E, A, V, T = cfg
    params = random_butd_params(R, D, H, E, A, V, "cuda", seed=sum(cfg))
    params["predict.weight_g"].mul_(8.0)
    params["embed.0.weight"].mul_(10.0)
    h = ButdHandle(R, D, H, E, A, V, max(B, 8), 20)
    h.bind(params)
    torch.manual_seed(sum(cfg))
    feats = torch.relu(torch.randn(B, R, D, device="cuda"))
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params.items()}
    ids, alphas = h.greedy(feats, T, want_alphas=True)
    with torch.no_grad():
        wi, wa, _ = ob.greedy(feats.cpu(), p, T)
    assert np.array_equal(ids.cpu().numpy(), wi.numpy())
    np.testing.assert_allclose(alphas.cpu().numpy(), wa.numpy(), atol=3e-5)
    rs = np.random.RandomState(sum(cfg))
    em = (rs.rand(T, B, E) < 0.5).astype(np.uint8)
    am = (rs.rand(T, B, R, A) < 0.5).astype(np.uint8)
    om = (rs.rand(T, B, H) < 0.5).astype(np.uint8)
    u = torch.tensor(rs.rand(T, B), dtype=torch.float32)
    reward = rs.randn(B, T).astype(np.float32)
    rng = make_rng(0, u.cuda(), torch.tensor(em, device="cuda"), torch.tensor(am, device="cuda"), torch.tensor(om, device="cuda"))
    seq, lp = h.sample(feats, T, rng)
    wseq, wlp, _ = ob.sample_rl(feats.cpu(), p, u.double().numpy(), em.astype(bool), am.astype(bool), om.astype(bool), T, early_exit=False)
    assert np.array_equal(seq.cpu().numpy(), wseq.numpy())
    np.testing.assert_allclose(lp.cpu().numpy(), wlp.detach().numpy(), atol=1e-4)
    loss = ob.reward_criterion(wlp, wseq, torch.from_numpy(reward))
    loss.backward()
    grads = h.new_grads()
    got, _ = h.sample_backward(torch.tensor(reward, device="cuda"), grads)
    assert abs(got.item() - loss.item()) < 1e-4
    for k, gr in grads.items():
        if k == "atten.affine.bias":
            continue
        want = p[k].grad.numpy()
        scale = max(1e-6, float(np.abs(want).max()))
        assert np.abs(gr.cpu().numpy() - want).max() <= 3e-4 * scale + 1e-7, (k, cfg)


def test_resident_gemm_decompositions_match_float64():
    """The resident-activation split-precision kernels (csrc/gemm_resident_x3.hip) at 33 .. 64 rows: 512-deep k ranges on one column
    tile with the planes in two halves (round 6; stage counts that are multiples of eight) and k ranges of four 64-deep stages on two
    tiles (the rest; a three-stage variant was measured and removed in round 3), and the 128-row kernel of
    round 4 (65 .. 128 rows: half a k range resident at a time, two live column tiles, direct 16-byte stores) -- ragged rows and
    columns, K = 768 .. 4096, K shapes whose stage count is no multiple of four fall back to the fp32-MFMA kernel -- against the
    float64 product: error at the fp32 kernel's level (3e-6 of max|C|)."""
    from simpleimagecaptionzoo_amd.butd import gemm
    torch.manual_seed(0)
    for M, N, K in ((64, 4096, 3072), (64, 4096, 4096), (64, 10112, 1024), (33, 2048, 768), (50, 4100, 3072), (64, 3072, 4096), (47, 2052, 1536),
                    (40, 2544, 512), (64, 2048, 1024), (36, 2060, 2560),          # round 6: one 512-deep range (no slabs), two, five
                    (128, 4096, 3072), (128, 4096, 4096), (128, 10112, 1024), (65, 2048, 512), (100, 4100, 3072), (127, 2052, 1536), (96, 3072, 8192)):
        X = torch.randn(M, K, device="cuda")
        W = torch.randn(N, K, device="cuda")
        b = torch.randn(N, device="cuda")
        want = X.double() @ W.double().t() + b.double()
        got = gemm("nt", X, W, b, 0).double()
        err = float((got - want).abs().max() / want.abs().max())
        assert err < 3e-6, (M, N, K, err)


