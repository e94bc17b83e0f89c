"""Round-3 parity additions ("next round" 4 and 6b).

  * gradients at the measured size against a FLOAT64 oracle: |HIP - f64| <= 2 |torch32 - f64| + 2e-4 max per unit, and the units of
    the two attention projections that still leave that bound must show the cause the round-2 test only asserted -- a kept
    attention pre-activation relu(enc_ctx + dec_ctx) (BUTD_Model.py:57-58) within fp32 rounding of zero in the float64 pass;
  * the row-count paths no full-size test touched: 8 and 16 decoder rows x 20 steps at full width (fp32 gemm_nt tiles: the
    strong-scaling shard of 64 images over 8 GPUs and BASELINE config 1's batch);
  * NIC at BASELINE config 1's own size (Flickr8K vocabulary 2543, E = H = 512, batch 16, 20 steps);
  * beam 5 x 128 images: 16 images per regime, and one run on un-sharpened random-init weights;
  * AoASpatial (49 regions): XE and REINFORCE decoder gradients against the oracle;
  * bench.py's N > 1 control flow (two ranks on one GPU, gloo) under weak and strong scaling.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _first_divergence(got, want):
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


def _butd_scst_case(B, T, seed, sharpen=6.0, options=None):
    """device rollouts + REINFORCE gradients of B rows x T steps at full width, and the fp32 / float64 oracle passes on the same inputs
    (options: {handle option: value} set before the run)"""
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
            w_seq, w_lp, w_slog = ob.sample_rl(feats_c.to(dt), p, u.astype(np.float64), em, am, om, T, early_exit=False, trace=trace)
            out[name] = (p, w_seq, w_lp, w_slog, trace)
        finally:
            torch.set_default_dtype(torch.float32)
    p32 = out["f32"][0]
    with torch.no_grad():
        w_greedy, _, w_glog = ob.greedy(feats_c, {k: v.detach() for k, v in p32.items()}, T)
    limit = max(1, B // 32)
    _excuse_greedy(greedy, w_greedy, w_glog, limit)
    ok = _excuse_sampled(seq, out["f32"][1], out["f32"][3], u, limit)
    ok &= (out["f64"][1].numpy() == seq).all(1)         # rows whose float64 draws agree as well take part in the gradient comparison
    assert ok.sum() >= B - 2 * limit
    np.testing.assert_allclose(lp[ok], out["f32"][2].detach().numpy()[ok], atol=1e-4)
    rw = (rs.randn(B, 1).astype(np.float32) * ok[:, None].astype(np.float32)).repeat(T, 1)
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


def test_fullsize_scst_gradients_64x20_against_float64_oracle():
    """The bench workload's REINFORCE gradients (64 rows x 20 steps, every dropout site injected) against a float64 oracle with the
    fp32 oracle as the yardstick (see check_grads_against_float64)."""
    rep, kink = _butd_scst_case(64, 20, seed=77)
    assert kink.sum() < 200           # a few dozen of the 1024 units, not a blanket excuse
    worst = max(v[0] for v in rep.values())
    assert worst < 2e-2, rep


@pytest.mark.parametrize("B", [8, 16])
def test_fullsize_small_row_counts_8_and_16_rows_x_20_steps(B):
    """8 rows (the shard of a 64-image batch on 8 GPUs under strong scaling) and 16 rows (BASELINE config 1's batch) at full width
    take the fp32-MFMA gemm_nt tiles (<= 32 rows) through every decoder-step GEMM: greedy ids, sampled ids and log-probs, and
    the REINFORCE gradients of a 20-step SCST rollout pair against the oracle (float64 criterion)."""
    rep, _ = _butd_scst_case(B, 20, seed=100 + B)
    assert max(v[0] for v in rep.values()) < 2e-2, rep


def test_nic_config1_size_matches_oracle():
    """BASELINE config 1 at its own size: NIC decoder, Flickr8K-size vocabulary 2543, E = H = 512, batch 16, 20 steps, random-init
    (un-sharpened) weights: greedy ids exact, sampled ids exact up to CDF-boundary draws, log-probs 1e-4, REINFORCE gradients
    2e-4 (NIC_Model.py:100-151)."""
    from oracle import butd as ob
    from oracle import nic as onic
    from simpleimagecaptionzoo_amd.butd import make_rng
    from simpleimagecaptionzoo_amd.nic import NicHandle
    from simpleimagecaptionzoo_amd.synth import random_nic_params
    En, Hn, Vn, B, T = 512, 512, 2543, 16, 20
    params = random_nic_params(En, Hn, Vn, "cuda", seed=7)
    h = NicHandle(En, Hn, Vn, B, T)
    h.bind(params)
    g = torch.Generator(device="cpu")
    g.manual_seed(11)
    feats_c = torch.randn(B, En, generator=g)
    feats = feats_c.cuda()
    p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in params.items()}
    ids = h.greedy(feats, T).cpu().numpy()
    with torch.no_grad():
        w_ids, w_glog = onic.greedy(feats_c, p, T)
    _excuse_greedy(ids, w_ids, w_glog, 1)
    rs = np.random.RandomState(12)
    om = rs.rand(T, B, Hn) < 0.5
    u = rs.rand(T, B).astype(np.float32)
    rng = make_rng(0, torch.tensor(u, device="cuda"), None, None, torch.tensor(om.astype(np.uint8), device="cuda"))
    seq, lp = h.sample(feats, T, rng)
    seq, lp = seq.cpu().numpy(), lp.cpu().numpy()
    w_seq, w_lp = onic.sample_rl(feats_c, p, u.astype(np.float64), om, T, early_exit=False)
    sdiv = _first_divergence(seq, w_seq.numpy())
    assert (sdiv >= 0).sum() <= 1
    ok = sdiv < 0
    np.testing.assert_allclose(lp[ok], w_lp.detach().numpy()[ok], atol=1e-4)
    rw = (rs.randn(B, 1).astype(np.float32) * ok[:, None]).repeat(T, 1)
    grads = h.new_grads()
    loss, _ = h.sample_backward(torch.tensor(rw, device="cuda"), grads)
    w_loss = ob.reward_criterion(w_lp, torch.from_numpy(np.where(ok[:, None], w_seq.numpy(), seq)), torch.from_numpy(rw))
    w_loss.backward()
    assert abs(loss.item() - w_loss.item()) < 1e-4
    for k, gt in grads.items():
        want = p[k].grad.numpy()
        scale = max(1e-6, float(np.abs(want).max()))
        assert np.abs(gt.cpu().numpy() - want).max() <= 2e-4 * scale + 1e-7, (k, float(np.abs(gt.cpu().numpy() - want).max()), scale)
    h.close()


@pytest.mark.parametrize("regime,sharpen", [("nat", 6.0), ("end_biased", 6.0), ("nat", 1.0)])
def test_fullsize_beam5_128_images_16_checked(regime, sharpen):
    """BASELINE config 3 (beam 5 x 128 images = 640 decoder rows): 16 images per regime against the oracle's one-image beam search
    (BUTD_Model.py:236-318), and one run on un-sharpened random-init weights (sharpen = 1: margins as narrow as they get)."""
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
    imgs = list(range(0, 128, 8))
    if regime == "end_biased":
        ids = h.greedy(feats, steps).cpu().numpy()
        tok = int(np.bincount(ids[ids > 3].ravel()).argmax())
        params["predict.weight_v"][2] = params["predict.weight_v"][tok]
        params["predict.weight_g"][2] = params["predict.weight_g"][tok]
        params["predict.bias"][2] = params["predict.bias"][tok] - 0.2
        h.refresh()
        early = [i for i in range(n_img) if tok in ids[i, :6]]
        imgs = (early + imgs)[:16]
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
    # un-sharpened weights: candidate scores of different beams can tie within fp32 rounding; one image of 16 may take the other branch
    assert len(differ) <= (1 if sharpen == 1.0 else 0), differ
    if regime == "end_biased":
        assert finished >= 1
    h.close()


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
    w_seq, w_lp = oa.sample_rl(feats_c, p, u.astype(np.float64), masks, T, early_exit=False)
    assert np.array_equal(seq.cpu().numpy(), w_seq.numpy())
    np.testing.assert_allclose(lp.cpu().numpy(), w_lp.detach().numpy(), atol=1e-4)
    rw = rs.randn(B, 1).astype(np.float32).repeat(T, 1)
    grads = h.new_grads()
    loss, _ = h.sample_backward(torch.tensor(rw, device="cuda"), grads)
    w_loss = ob.reward_criterion(w_lp, w_seq, torch.from_numpy(rw))
    w_loss.backward()
    assert abs(loss.item() - w_loss.item()) < 1e-4
    compare(grads, p, 3e-4)


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_rank_rehearsal_prints_the_contract_line(scaling):
    """bench.py's N > 1 control flow in fresh child processes (torch.distributed.run, two ranks on the one GPU of the box over
    gloo: ICZ_REHEARSE_ONE_GPU=1): normaliser all-reduce, gradient hook, barriers, MAX over ranks, rank-0 JSON.  The numbers
    mean nothing here; the line's shape and the rank count do."""
    env = dict(os.environ, ICZ_REHEARSE_ONE_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 400) + (0 if scaling == "weak" else 1)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--headline-only",
           "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["ranks_seen"] == 2 and j["scaling"] == scaling and j["steps"] == 2 and j["warmup"] == 1
    assert j["config"]["global_batch"] == (128 if scaling == "weak" else 64) and j["config"]["parallelism"] == "dp2"
    assert j["value"] > 0 and j["unit"] == "captions/s" and j["higher_is_better"] is True and "grad_allreduce_ms" in j
    _check_dp_fields(j)


def _check_dp_fields(j):
    """round 5: the N > 1 line explains itself -- per-phase times (max over ranks) and the same steps with the gradient exchange not
    overlapped with the backward pass"""
    ph = j["phases_ms"]
    assert set(("rollouts", "reward", "backward", "allreduce_exposed", "adam")) <= set(ph) and all(ph[k] >= 0 for k in ph)
    assert ph["rollouts"] > 0 and ph["backward"] > 0 and ph["adam"] > 0
    ov = j["dp_overlap"]
    assert ov["on_ms"] > 0 and ov["off_ms"] > 0 and ov["allreduce_exposed_off_ms"] > 0 and ov["allreduce_exposed_on_ms"] >= 0
    assert j["vs_baseline"] is None and j["vs_reference_in_container"] > 0


@pytest.mark.parametrize("n,extra", [(2, ["--scaling", "weak"]), (5, ["--scaling", "strong", "--batch", "40"])])
def test_bench_starts_its_own_ranks(n, extra):
    """`python bench.py --gpus N` with no launcher and no WORLD_SIZE (the form the driver uses at N = 1): bench.py must start the N
    ranks itself -- fresh child processes, the parent stays off the GPU -- and relay ONE line with n_gpus = ranks_seen = N
    (rehearsal mode: all ranks on the one GPU of the box over gloo).  Five ranks x 8 rows: uneven launch timing, the <= 32-row
    decoder path, four gradient slices reduced from the library's callback under graph replay at ranks >= 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ICZ_REHEARSE_ONE_GPU"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--headline-only"] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    j = json.loads(line[0])
    assert j["n_gpus"] == n and j["ranks_seen"] == n and j["config"]["parallelism"] == "dp%d" % n
    assert j["config"]["global_batch"] == (128 if n == 2 else 40) and j["value"] > 0 and "grad_allreduce_ms" in j
    _check_dp_fields(j)


def test_bench_launcher_kills_the_other_ranks_when_one_dies():
    """A rank that exits early used to leave the others in the rendezvous until the process-group timeout, with their output
    discarded: now the launcher polls every child, kills the rest at the first failure (or at ICZ_BENCH_RANK_TIMEOUT), prints
    every rank's tail and exits non-zero -- within seconds."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ICZ_REHEARSE_ONE_GPU="1", ICZ_BENCH_TEST_FAIL_RANK="1", ICZ_BENCH_RANK_TIMEOUT="300")
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and time.time() - t0 < 120
    assert "rank 1 exited with code" in r.stderr and "---- rank 1" in r.stderr and "told to fail" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_refuses_a_rank_count_that_is_not_the_one_asked_for():
    """--gpus 2 under WORLD_SIZE = 1 used to print a warning and an n_gpus = 1 line; now it is an error before any work starts."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--headline-only"],
                       env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "--gpus 2 but WORLD_SIZE = 1" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
