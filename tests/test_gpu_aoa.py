"""GPU parity of the AoA captioner (csrc/aoa.hip, aoa_train.hip) against the reference goldens (tests/golden/aoa_tiny.npz)."""
import os

import numpy as np
import pytest
import torch

from synth import feats_from_seed

pytestmark = pytest.mark.gpu

MASKS = ("proj", "ref_att", "ref_aoa", "ref_sc", "emb", "ctx", "att", "out")


@pytest.fixture(scope="module")
def g(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "aoa_tiny.npz")))


def dims(g):
    return [int(x) for x in g["dims"]]      # B, R, D, Hd, E, V, NH


def feats_of(g):
    B, R, D = dims(g)[:3]
    return torch.from_numpy(feats_from_seed(int(g["feats_seed"]), B, R, D)).cuda()


def make(g, sd=None, max_rows=16):
    from simpleimagecaptionzoo_amd.aoa import AoaHandle
    B, R, D, Hd, E, V, NH = dims(g)
    sd = sd if sd is not None else {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda") for k, v in sd.items()}
    h = AoaHandle(R, D, Hd, E, V, NH, max_rows, 20)
    h.bind(params)
    return h


def rng_of(g, prefix, uniforms=None):
    from simpleimagecaptionzoo_amd.aoa import make_aoa_rng
    w = dict(zip(MASKS, [int(x) for x in g["mask_widths"]]))
    masks = {k: torch.from_numpy(np.ascontiguousarray(np.unpackbits(g[prefix + k], axis=-1)[..., :w[k]])).cuda() for k in MASKS}
    return make_aoa_rng(0, uniforms, masks)


def regime_sd(g, regime):
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
    if regime == "early":
        sd["decoder.predict.bias"][2] = 4.0
    elif regime == "never":
        sd["decoder.predict.bias"][2] = -1e4
    elif regime == "track":
        tok = int(g["beam_track_tok"])
        sd["decoder.predict.weight_v"][2] = sd["decoder.predict.weight_v"][tok]
        sd["decoder.predict.weight_g"][2] = sd["decoder.predict.weight_g"][tok]
        sd["decoder.predict.bias"][2] = sd["decoder.predict.bias"][tok] - 0.2
    return sd


def check_grads(grads, g, prefix):
    for k, v in grads.items():
        want = g[prefix + k[len("decoder."):]]
        got = v.cpu().numpy()
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 2e-4 * scale + 2e-6, (k, np.abs(got - want).max(), scale)


def test_aoa_refiner_and_greedy(g):
    h = make(g)
    feats = feats_of(g)
    np.testing.assert_allclose(h.refine(feats).cpu().numpy(), g["refined_eval"], atol=5e-5, rtol=1e-4)
    assert np.array_equal(h.greedy(feats, 20).cpu().numpy(), g["greedy_ids"])


def test_aoa_beam_token_exact(g):
    feats = feats_of(g)
    for regime in ("nat", "early", "never", "track"):
        h = make(g, regime_sd(g, regime))
        for k in (1, 3, 5):
            seqs, lens = h.beam_search(feats[:3], k, 50)
            seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
            for i in range(3):
                want = g["beam_%s_k%d_i%d" % (regime, k, i)].ravel()
                assert lens[i] == want.shape[0] and np.array_equal(seqs[i, :lens[i]], want), (regime, k, i)


def test_aoa_xe_logits_loss_and_decoder_grads(g):
    h = make(g)
    feats = feats_of(g)
    logits = h.xe_forward(feats, torch.tensor(g["xe_captions"], device="cuda"), g["xe_lengths"].tolist(), rng_of(g, "xe_mask."), True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss = h.xe_backward(grads, 0.1)
    assert abs(loss.item() - float(g["xe_loss"])) < 1e-4
    check_grads(grads, g, "xe_grad.")


def test_aoa_xe_with_scheduled_sampling(g):
    """AoA_Decoder.forward with the decoder's ss_prob = 0.5 (AoA_Model.py:258-270), gate / draw uniforms injected; the
    Captioner attribute drives the same path."""
    h = make(g)
    feats = feats_of(g)
    caps, lengths = torch.tensor(g["xe_captions"], device="cuda"), g["xe_lengths"].tolist()
    h.set_scheduled_sampling(float(g["ss_prob"]), g["ss_gate"], g["ss_draw"].astype(np.float32))
    logits = h.xe_forward(feats, caps, lengths, rng_of(g, "xe_mask."), True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["ss_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss = h.xe_backward(grads, 0.1)
    assert abs(loss.item() - float(g["ss_loss"])) < 1e-4
    check_grads(grads, g, "ss_grad.")
    h.set_scheduled_sampling(0.0)
    logits = h.xe_forward(feats, caps, lengths, rng_of(g, "xe_mask."), True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)


def test_aoa_sample_rl_and_reinforce_grads(g):
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
    sd["decoder.predict.bias"][2] = float(g["rl_end_bias"])
    h = make(g, sd)
    feats = feats_of(g)
    rng = rng_of(g, "rl_mask.", torch.tensor(g["rl_u"], dtype=torch.float32, device="cuda"))
    seq, lp = h.sample(feats, 20, rng)
    assert np.array_equal(seq.cpu().numpy(), g["rl_seq"])
    np.testing.assert_allclose(lp.cpu().numpy(), g["rl_logprobs"], atol=1e-4)
    grads = h.new_grads()
    loss, msum = h.sample_backward(torch.tensor(g["rl_reward"], device="cuda"), grads)
    assert abs(loss.item() - float(g["rl_loss"])) < 1e-4
    assert abs(msum.item() - h.sample_mask_sum()) < 0.5
    check_grads(grads, g, "rl_grad.")


def test_aoa_philox_paths_run_and_are_reproducible(g):
    """No explicit masks: dropout / sampling bits come from the Philox streams; same seed -> same rollout and gradients."""
    from simpleimagecaptionzoo_amd.aoa import make_aoa_rng
    h = make(g)
    feats = feats_of(g)
    out = []
    for _ in range(2):
        seq, lp = h.sample(feats, 20, make_aoa_rng(1234))
        grads = h.new_grads()
        loss, _ = h.sample_backward(torch.ones(seq.shape, device="cuda"), grads)
        out.append((seq.clone(), lp.clone(), loss.item(), {k: v.clone() for k, v in grads.items()}))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
    for k in out[0][3]:
        assert torch.equal(out[0][3][k], out[1][3][k]), k
        assert torch.isfinite(out[0][3][k]).all()
    seq2, _ = h.sample(feats, 20, make_aoa_rng(99))
    assert not torch.equal(seq2, out[0][0])


def test_aoa_captioner_state_dict_and_engine(g):
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner
    B, R, D, Hd, E, V, NH = dims(g)
    cap = AoADetection_Captioner(V, NH, Hd, E, num_regions=R, enc_dim=D, max_batch=8).cuda()
    want = sorted(k[3:] for k in g if k.startswith("sd."))
    assert sorted(cap.state_dict().keys()) == want
    for k, v in cap.state_dict().items():
        assert tuple(v.shape) == g["sd." + k].shape, k
    cap.load_state_dict({k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")})
    cap.eval()
    vi = {"bu_feats": feats_of(g), "bu_bboxes": None, "bu_masks": None}
    assert np.array_equal(cap.sampler(vi, 20).cpu().numpy(), g["greedy_ids"])
    one = {"bu_feats": feats_of(g)[1:2], "bu_bboxes": None, "bu_masks": None}
    assert np.array_equal(cap.beam_search_sampler(one, 3).cpu().numpy(), g["beam_nat_k3_i1"])
    assert len(cap.get_param_groups({"lr": 1e-4})[0]["params"]) == 18


class _Crit:
    smoothing = 0.1


def _supp(feats):
    return tuple({"bu_feat": feats[i], "bu_bbox": np.zeros((feats.shape[1], 4), np.float32)} for i in range(feats.shape[0]))


@pytest.mark.parametrize("dp", [False, True])
def test_aoa_engine_xe_step_scst_step_and_eval(g, dp):
    """AoADetection_Eng: one training_epoch step = golden XE gradients -> clamp 0.1 -> Adam (oracle restatement) on the decoder
    only; then one SCST step and the evaluation JSON run end to end on the device.  dp: the same through the data-parallel path on
    a one-rank RCCL group (sums over one rank are identities): normaliser all-reduce, the AoA library's gradient-ready callback
    (icz_aoa_set_grad_callback: predict.* before the reverse-time loop, embed + lstm.* before the attention block's weight
    gradients) starting an all-reduce per group, the remainder reduced after the call, clamp + Adam afterwards."""
    import torch.distributed as td
    from simpleimagecaptionzoo_amd import dist as icz_dist
    real = icz_dist.is_distributed
    if dp:
        if not td.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29534")
            td.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        icz_dist.is_distributed = lambda: True
    try:
        eng = _aoa_engine_steps(g)
        if dp:
            assert len(eng._stage_slices) == 3 and eng._hooked is not None and not eng._pending
            lo, hi = eng._stage_slices[0]
            assert hi - lo >= sum(eng._gviews[k].numel() for k in eng._GRAD_STAGES[0])
    finally:
        icz_dist.is_distributed = real


def _aoa_engine_steps(g):
    from oracle.butd import Adam
    from simpleimagecaptionzoo_amd.engine import AoADetection_Eng, init_optimizer
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    B, R, D, Hd, E, V, NH = dims(g)
    vocab = synthetic_vocab(V)
    words = [vocab.ix2word[i] for i in range(V)]
    gts = synthetic_references(B, words, seed=5)
    eng = AoADetection_Eng({"model_type": "AoADetection", "embed_dim": E, "hidden_dim": Hd, "num_heads": NH, "num_regions": R,
                            "enc_dim": D}, "SYN", vocab, data_dir="/tmp/", use_bu="fixed", device="cuda:0",
                           cider_df=document_frequency(gts), max_batch=8)
    sd0 = {k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")}
    eng.model.load_state_dict(sd0, strict=True)
    feats = feats_from_seed(int(g["feats_seed"]), B, R, D)
    lr = 4e-4
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": lr}), lr)
    batch = (tuple(range(B)), None, torch.tensor(g["xe_captions"]), [int(x) + 1 for x in g["xe_lengths"]], _supp(feats))
    losses = eng.training_epoch([batch], opt, _Crit(), tqdm_visible=False, rngs=[rng_of(g, "xe_mask.")])
    assert abs(losses[0].item() - float(g["xe_loss"])) < 1e-4
    dec = {k: v.clone() for k, v in sd0.items() if k.startswith("decoder.")}
    Adam(dec, lr).step({k: torch.tensor(g["xe_grad." + k[len("decoder."):]]) for k in dec}, 0.1)
    after = eng.model.state_dict()
    for k, v in after.items():
        want = dec[k] if k in dec else sd0[k]        # the refiner / projection are not in the optimizer
        # linear_K.bias shifts every score of a softmax row equally: its gradient is identically zero, the reference feeds
        # Adam rounding noise there (|update| <= lr), like BUTD's atten.affine.bias
        tol = lr * 1.01 if k == "decoder.aoa_block.linear_K.bias" else 2e-5
        np.testing.assert_allclose(v.cpu().numpy(), want.numpy(), atol=tol, rtol=0, err_msg=k)
    before = {k: v.clone() for k, v in eng.model.state_dict().items()}
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
    losses = eng.SCST_training_epoch([(tuple(range(B)), None, gts, _supp(feats))], opt, None, tqdm_visible=False)
    assert np.isfinite(losses[0].item())
    now = eng.model.state_dict()
    assert any(not torch.equal(now[k], before[k]) for k in now if k.startswith("decoder."))
    assert all(torch.equal(now[k], before[k]) for k in now if not k.startswith("decoder."))
    res = eng.eval_captions_json_generation([(tuple(range(B)), None, _supp(feats))], eval_beam_size=3, tqdm_visible=False)
    assert len(res) == B and all(isinstance(r["caption"], str) and r["image_id"] == i for i, r in enumerate(res))
    return eng


@pytest.mark.parametrize("cfg", [(3, 36, 96, 64, 32, 101, 8), (5, 20, 64, 128, 64, 203, 8), (2, 49, 128, 96, 48, 77, 4)])
def test_aoa_random_shapes_match_oracle(cfg):
    """Randomly initialised AoA captioners of assorted sizes (heads, regions, widths): refined features, greedy ids and the
    XE gradients of the decoder against the oracle / torch autograd."""
    from oracle import aoa as oa
    from oracle import butd as ob
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner
    B, R, D, Hd, E, V, NH = cfg
    torch.manual_seed(sum(cfg))
    cap = AoADetection_Captioner(V, NH, Hd, E, num_regions=R, enc_dim=D, max_batch=8, max_beam=2).cuda()
    with torch.no_grad():
        cap.decoder.predict.weight_g.mul_(8.0)
        for l in cap.aoa_refine.aoa_layers:
            for p_ in l.parameters():
                p_.add_(torch.randn_like(p_) * 0.02)
    h = cap._handle()
    p = {k: v.detach().cpu().clone() for k, v in cap.state_dict().items()}
    oa_nh, oa.NH = oa.NH, NH
    try:
        feats = torch.relu(torch.randn(B, R, D, device="cuda"))
        np.testing.assert_allclose(h.refine(feats).cpu().numpy(), oa.refine(feats.cpu(), p).numpy(), atol=1e-4, rtol=1e-4)
        T = 4
        want_ids, _ = oa.greedy(feats.cpu(), p, T)
        assert np.array_equal(h.greedy(feats, T).cpu().numpy(), want_ids.numpy())
        lengths = sorted([4, 3, 2, 2, 1][:B], reverse=True)
        caps = torch.zeros(B, 5, dtype=torch.int64)
        rs = np.random.RandomState(sum(cfg))
        for b, n in enumerate(lengths):
            caps[b, 0] = 1
            caps[b, 1:n] = torch.from_numpy(rs.randint(4, V, size=n - 1))
            caps[b, n] = 2
        logits = h.xe_forward(feats, caps.cuda(), lengths, None, train=False, want_logits=True)
        for k in p:
            p[k].requires_grad_(k.startswith("decoder."))
        want_logits = oa.forward_xe(feats.cpu(), caps, lengths, p)
        np.testing.assert_allclose(logits.cpu().numpy(), want_logits.detach().numpy(), atol=3e-4, rtol=1e-4)
        tgt = torch.tensor([caps[b, t + 1] for b, t in ob.packed_order(lengths)])
        loss = ob.label_smoothing_loss(want_logits, tgt, 0.1)
        loss.backward()
        grads = h.new_grads()
        got = h.xe_backward(grads, 0.1)
        assert abs(got.item() - loss.item()) < 1e-4
        for k, g_ in grads.items():
            if k == "decoder.aoa_block.linear_K.bias":
                continue
            want = p[k].grad.numpy()
            scale = max(1e-6, float(np.abs(want).max()))
            assert np.abs(g_.cpu().numpy() - want).max() <= 3e-4 * scale + 1e-7, (k, cfg)
    finally:
        oa.NH = oa_nh


def test_aoa_concurrent_rollouts_equal_sequential(g):
    """icz_aoa_scst_rollouts (greedy on bank 0 / side stream beside the sampled rollout on bank 1) = greedy() then sample()."""
    from simpleimagecaptionzoo_amd.aoa import make_aoa_rng
    h = make(g)
    feats = feats_of(g)
    want_ids = h.greedy(feats, 20).clone()
    want_seq, want_lp = h.sample(feats, 20, make_aoa_rng(77))
    want_seq, want_lp = want_seq.clone(), want_lp.clone()
    for _ in range(3):
        ids, seq, lp = h.rollouts(feats, 20, make_aoa_rng(77))
        assert torch.equal(ids, want_ids) and torch.equal(seq, want_seq) and torch.equal(lp, want_lp)
    grads = h.new_grads()
    loss, _ = h.sample_backward(torch.ones(seq.shape, device="cuda"), grads)
    h.sample(feats, 20, make_aoa_rng(77))
    grads2 = h.new_grads()
    loss2, _ = h.sample_backward(torch.ones(seq.shape, device="cuda"), grads2)
    assert loss.item() == loss2.item() and all(torch.equal(grads[k], grads2[k]) for k in grads)
    assert np.array_equal(ids.cpu().numpy(), g["greedy_ids"])


def test_aoa_eval_test_image_words_and_attention_maps(g):
    """AoADetection_Captioner.eval_test_image (AoA_Model.py:755-786; Engine.py:325,339): the caption of the golden greedy / beam
    ids and the head-averaged decoder attention of every step (:118) -- the oracle's, step by step over the same tokens."""
    from oracle import aoa as oa
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    B, R, D, Hd, E, V, NH = dims(g)
    cap = AoADetection_Captioner(V, NH, Hd, E, num_regions=R, enc_dim=D, max_batch=8).cuda()
    sd = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    cap.load_state_dict({k: torch.tensor(v) for k, v in sd.items()})
    cap.eval()
    vocab = synthetic_vocab(V)
    p = {k: torch.tensor(np.asarray(v), dtype=torch.float32) for k, v in sd.items()}
    oa_nh, oa.NH = oa.NH, NH
    feats = feats_of(g)
    for img in range(2):
        vi = {"bu_feats": feats[img:img + 1], "bu_bboxes": None, "bu_masks": None}
        f1 = feats[img:img + 1].cpu()
        enc = oa.refine(f1, p)
        for beam, ids in ((-1, g["greedy_ids"][img].tolist()), (3, g["beam_nat_k3_i%d" % img].ravel().astype(int).tolist())):
            words, (alphas,) = cap.eval_test_image(vi, vocab, max_len=20, eval_beam_size=beam)
            cut = ids[:ids.index(2)] if 2 in ids else ids
            assert words == [vocab.ix2word[i] for i in cut if i != 1]
            fed = ([1] + ids[:-1]) if beam == -1 else ids[:-1]        # tokens fed step by step: <sta> first
            st, want = oa._zero(1, Hd), []
            for tok in fed:
                _, al, st = oa.dec_step(torch.tensor([tok]), st, enc, enc.mean(1), p)
                want.append(al)
            assert tuple(alphas.shape) == (1, len(fed), R)
            np.testing.assert_allclose(alphas.cpu().numpy()[0], torch.cat(want, 0).numpy(), atol=1e-4)
    oa.NH = oa_nh
