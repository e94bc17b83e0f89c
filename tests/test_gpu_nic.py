"""GPU parity of the NIC decoder (csrc/nic.hip) against the reference goldens."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def make(g, sd=None, max_rows=16):
    from simpleimagecaptionzoo_amd.nic import NicHandle
    B, H, E, V = [int(x) for x in g["dims"]]
    sd = sd if sd is not None else {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float32, device="cuda") for k, v in sd.items()}
    h = NicHandle(E, H, V, max_rows, 20)
    h.bind(params)
    return h


def regime_sd(g, regime):
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
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


def check_grads(grads, g, prefix):
    for k, v in grads.items():
        want = g[prefix + k]
        got = v.cpu().numpy()
        scale = max(1e-3, float(np.abs(want).max()))
        assert np.abs(got - want).max() <= 2e-4 * scale + 2e-6, (k, np.abs(got - want).max(), scale)


@pytest.mark.parametrize("name", ["nic_dec_tiny", "nic_dec_odd"])
def test_nic_greedy_and_beam_token_exact(golden_dir, name):
    g = load(golden_dir, name)
    feats = torch.tensor(g["feats"], device="cuda")
    h = make(g)
    assert np.array_equal(h.greedy(feats, 20).cpu().numpy(), g["greedy_ids"])
    n = min(feats.shape[0], 3)
    for regime in ("nat", "early", "never", "track"):
        h = make(g, regime_sd(g, regime))
        for k in (1, 3, 5):
            seqs, lens = h.beam_search(feats[:n], k, 50)
            seqs, lens = seqs.cpu().numpy(), lens.cpu().numpy()
            for i in range(n):
                want = g["beam_%s_k%d_i%d" % (regime, k, i)].ravel()
                assert lens[i] == want.shape[0] and np.array_equal(seqs[i, :lens[i]], want), (regime, k, i)


@pytest.mark.parametrize("name", ["nic_dec_tiny", "nic_dec_odd"])
def test_nic_xe_and_reinforce_gradients(golden_dir, name):
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, name)
    feats = torch.tensor(g["feats"], device="cuda")
    h = make(g)
    rng = make_rng(0, None, None, None, torch.tensor(g["xe_out_mask"], device="cuda"))
    logits = h.xe_forward(feats, torch.tensor(g["xe_captions"], device="cuda"), g["xe_lengths"].tolist(), rng, True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss, dfe = h.xe_backward(grads, 0.1, want_dfeats=True)
    assert abs(loss.item() - float(g["xe_loss"])) < 1e-4
    check_grads(grads, g, "xe_grad.")
    np.testing.assert_allclose(dfe.cpu().numpy(), g["xe_dfeats"], atol=2e-5, rtol=2e-4)
    # sample_rl + REINFORCE
    sd = {k[3:]: v.copy() for k, v in g.items() if k.startswith("sd.")}
    sd["predict.bias"][2] = float(g["rl_end_bias"])
    h = make(g, sd)
    rng = make_rng(0, torch.tensor(g["rl_u"], dtype=torch.float32, device="cuda"), None, None, torch.tensor(g["rl_out_mask"], device="cuda"))
    seq, lp = h.sample(feats, 20, rng)
    assert np.array_equal(seq.cpu().numpy(), g["rl_seq"])
    np.testing.assert_allclose(lp.cpu().numpy(), g["rl_logprobs"], atol=1e-4)
    grads = h.new_grads()
    loss, msum, dfe = h.sample_backward(torch.tensor(g["rl_reward"], device="cuda"), grads, want_dfeats=True)
    assert abs(loss.item() - float(g["rl_loss"])) < 1e-4
    check_grads(grads, g, "rl_grad.")
    np.testing.assert_allclose(dfe.cpu().numpy(), g["rl_dfeats"], atol=2e-5, rtol=2e-4)


@pytest.mark.parametrize("name", ["nic_dec_tiny", "nic_dec_odd"])
def test_nic_xe_with_scheduled_sampling(golden_dir, name):
    """NIC DecoderRNN.forward with the decoder's ss_prob = 0.5 (NIC_Model.py:77-89), gate / draw uniforms injected."""
    from simpleimagecaptionzoo_amd.butd import make_rng
    g = load(golden_dir, name)
    feats = torch.tensor(g["feats"], device="cuda")
    h = make(g)
    rng = make_rng(0, None, None, None, torch.tensor(g["xe_out_mask"], device="cuda"))
    caps, lengths = torch.tensor(g["xe_captions"], device="cuda"), g["xe_lengths"].tolist()
    h.set_scheduled_sampling(float(g["ss_prob"]), g["ss_gate"], g["ss_draw"].astype(np.float32))
    logits = h.xe_forward(feats, caps, lengths, rng, True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["ss_packed_logits"], atol=2e-4, rtol=1e-4)
    grads = h.new_grads()
    loss, dfe = h.xe_backward(grads, 0.1, want_dfeats=True)
    assert abs(loss.item() - float(g["ss_loss"])) < 1e-4
    check_grads(grads, g, "ss_grad.")
    np.testing.assert_allclose(dfe.cpu().numpy(), g["ss_dfeats"], atol=2e-5, rtol=2e-4)
    h.set_scheduled_sampling(0.0)
    logits = h.xe_forward(feats, caps, lengths, rng, True, True)
    np.testing.assert_allclose(logits.cpu().numpy(), g["xe_packed_logits"], atol=2e-4, rtol=1e-4)


def test_nic_captioner_state_dict_keys(golden_dir):
    from simpleimagecaptionzoo_amd.nic import NICDecoder_Captioner
    g = load(golden_dir, "nic_dec_tiny")
    B, H, E, V = [int(x) for x in g["dims"]]
    cap = NICDecoder_Captioner(E, H, V, max_batch=8).cuda()
    want = sorted("decoder." + k[3:] for k in g if k.startswith("sd."))
    assert sorted(cap.state_dict().keys()) == want
    cap.load_state_dict({"decoder." + k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")})
    cap.eval()
    ids = cap.sampler({"img_feats": torch.tensor(g["feats"], device="cuda")}, 20)
    assert np.array_equal(ids.cpu().numpy(), g["greedy_ids"])


def test_nic_engine_xe_step_and_scst(golden_dir):
    """NIC_Eng: one training_epoch step = golden XE gradients -> clamp 0.1 -> Adam (oracle restatement); an SCST step and the
    evaluation JSON run on the device."""
    from oracle.butd import Adam
    from simpleimagecaptionzoo_amd.butd import make_rng
    from simpleimagecaptionzoo_amd.engine import NIC_Eng, init_optimizer
    from simpleimagecaptionzoo_amd.synth import document_frequency, synthetic_references
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    g = load(golden_dir, "nic_dec_tiny")
    B, H, E, V = [int(x) for x in g["dims"]]
    vocab = synthetic_vocab(V)
    words = [vocab.ix2word[i] for i in range(V)]
    gts = synthetic_references(B, words, seed=5)
    eng = NIC_Eng({"model_type": "NIC", "embed_dim": E, "hidden_dim": H}, "SYN", vocab, data_dir="/tmp/", device="cuda:0",
                  cider_df=document_frequency(gts), max_batch=8)
    sd0 = {"decoder." + k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")}
    eng.model.load_state_dict(sd0, strict=True)
    feats = torch.tensor(g["feats"], device="cuda")
    lr = 4e-4
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": lr}), lr)
    rng = make_rng(0, None, None, None, torch.tensor(g["xe_out_mask"], device="cuda"))
    batch = (tuple(range(B)), None, torch.tensor(g["xe_captions"]), [int(x) + 1 for x in g["xe_lengths"]], {"img_feats": feats})
    losses = eng.training_epoch([batch], opt, type("C", (), {"smoothing": 0.1})(), tqdm_visible=False, rngs=[rng])
    assert abs(losses[0].item() - float(g["xe_loss"])) < 1e-4
    want = {k: v.clone() for k, v in sd0.items()}
    Adam(want, lr).step({k: torch.tensor(g["xe_grad." + k[len("decoder."):]]) for k in want}, 0.1)
    for k, v in eng.model.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), want[k].numpy(), atol=2e-5, rtol=0, err_msg=k)
    opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5}), 2e-5)
    losses = eng.SCST_training_epoch([(tuple(range(B)), None, gts, {"img_feats": feats})], opt, None, tqdm_visible=False)
    assert np.isfinite(losses[0].item())
    res = eng.eval_captions_json_generation([(tuple(range(B)), None, {"img_feats": feats})], eval_beam_size=3, tqdm_visible=False)
    assert len(res) == B and all(isinstance(r["caption"], str) for r in res)


def test_nic_eval_test_image(golden_dir):
    """NICDecoder_Captioner.eval_test_image (NIC_Model.py:306-331): the caption words of the golden ids and no attention maps."""
    from simpleimagecaptionzoo_amd.nic import NICDecoder_Captioner
    from simpleimagecaptionzoo_amd.vocab import synthetic_vocab
    g = dict(np.load(os.path.join(golden_dir, "nic_dec_tiny.npz")))
    B, H, E, V = [int(x) for x in g["dims"]][:4]
    cap = NICDecoder_Captioner(E, H, V, max_batch=8).cuda()
    cap.load_state_dict({"decoder." + k[3:]: torch.tensor(v) for k, v in g.items() if k.startswith("sd.")})
    cap.eval()
    vocab = synthetic_vocab(V)
    feats = torch.tensor(g["feats"], device="cuda")
    for img in range(2):
        vi = {"img_feats": feats[img:img + 1]}
        for beam, ids in ((-1, g["greedy_ids"][img].tolist()), (3, g["beam_nat_k3_i%d" % img].ravel().astype(int).tolist())):
            words, extra = cap.eval_test_image(vi, vocab, max_len=20, eval_beam_size=beam)
            cut = ids[:ids.index(2)] if 2 in ids else ids
            assert extra == [] and words == [vocab.ix2word[i] for i in cut if i != 1]


# ---- at the benchmark width / handle behaviour (moved here from the per-round files in round 6)
from _fullwidth import (_excuse_greedy, _first_divergence)  # noqa: E402


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
