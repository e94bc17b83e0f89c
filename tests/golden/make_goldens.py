#!/usr/bin/env python3
"""Golden-vector generator (runs ONLY in the build container, CPU).

Imports the reference implementation from /root/reference *unmodified* (third-party modules the
hot path never touches are stubbed in sys.modules), drives it on small seeded
inputs and writes input/output vectors as .npz / .json fixtures next to this script.  The fixtures are
data only (tensors, ids, strings, scores); no reference source travels with them.

The patches applied *around* (not inside) the reference while it runs:
  * torch<=1.4 integer-division semantics for `LongTensor / int` in beam search
    (Models/BUTD_Model.py:277, AoA_Model.py:462, NIC_Model.py:181),
  * `torch.nn.functional.dropout` draws its keep-mask from arrays we supply (so the masks can be
    handed to the HIP path), and
  * `torch.multinomial(p, 1)` is an inverse-CDF draw from uniforms we supply,
  * for the scheduled-sampling vectors only: `Tensor.uniform_` (the gate draw of DecoderRNN.forward) returns uniforms we
    supply, and the decoder's own `ss_prob` attribute is set to 0.5 for that run (Engine.py:143 sets the Captioner's).
These replace torch's RNG stream (which no other implementation can reproduce) by explicit
inputs; the arithmetic of the reference is untouched.

Usage:  python tests/golden/make_goldens.py
"""
import contextlib
import json
import os
import pickle
import sys
import tempfile
import types
import warnings
from collections import defaultdict

import numpy as np
import torch

warnings.filterwarnings("ignore")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from synth import feats_from_seed, masks_from_seed, pin_tensor  # noqa: E402
REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _mod(name, **kw):
    m = types.ModuleType(name)
    m.__dict__.update(kw)
    sys.modules[name] = m
    return m


def bootstrap():
    tv = _mod("torchvision")
    tv.models = _mod("torchvision.models")
    tv.transforms = _mod("torchvision.transforms")
    sk = _mod("skimage")
    sk.transform = _mod("skimage.transform")
    sk.io = _mod("skimage.io")
    pc = _mod("pycocotools")
    pc.coco = _mod("pycocotools.coco", COCO=object)
    _mod("nltk")
    _mod("gensim")
    sys.path.insert(0, REF)


@contextlib.contextmanager
def legacy_int_div():
    orig = torch.Tensor.__truediv__

    def td(self, other):
        if not self.is_floating_point() and isinstance(other, int):
            return torch.div(self, other, rounding_mode="floor")
        return orig(self, other)

    torch.Tensor.__truediv__ = td
    try:
        yield
    finally:
        torch.Tensor.__truediv__ = orig


class Injector:
    """Supplies dropout keep-masks and multinomial uniforms in call order."""

    def __init__(self):
        self.masks = None      # list of arrays, one per training-mode dropout call *kind*, see set()
        self.uniforms = None
        self.reset()

    def reset(self):
        self.d_calls = 0
        self.m_calls = 0

    def set(self, masks_per_step, uniforms):
        # masks_per_step: list (len = #dropout calls per step) of arrays [T, B, ...] uint8
        self.masks = masks_per_step
        self.uniforms = uniforms
        self.reset()

    def set_sequential(self, seed, uniforms=None):
        """Masks are drawn on the fly (keep-probability 1-p, seeded) in call order and recorded in self.rec."""
        self.seq_rng = np.random.RandomState(seed)
        self.rec = []
        self.masks = None
        self.uniforms = uniforms
        self.reset()

    def dropout(self, inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        if self.masks is None:
            m = (self.seq_rng.rand(*inp.shape) < (1.0 - p)).astype(np.uint8)
            self.rec.append((float(p), m))
            return inp * torch.from_numpy(m.astype(np.float32)) * (1.0 / (1.0 - p))
        n = len(self.masks)
        t, kind = divmod(self.d_calls, n)
        self.d_calls += 1
        m = self.masks[kind][t]
        b = inp.shape[0]
        m = torch.from_numpy(m[:b].astype(np.float32)).reshape(inp.shape)
        return inp * m * (1.0 / (1.0 - p))

    def multinomial(self, prob, num_samples=1, replacement=False, generator=None):
        assert num_samples == 1
        u = torch.from_numpy(self.uniforms[self.m_calls][: prob.shape[0]].astype(np.float64))
        self.m_calls += 1
        c = torch.cumsum(prob.double(), dim=1)
        tgt = (u * c[:, -1]).unsqueeze(1)
        idx = torch.searchsorted(c, tgt, right=True).clamp_(max=prob.shape[1] - 1)
        return idx.long()


INJ = Injector()


@contextlib.contextmanager
def injected():
    import torch.nn.functional as F
    od, om = F.dropout, torch.multinomial
    F.dropout = INJ.dropout
    torch.multinomial = INJ.multinomial
    try:
        yield
    finally:
        F.dropout = od
        torch.multinomial = om


@contextlib.contextmanager
def scheduled_sampling_draws(u_gate, u_draw):
    """The two random draws of the scheduled-sampling branch of DecoderRNN.forward (BUTD_Model.py:120-130), supplied:
    `torch.zeros(n).uniform_(0, 1)` (once per time step >= 2) returns u_gate[t][:n]; `torch.multinomial(prob_prev, 1)`
    becomes the inverse-CDF draw with u_draw[t] (same stand-in as Injector.multinomial)."""
    state = {"t": 1}
    ou, om = torch.Tensor.uniform_, torch.multinomial

    def uni(self, a=0.0, b=1.0, generator=None):
        state["t"] += 1
        self.copy_(torch.from_numpy(u_gate[state["t"]][: self.shape[0]].astype(np.float32)))
        return self

    def multi(prob, num_samples=1, replacement=False, generator=None):
        assert num_samples == 1
        u = torch.from_numpy(u_draw[state["t"]][: prob.shape[0]].astype(np.float64))
        c = torch.cumsum(prob.double(), dim=1)
        tgt = (u * c[:, -1]).unsqueeze(1)
        return torch.searchsorted(c, tgt, right=True).clamp_(max=prob.shape[1] - 1).long()

    torch.Tensor.uniform_ = uni
    torch.multinomial = multi
    try:
        yield state
    finally:
        torch.Tensor.uniform_ = ou
        torch.multinomial = om


def make_vocab(V):
    from ClassRepository.CaptionVocabClass import Caption_Vocabulary
    v = Caption_Vocabulary()
    for w in ["<pad>", "<sta>", "<end>", "<unk>"]:
        v.add_word(w)
    for i in range(V - 4):
        v.add_word("w%d" % i)
    return v


def sharpen(dec, V, eg=30.0, lg=3.0, hg=0.5, pg=15.0, bs=0.3):
    """Random-init decoders collapse onto one bias-dominated token; rescale the freshly initialised
    weights (inputs to the reference, not its code) so outputs depend on features and tokens and the
    top-2 logit margins sit well above fp32 reordering noise."""
    with torch.no_grad():
        dec.embed[0].weight.mul_(eg)
        for n in ("TD_atten", "language_model"):
            getattr(dec, n).weight_ih.mul_(lg)
            getattr(dec, n).weight_hh.mul_(hg)
        dec.predict.weight_g.mul_(pg)
        dec.predict.bias.copy_(torch.randn(V) * bs)


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def synth_refs(rng, vocab_words, n_img, oov_rate=0.08):
    gts = {}
    for i in range(n_img):
        refs = []
        for _ in range(5):
            L = rng.randint(5, 11)
            z = np.minimum(rng.zipf(1.3, size=L), len(vocab_words)) - 1
            ws = [vocab_words[j] for j in z]
            for p in range(L):
                if rng.rand() < oov_rate:
                    ws[p] = "oov%d" % rng.randint(0, 6)
            refs.append(" ".join(ws))
        gts[i] = refs
    return gts


def build_df(gts_train):
    """Rule of PreProcess/CIDEr_idf_preproccess.py:41-83 applied to synthetic 'train' refs."""
    from cider.pyciderevalcap.ciderD.ciderD_scorer import cook_refs
    df = defaultdict(float)
    for refs in gts_train.values():
        cr = cook_refs(refs)
        for ng in set(ng for r in cr for ng in r.keys()):
            df[ng] += 1
    return {"document_frequency": df, "ref_len": len(gts_train)}


def df_to_json(df):
    return {"ref_len": df["ref_len"],
            "document_frequency": [[list(k), float(v)] for k, v in df["document_frequency"].items()]}


# ------------------------------------------------------------------------------------------------
def gen_butd_decoder(tag, B, R, D, H, E, A, V, seed, cap_len=(6, 13)):
    """G-step, G-greedy, G-beam, G-xe, G-rl on Models/BUTD_Model.py DecoderRNN.  cap_len: range of the XE caption lengths
    (incl. <sta> / <end>); the reference never truncates captions (Datasets.py:47-51)."""
    from Models.BUTD_Model import DecoderRNN
    from Utils import LabelSmoothingLoss, RewardCriterion
    from torch.nn.utils.rnn import pack_padded_sequence
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)
    dec = DecoderRNN(atten_dim=A, embed_dim=E, hidden_dim=H, vocab_size=V, enc_dim=D)
    sharpen(dec, V)
    feats = torch.relu(torch.randn(B, R, D))
    out = {"feats": feats.numpy()}
    out.update({"sd." + k: v for k, v in sd_to_np(dec.state_dict()).items()})

    # ---- G-step: one decoder step from random state (eval mode)
    dec.eval()
    with torch.no_grad():
        h1, c1, h2, c2 = [torch.randn(B, H) * 0.5 for _ in range(4)]
        it = torch.from_numpy(rng.randint(0, V, size=(B,))).long()
        emb = dec.embed(it)
        mean = feats.mean(1)
        nh1, nc1 = dec.TD_atten(torch.cat([h2, mean, emb], 1), (h1, c1))
        ctx, alpha = dec.atten(feats, nh1)
        nh2, nc2 = dec.language_model(torch.cat([ctx, nh1], 1), (h2, c2))
        logits = dec.predict(dec.dropout(nh2))
    out.update(step_h1=h1.numpy(), step_c1=c1.numpy(), step_h2=h2.numpy(), step_c2=c2.numpy(),
               step_it=it.numpy(), step_nh1=nh1.numpy(), step_nc1=nc1.numpy(), step_ctx=ctx.numpy(),
               step_alpha=alpha.numpy(), step_nh2=nh2.numpy(), step_nc2=nc2.numpy(),
               step_logits=logits.numpy())

    # ---- G-greedy (BUTD_Model.py:153-189) with per-step logits recorded through a forward hook
    rec = []
    hk = dec.predict.register_forward_hook(lambda m, i, o: rec.append(o.detach().clone()))
    with torch.no_grad():
        ids, alphas = dec.sample(feats, max_len=20)
    hk.remove()
    out.update(greedy_ids=ids.numpy(), greedy_alphas=alphas.numpy(),
               greedy_logits=torch.stack(rec, 1).numpy())

    # ---- G-beam (BUTD_Model.py:236-318), batch 1, k in {1,3,5}; four regimes for the <end> logit:
    #   nat   : as initialised;  early: bias[<end>] = 4 (finishes at step 1);  never: bias = -1e4 (50-step cap);
    #   track : <end>'s output row := the most frequent greedy token's row, bias 0.2 lower -> <end> enters the
    #           top-k in the middle of a sentence (shrinking k, best-complete selection)
    base_bias = dec.predict.bias.detach().clone()
    base_v2 = dec.predict.weight_v.detach()[2].clone()
    base_g2 = dec.predict.weight_g.detach()[2].clone()
    tok = int(np.bincount(ids.numpy().ravel()).argmax())
    out["beam_track_tok"] = np.int64(tok)
    for regime, end_bias in (("nat", None), ("early", 4.0), ("never", -1e4), ("track", None)):
        with torch.no_grad():
            dec.predict.bias.copy_(base_bias)
            if end_bias is not None:
                dec.predict.bias[2] = end_bias
            if regime == "track":
                dec.predict.weight_v[2] = dec.predict.weight_v[tok]
                dec.predict.weight_g[2] = dec.predict.weight_g[tok]
                dec.predict.bias[2] = dec.predict.bias[tok] - 0.2
        for k in (1, 3, 5):
            for img in range(min(B, 3)):
                with torch.no_grad(), legacy_int_div():
                    seq, al = dec.beam_search_sample(feats[img:img + 1], beam_size=k)
                key = "beam_%s_k%d_i%d" % (regime, k, img)
                out[key] = np.asarray(seq.numpy(), dtype=np.float32)
                out[key + "_alpha"] = al.numpy().astype(np.float32)
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
        dec.predict.weight_v[2] = base_v2
        dec.predict.weight_g[2] = base_g2
    out["beam_end_bias"] = np.array([np.nan, 4.0, -1e4, np.nan], dtype=np.float32)

    # ---- G-xe (BUTD_Model.py:97-151 + Utils.py:268-286 + Engine.py:178-187), train mode, injected masks
    lengths_full = sorted(rng.randint(cap_len[0], cap_len[1], size=B).tolist(), reverse=True)  # incl. <sta> and <end>
    L = max(lengths_full)
    caps = np.zeros((B, L), dtype=np.int64)
    for b, l in enumerate(lengths_full):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
        caps[b, l - 1] = 2
    captions = torch.from_numpy(caps)
    lengths = [l - 1 for l in lengths_full]
    T = max(lengths)
    xe_emb_mask = (rng.rand(T, B, E) < 0.5).astype(np.uint8)
    xe_att_mask = (rng.rand(T, B, R, A) < 0.5).astype(np.uint8)
    xe_out_mask = (rng.rand(T, B, H) < 0.5).astype(np.uint8)
    dec.train()
    dec.zero_grad()
    INJ.set([xe_emb_mask, xe_att_mask, xe_out_mask], None)
    with injected():
        packed, xe_alphas = dec(feats, captions, lengths)
    targets = pack_padded_sequence(captions[:, 1:], lengths, batch_first=True)
    crit = LabelSmoothingLoss(smoothing=0.1)
    loss = crit(packed[0], targets[0])
    loss.backward()
    out.update(xe_captions=caps, xe_lengths=np.array(lengths), xe_emb_mask=xe_emb_mask,
               xe_att_mask=np.packbits(xe_att_mask, axis=-1), xe_out_mask=xe_out_mask,
               xe_packed_logits=packed[0].detach().numpy(), xe_packed_targets=targets[0].numpy(),
               xe_batch_sizes=packed[1].numpy(), xe_loss=np.float32(loss.item()))
    for n_, p in dec.named_parameters():
        out["xe_grad." + n_] = p.grad.detach().numpy().copy()
    # a second loss value with smoothing 0 (plain XE through the same class)
    out["xe_loss_s0"] = np.float32(LabelSmoothingLoss(0.0)(packed[0].detach(), targets[0]).item())

    # ---- G-ss: the same XE step with scheduled sampling switched on in the decoder (BUTD_Model.py:120-132; Engine.py:143
    #      sets the attribute on the Captioner, which the decoder never reads -- here it is set where it is read).
    #      Own random stream, so that every other array of this file stays what it was.
    ss_rs = np.random.RandomState(seed + 4242)
    ss_gate = ss_rs.rand(T, B).astype(np.float32)
    ss_draw = ss_rs.rand(T, B)
    dec.train()
    dec.zero_grad()
    dec.ss_prob = 0.5
    INJ.set([xe_emb_mask, xe_att_mask, xe_out_mask], None)
    toks = []
    hk = dec.embed.register_forward_hook(lambda m, i, o: toks.append(i[0].detach().clone()))
    with injected(), scheduled_sampling_draws(ss_gate, ss_draw):
        ss_packed, _ = dec(feats, captions, lengths)
    hk.remove()
    dec.ss_prob = 0.0
    ss_loss = crit(ss_packed[0], targets[0])
    ss_loss.backward()
    ss_tok = np.zeros((T, B), dtype=np.int64)
    for t_, it_ in enumerate(toks):
        ss_tok[t_, : it_.shape[0]] = it_.numpy()
    out.update(ss_prob=np.float32(0.5), ss_gate=ss_gate, ss_draw=ss_draw, ss_tokens=ss_tok,
               ss_packed_logits=ss_packed[0].detach().numpy(), ss_loss=np.float32(ss_loss.item()))
    for n_, p in dec.named_parameters():
        out["ss_grad." + n_] = p.grad.detach().numpy().copy()

    # ---- G-rl (BUTD_Model.py:191-234 + Utils.py:295-317), injected masks + uniforms
    T = 20
    rl_emb_mask = (rng.rand(T, B, E) < 0.5).astype(np.uint8)
    rl_att_mask = (rng.rand(T, B, R, A) < 0.5).astype(np.uint8)
    rl_out_mask = (rng.rand(T, B, H) < 0.5).astype(np.uint8)
    rl_u = rng.rand(T, B)
    # steer a couple of rows into emitting <end> early so the finished-mask path is exercised
    with torch.no_grad():
        dec.predict.bias[2] = 2.5
    out["rl_end_bias"] = np.float32(2.5)
    dec.train()
    dec.zero_grad()
    INJ.set([rl_emb_mask, rl_att_mask, rl_out_mask], rl_u)
    rec = []
    hk = dec.predict.register_forward_hook(lambda m, i, o: rec.append(o.detach().clone()))
    with injected():
        seq, slp = dec.sample_rl(feats, max_len=T)
    hk.remove()
    steps_run = len(rec)
    reward = torch.from_numpy(rng.randn(B, 1).astype(np.float32)).repeat(1, T)
    rl_loss = RewardCriterion()(slp, seq, reward)
    rl_loss.backward()
    out.update(rl_emb_mask=rl_emb_mask, rl_att_mask=np.packbits(rl_att_mask, axis=-1),
               rl_out_mask=rl_out_mask, rl_u=rl_u, rl_seq=seq.numpy(), rl_logprobs=slp.detach().numpy(),
               rl_logits=torch.stack(rec, 1).numpy(), rl_steps_run=np.int64(steps_run),
               rl_reward=reward.numpy(), rl_loss=np.float32(rl_loss.item()))
    for n_, p in dec.named_parameters():
        out["rl_grad." + n_] = p.grad.detach().numpy().copy()
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
    out["dims"] = np.array([B, R, D, H, E, A, V], dtype=np.int64)
    save(tag, **out)


# ------------------------------------------------------------------------------------------------
def gen_nic_decoder(tag, B, H, E, V, seed):
    """NIC single-LSTM decoder (Models/NIC_Model.py:39-212): greedy, beam, XE, sample_rl + REINFORCE."""
    from Models.NIC_Model import DecoderRNN
    from Utils import LabelSmoothingLoss, RewardCriterion
    from torch.nn.utils.rnn import pack_padded_sequence
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)
    dec = DecoderRNN(embed_dim=E, hidden_dim=H, vocab_size=V)
    with torch.no_grad():            # decisive outputs (see sharpen())
        dec.embed.weight.mul_(4.0)
        dec.lstm.weight_ih.mul_(4.0)
        dec.lstm.weight_hh.mul_(0.5)
        dec.predict.weight_g.mul_(20.0)
        dec.predict.bias.copy_(torch.randn(V) * 0.3)
    feats = torch.randn(B, E)
    out = {"feats": feats.numpy(), "dims": np.array([B, H, E, V], dtype=np.int64)}
    out.update({"sd." + k: v for k, v in sd_to_np(dec.state_dict()).items()})
    dec.eval()
    rec = []
    hk = dec.predict.register_forward_hook(lambda m, i, o: rec.append(o.detach().clone()))
    with torch.no_grad():
        ids = dec.sample(feats, max_len=20)
    hk.remove()
    out.update(greedy_ids=ids.numpy(), greedy_logits=torch.stack(rec, 1).numpy())
    base_bias = dec.predict.bias.detach().clone()
    base_v2 = dec.predict.weight_v.detach()[2].clone()
    base_g2 = dec.predict.weight_g.detach()[2].clone()
    tok = int(np.bincount(ids.numpy().ravel()).argmax())
    out["beam_track_tok"] = np.int64(tok)
    for regime, end_bias in (("nat", None), ("early", 4.0), ("never", -1e4), ("track", None)):
        with torch.no_grad():
            dec.predict.bias.copy_(base_bias)
            if end_bias is not None:
                dec.predict.bias[2] = end_bias
            if regime == "track":
                dec.predict.weight_v[2] = dec.predict.weight_v[tok]
                dec.predict.weight_g[2] = dec.predict.weight_g[tok]
                dec.predict.bias[2] = dec.predict.bias[tok] - 0.2
        for k in (1, 3, 5):
            for img in range(min(B, 3)):
                with torch.no_grad(), legacy_int_div():
                    seq = dec.beam_search_sample(feats[img:img + 1], beam_size=k)
                out["beam_%s_k%d_i%d" % (regime, k, img)] = np.asarray(seq.numpy(), dtype=np.float32)
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
        dec.predict.weight_v[2] = base_v2
        dec.predict.weight_g[2] = base_g2
    # XE
    lengths_full = sorted(rng.randint(6, 13, size=B).tolist(), reverse=True)
    L = max(lengths_full)
    caps = np.zeros((B, L), dtype=np.int64)
    for b, l in enumerate(lengths_full):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
        caps[b, l - 1] = 2
    captions = torch.from_numpy(caps)
    lengths = [l - 1 for l in lengths_full]
    T = max(lengths)
    xe_out_mask = (rng.rand(T, B, H) < 0.5).astype(np.uint8)
    dec.train()
    dec.zero_grad()
    f_xe = feats.clone().requires_grad_(True)
    INJ.set([xe_out_mask], None)
    with injected():
        packed = dec(f_xe, captions, lengths)
    targets = pack_padded_sequence(captions[:, 1:], lengths, batch_first=True)
    loss = LabelSmoothingLoss(smoothing=0.1)(packed[0], targets[0])
    loss.backward()
    out.update(xe_captions=caps, xe_lengths=np.array(lengths), xe_out_mask=xe_out_mask,
               xe_packed_logits=packed[0].detach().numpy(), xe_loss=np.float32(loss.item()),
               xe_dfeats=f_xe.grad.numpy().copy())
    for n_, p in dec.named_parameters():
        out["xe_grad." + n_] = p.grad.detach().numpy().copy()
    # XE with scheduled sampling switched on in the decoder (NIC_Model.py:79-89), own random stream
    ss_rs = np.random.RandomState(seed + 4242)
    ss_gate = ss_rs.rand(T, B).astype(np.float32)
    ss_draw = ss_rs.rand(T, B)
    dec.train()
    dec.zero_grad()
    dec.ss_prob = 0.5
    f_ss = feats.clone().requires_grad_(True)
    INJ.set([xe_out_mask], None)
    toks = []
    hk = dec.embed.register_forward_hook(lambda m, i, o: toks.append(i[0].detach().clone()))
    with injected(), scheduled_sampling_draws(ss_gate, ss_draw):
        ss_packed = dec(f_ss, captions, lengths)
    hk.remove()
    dec.ss_prob = 0.0
    ss_loss = LabelSmoothingLoss(smoothing=0.1)(ss_packed[0], targets[0])
    ss_loss.backward()
    ss_tok = np.zeros((T, B), dtype=np.int64)
    for t_, it_ in enumerate(toks):
        ss_tok[t_, : it_.shape[0]] = it_.numpy()
    out.update(ss_prob=np.float32(0.5), ss_gate=ss_gate, ss_draw=ss_draw, ss_tokens=ss_tok,
               ss_packed_logits=ss_packed[0].detach().numpy(), ss_loss=np.float32(ss_loss.item()),
               ss_dfeats=f_ss.grad.numpy().copy())
    for n_, p in dec.named_parameters():
        out["ss_grad." + n_] = p.grad.detach().numpy().copy()
    # sample_rl
    T = 20
    rl_out_mask = (rng.rand(T, B, H) < 0.5).astype(np.uint8)
    rl_u = rng.rand(T, B)
    with torch.no_grad():
        dec.predict.bias[2] = 2.5
    out["rl_end_bias"] = np.float32(2.5)
    dec.train()
    dec.zero_grad()
    f_rl = feats.clone().requires_grad_(True)
    INJ.set([rl_out_mask], rl_u)
    with injected():
        seq, slp = dec.sample_rl(f_rl, max_len=T)
    reward = torch.from_numpy(rng.randn(B, 1).astype(np.float32)).repeat(1, T)
    rl_loss = RewardCriterion()(slp, seq, reward)
    rl_loss.backward()
    out.update(rl_out_mask=rl_out_mask, rl_u=rl_u, rl_seq=seq.numpy(), rl_logprobs=slp.detach().numpy(),
               rl_reward=reward.numpy(), rl_loss=np.float32(rl_loss.item()), rl_dfeats=f_rl.grad.numpy().copy())
    for n_, p in dec.named_parameters():
        out["rl_grad." + n_] = p.grad.detach().numpy().copy()
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
    save(tag, **out)


# ------------------------------------------------------------------------------------------------
def gen_aoa(tag, B, Hd, E, V, seed, counts=None):
    """AoADetection_Captioner (Models/AoA_Model.py:657-753): feature projection + 6-layer AoA refiner (eval and train
    mode), decoder greedy / beam / XE / sample_rl with REINFORCE gradients of the decoder parameters (the only ones the
    reference optimises, AoA_Model.py:669-674).  counts = regions per image of an 'adaptive' batch: features padded with
    zero rows to max(counts) and prefix bu_masks, as AoA_Engine.modify_visual_inputs builds them (AoA_Engine.py:33-46)."""
    from Models.AoA_Model import AoADetection_Captioner, pack_wrapper
    from Utils import LabelSmoothingLoss, RewardCriterion
    from torch.nn.utils.rnn import pack_padded_sequence
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)
    R, D, NH = (max(counts) if counts else 36), 2048, 8
    m = AoADetection_Captioner(vocab_size=V, num_heads=NH, hidden_dim=Hd, embed_dim=E)
    dec = m.decoder
    with torch.no_grad():
        dec.embed[0].weight.mul_(30.0)
        dec.lstm.weight_ih.mul_(3.0)
        dec.lstm.weight_hh.mul_(0.5)
        dec.predict.weight_g.mul_(15.0)
        dec.predict.bias.copy_(torch.randn(V) * 0.3)
        for name, prm in m.named_parameters():          # non-trivial LayerNorm gains / biases
            if name.endswith("norm.gain") or name.endswith("h_norm.gain"):
                prm.add_(torch.randn_like(prm) * 0.2)
            if name.endswith("norm.bias") or name.endswith("h_norm.bias"):
                prm.add_(torch.randn_like(prm) * 0.1)
    fseed = seed * 1000 + 1
    feats = torch.from_numpy(feats_from_seed(fseed, B, R, D))
    out = {"dims": np.array([B, R, D, Hd, E, V, NH], dtype=np.int64), "feats_seed": np.int64(fseed)}
    bu_masks = None
    if counts:
        assert len(counts) == B and len(set(counts)) == B       # distinct: the reference's length sort has no tie rule
        bu_masks = (torch.arange(R).unsqueeze(0) < torch.tensor(counts).view(-1, 1)).float()
        feats = feats * bu_masks.unsqueeze(-1)
        out["region_counts"] = np.array(counts, dtype=np.int32)
    out.update({"sd." + k: v for k, v in sd_to_np(m.state_dict()).items()})
    vi = {"bu_feats": feats, "bu_bboxes": None, "bu_masks": bu_masks}
    m.eval()
    with torch.no_grad():
        refined = m.aoa_refine(pack_wrapper(m.img_feats_porjection, feats, bu_masks), bu_masks)
    out["refined_eval"] = refined.numpy()
    rec = []
    hk = dec.predict.register_forward_hook(lambda mod, i, o: rec.append(o.detach().clone()))
    with torch.no_grad():
        ids = m.sampler(vi, max_len=20)
    hk.remove()
    out.update(greedy_ids=ids.numpy(), greedy_logits=torch.stack(rec, 1).numpy())
    base_bias = dec.predict.bias.detach().clone()
    base_v2 = dec.predict.weight_v.detach()[2].clone()
    base_g2 = dec.predict.weight_g.detach()[2].clone()
    tok = int(np.bincount(ids.numpy().ravel()).argmax())
    out["beam_track_tok"] = np.int64(tok)
    for regime, end_bias in (("nat", None), ("early", 4.0), ("never", -1e4), ("track", None)):
        with torch.no_grad():
            dec.predict.bias.copy_(base_bias)
            if end_bias is not None:
                dec.predict.bias[2] = end_bias
            if regime == "track":
                dec.predict.weight_v[2] = dec.predict.weight_v[tok]
                dec.predict.weight_g[2] = dec.predict.weight_g[tok]
                dec.predict.bias[2] = dec.predict.bias[tok] - 0.2
        for k in (1, 3, 5):
            for img in range(min(B, 3)):
                # beam search runs one image at a time: its batch is never padded, so bu_masks is None there (AoA_Engine.py:41-42)
                one = {"bu_feats": feats[img:img + 1, :counts[img]] if counts else feats[img:img + 1], "bu_bboxes": None, "bu_masks": None}
                with torch.no_grad(), legacy_int_div():
                    seq = m.beam_search_sampler(one, beam_size=k)
                out["beam_%s_k%d_i%d" % (regime, k, img)] = np.asarray(seq.numpy(), dtype=np.float32)
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
        dec.predict.weight_v[2] = base_v2
        dec.predict.weight_g[2] = base_g2

    def split_masks(recs, T):
        """recorded (p, mask) list in call order -> named arrays (bit-packed along the last axis)."""
        it = iter(recs)
        d = {"proj": next(it)[1]}
        if counts:      # the projection ran on the packed valid rows (pack_wrapper): region-major over the images sorted by
            order = np.argsort(-np.asarray(counts), kind="stable")        # decreasing count -> back to the padded [B, R, Hd]
            padded = np.zeros((B, R, Hd), dtype=np.uint8)
            row = 0
            for r in range(R):
                for b in order:
                    if counts[b] > r:
                        padded[b, r] = d["proj"][row]
                        row += 1
            assert row == d["proj"].shape[0] == sum(counts)
            d["proj"] = padded
        ra, rg, rs = [], [], []
        for _ in range(6):
            ra.append(next(it)[1]); rg.append(next(it)[1]); rs.append(next(it)[1])
        d["ref_att"], d["ref_aoa"], d["ref_sc"] = np.stack(ra), np.stack(rg), np.stack(rs)
        e, c, a, o = [], [], [], []
        for _ in range(T):
            e.append(next(it)[1]); c.append(next(it)[1]); a.append(next(it)[1]); o.append(next(it)[1])
        def pad(lst):      # XE: the batch shrinks with t -> pad rows to B
            full = np.zeros((T,) + (B,) + lst[0].shape[1:], dtype=np.uint8)
            for t_, x in enumerate(lst):
                full[t_, :x.shape[0]] = x
            return full
        d["emb"], d["ctx"], d["att"], d["out"] = pad([x.reshape(x.shape[0], -1) for x in e]), pad(c), pad(a), pad(o)
        assert next(it, None) is None
        return {k: np.packbits(v, axis=-1) for k, v in d.items()}, {k: v.shape[-1] for k, v in d.items()}

    # ---- XE (train mode, every dropout site injected)
    lengths_full = sorted(rng.randint(6, 13, size=B).tolist(), reverse=True)
    L = max(lengths_full)
    caps = np.zeros((B, L), dtype=np.int64)
    for b, l in enumerate(lengths_full):
        caps[b, 0] = 1
        caps[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
        caps[b, l - 1] = 2
    captions = torch.from_numpy(caps)
    lengths = [l - 1 for l in lengths_full]
    m.train()
    m.zero_grad()
    INJ.set_sequential(seed * 7 + 1)
    with injected():
        packed = m(vi, captions, lengths)
    targets = pack_padded_sequence(captions[:, 1:], lengths, batch_first=True)
    loss = LabelSmoothingLoss(smoothing=0.1)(packed[0], targets[0])
    loss.backward()
    mk, widths = split_masks(INJ.rec, max(lengths))
    out.update({"xe_mask." + k: v for k, v in mk.items()})
    out["mask_widths"] = np.array([widths[k] for k in ("proj", "ref_att", "ref_aoa", "ref_sc", "emb", "ctx", "att", "out")])
    out.update(xe_captions=caps, xe_lengths=np.array(lengths), xe_packed_logits=packed[0].detach().numpy(),
               xe_loss=np.float32(loss.item()))
    for n_, p_ in dec.named_parameters():
        out["xe_grad." + n_] = p_.grad.detach().numpy().copy()
    # ---- XE with scheduled sampling switched on in the decoder (AoA_Model.py:260-270); the dropout masks are the XE run's
    #      (same seed, same call order), the gate / draw uniforms come from their own stream
    T_xe = max(lengths)
    ss_rs = np.random.RandomState(seed + 4242)
    ss_gate = ss_rs.rand(T_xe, B).astype(np.float32)
    ss_draw = ss_rs.rand(T_xe, B)
    m.train()
    m.zero_grad()
    dec.ss_prob = 0.5
    INJ.set_sequential(seed * 7 + 1)
    toks = []
    hk = dec.embed.register_forward_hook(lambda mod, i, o: toks.append(i[0].detach().clone()))
    with injected(), scheduled_sampling_draws(ss_gate, ss_draw):
        ss_packed = m(vi, captions, lengths)
    hk.remove()
    dec.ss_prob = 0.0
    ss_mk, _ = split_masks(INJ.rec, T_xe)
    assert all(np.array_equal(ss_mk[k], mk[k]) for k in mk)
    ss_loss = LabelSmoothingLoss(smoothing=0.1)(ss_packed[0], targets[0])
    ss_loss.backward()
    ss_tok = np.zeros((T_xe, B), dtype=np.int64)
    for t_, it_ in enumerate(toks):
        ss_tok[t_, : it_.shape[0]] = it_.numpy()
    out.update(ss_prob=np.float32(0.5), ss_gate=ss_gate, ss_draw=ss_draw, ss_tokens=ss_tok,
               ss_packed_logits=ss_packed[0].detach().numpy(), ss_loss=np.float32(ss_loss.item()))
    for n_, p_ in dec.named_parameters():
        out["ss_grad." + n_] = p_.grad.detach().numpy().copy()
    # ---- sample_rl + REINFORCE
    T = 20
    rl_u = rng.rand(T, B)
    with torch.no_grad():
        dec.predict.bias[2] = 2.5
    out["rl_end_bias"] = np.float32(2.5)
    m.train()
    m.zero_grad()
    INJ.set_sequential(seed * 7 + 2, rl_u)
    rec = []
    hk = dec.predict.register_forward_hook(lambda mod, i, o: rec.append(o.detach().clone()))
    with injected():
        seq, slp = m.sampler_rl(vi, max_len=T)
    hk.remove()
    steps_run = len(rec)
    # the rollout may stop early (all finished): pad the recorded per-step masks with zeros up to T
    recs = list(INJ.rec)
    while len(recs) < 19 + 4 * T:
        recs.append((0.5, np.zeros_like(recs[19 + (len(recs) - 19) % 4][1])))
    mk, _ = split_masks(recs, T)
    out.update({"rl_mask." + k: v for k, v in mk.items()})
    reward = torch.from_numpy(rng.randn(B, 1).astype(np.float32)).repeat(1, T)
    rl_loss = RewardCriterion()(slp, seq, reward)
    rl_loss.backward()
    out.update(rl_u=rl_u, rl_seq=seq.numpy(), rl_logprobs=slp.detach().numpy(), rl_steps_run=np.int64(steps_run),
               rl_reward=reward.numpy(), rl_loss=np.float32(rl_loss.item()))
    for n_, p_ in dec.named_parameters():
        out["rl_grad." + n_] = p_.grad.detach().numpy().copy()
    with torch.no_grad():
        dec.predict.bias.copy_(base_bias)
    save(tag, **out)


# ------------------------------------------------------------------------------------------------
def gen_cider(tag, seed):
    """G-cider: CiderD.compute_score (ciderD.py:30-55) on hand-made edge cases + abstract48S sample."""
    from cider.pyciderevalcap.ciderD.ciderD import CiderD
    rng = np.random.RandomState(seed)
    words = ["<pad>", "<sta>", "<end>", "<unk>"] + ["w%d" % i for i in range(46)]
    train = synth_refs(rng, words, 200)
    df = build_df(train)
    gts = synth_refs(rng, words, 12)
    # edge-case references
    gts[100] = ["oov1 oov2 oov3", "oov4 oov1 oov2 oov3 oov5"]            # OOV-only refs
    gts[101] = ["w1 w1 w1 w1 w1 w1", "w1 w2 w1 w2 w1 w2", "w1"]             # repeats -> clipping
    gts[102] = ["w3 w4 w5 w6 w7 w8 w9 w10 w11 w12 w13 w14 w15 w16 w17 w18"]  # long ref -> large delta
    gts[103] = ["w5"]                                                        # one-word ref
    res = []
    for i in range(12):
        L = rng.randint(1, 21)
        z = np.minimum(rng.zipf(1.3, size=L), len(words)) - 1
        res.append({"image_id": i, "caption": [" ".join(words[j] for j in z)]})
    res += [
        {"image_id": 0, "caption": [""]},                      # empty hypothesis
        {"image_id": 1, "caption": ["<pad>"]},                 # all-zero sampled row (Utils.py:338-346)
        {"image_id": 2, "caption": ["w0"]},                    # one word
        {"image_id": 3, "caption": [gts[3][0]]},               # exact copy of a reference
        {"image_id": 100, "caption": ["w1 w2 w3"]},
        {"image_id": 100, "caption": ["<unk> <unk> <unk>"]},
        {"image_id": 101, "caption": ["w1 w1 w1 w1 w1 w1 w1 w1 w1 w1"]},
        {"image_id": 101, "caption": ["w1 w2 w1 w2"]},
        {"image_id": 102, "caption": ["w3 w4"]},
        {"image_id": 102, "caption": ["w3 w4 w5 w6 w7 w8 w9 w10 w11 w12 w13 w14 w15 w16 w17 w18 w19 w20 w21 w22"]},
        {"image_id": 103, "caption": ["w5"]},
        {"image_id": 103, "caption": ["w5 w5"]},
        {"image_id": 4, "caption": ["zz yy xx"]},              # n-grams unseen in df
    ]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "cider", "data"))
        with open(os.path.join(td, "cider", "data", "SYN-train.p"), "wb") as f:
            pickle.dump(df, f, protocol=2)
        os.chdir(td)
        try:
            score, scores = CiderD(df="SYN-train").compute_score(gts, res)
        finally:
            os.chdir(cwd)
    fx = {"df": df_to_json(df), "gts": {str(k): v for k, v in gts.items()}, "res": res,
          "score": float(score), "scores": [float(s) for s in scores]}

    # realistic sentences: abstract48S refs / candsB, whitespace tokenised + lower-cased, df from the
    # same 500 images (rule of CIDEr_idf_preproccess.py); first 60 candidates only (size)
    refs = json.load(open(os.path.join(REF, "cider/data/abstract48S.json")))
    cands = json.load(open(os.path.join(REF, "cider/data/abstract_candsB.json")))
    g2 = defaultdict(list)
    for r in refs:
        g2[r["image_id"]].append(r["caption"].lower())
    cand_ids = []
    r2 = []
    for c in cands[:60]:
        r2.append({"image_id": c["image_id"], "caption": [c["caption"].lower()]})
        cand_ids.append(c["image_id"])
    g2s = {k: v[:5] for k, v in g2.items() if k in set(cand_ids)}
    df2 = build_df({k: v[:5] for k, v in g2.items()})
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "cider", "data"))
        with open(os.path.join(td, "cider", "data", "ABS-train.p"), "wb") as f:
            pickle.dump(df2, f, protocol=2)
        os.chdir(td)
        try:
            score2, scores2 = CiderD(df="ABS-train").compute_score(g2s, r2)
        finally:
            os.chdir(cwd)
    # only the df entries that any hyp/ref n-gram can touch are needed to reproduce the scores
    from cider.pyciderevalcap.ciderD.ciderD_scorer import precook
    need = set()
    for r in r2:
        need |= set(precook(r["caption"][0]).keys())
    for v in g2s.values():
        for s in v:
            need |= set(precook(s).keys())
    df2s = {"ref_len": df2["ref_len"],
            "document_frequency": {k: v for k, v in df2["document_frequency"].items() if k in need}}
    fx["abstract"] = {"df": df_to_json(df2s), "gts": g2s, "res": r2, "score": float(score2),
                      "scores": [float(s) for s in scores2]}
    with open(os.path.join(OUT, tag + ".json"), "w") as f:
        json.dump(fx, f)
    print("wrote", tag + ".json", "%.1f KB" % (os.path.getsize(os.path.join(OUT, tag + ".json")) / 1024))


# ------------------------------------------------------------------------------------------------
def gen_corpus_cider(tag, seed):
    """Corpus-level CIDEr of the evaluation path (coco_caption/pycocoevalcap/cider/cider.py:34-56 ->
    cider_scorer.py:96-195; called by COCOEvalCap.evaluate, eval.py:24-69): df and log(ref_len) come from the evaluated
    references themselves.  Inputs are already tokenised strings (the Java PTB tokeniser is not runnable here)."""
    from coco_caption.pycocoevalcap.cider.cider import Cider
    rng = np.random.RandomState(seed)
    words = ["w%d" % i for i in range(60)]
    cases = {}
    # synthetic corpus: Zipf references with OOV words, hypotheses of every length 0..20, duplicates, copies of references
    gts = synth_refs(rng, words, 40)
    res = {}
    for i in range(40):
        L = i % 21
        z = np.minimum(rng.zipf(1.3, size=L), len(words)) - 1
        res[i] = [" ".join(words[j] for j in z)]
    res[3] = [gts[3][0]]
    res[4] = [gts[4][1] + " " + gts[4][1]]
    res[5] = ["never seen words here"]
    cases["synthetic"] = (gts, res)
    # one image only: log(ref_len) = 0 and every idf is 0
    cases["single"] = ({7: ["w1 w2 w3 w4", "w2 w3 w4 w5"]}, {7: ["w1 w2 w3 w4"]})
    # realistic sentences (abstract48S references / candidates, lower-cased, whitespace tokens), first 80 images with candidates
    refs = json.load(open(os.path.join(REF, "cider/data/abstract48S.json")))
    cands = json.load(open(os.path.join(REF, "cider/data/abstract_candsB.json")))
    g2 = defaultdict(list)
    for r in refs:
        g2[r["image_id"]].append(r["caption"].lower())
    r2 = {}
    for c in cands:
        if c["image_id"] in g2 and c["image_id"] not in r2 and len(r2) < 80:
            r2[c["image_id"]] = [c["caption"].lower()]
    cases["abstract80"] = ({k: g2[k][:5] for k in r2}, r2)
    fx = {}
    for name, (g, r) in cases.items():
        score, scores = Cider().compute_score(g, r)
        fx[name] = {"ids": [str(k) for k in g.keys()], "gts": {str(k): v for k, v in g.items()},
                    "res": {str(k): v for k, v in r.items()}, "score": float(score).hex(),
                    "scores": [float(x).hex() for x in scores]}
        print(tag, name, "CIDEr %.4f over %d images" % (score, len(g)))
    with open(os.path.join(OUT, tag + ".json"), "w") as f:
        json.dump(fx, f)
    print("wrote", tag + ".json", "%.1f KB" % (os.path.getsize(os.path.join(OUT, tag + ".json")) / 1024))


def gen_engine(tag, seed, B=6, V=53, H=16, E=16, A=16):
    """Engine-level goldens for BUTDDetection_Eng (D is fixed at 2048 by BUTD_Model.py:449):
    E1 training_epoch (2 steps), E2 SCST_training_epoch (2 steps), E3 eval_captions_json_generation
    (greedy + beam 3), R1 get_self_critical_reward, E4 modify_visual_inputs.
    Features / masks / uniforms are stored as seeds (tests/golden/synth.py regenerates them); updated
    parameters are pinned by pin_tensor() (full copy if small, else 256-point sample + two moments)."""
    from ModelEngines.BUTD_Engine import BUTDDetection_Eng
    from Utils import init_optimizer, LabelSmoothingLoss, RewardCriterion, get_self_critical_reward
    torch.manual_seed(seed)
    rng = np.random.RandomState(seed)
    R, D = 36, 2048
    vocab = make_vocab(V)
    words = [vocab.ix2word[i] for i in range(V)]
    cwd = os.getcwd()
    td = tempfile.mkdtemp()
    os.makedirs(os.path.join(td, "cider", "data"))
    cfg = os.path.join(td, "m.json")
    json.dump({"model_type": "BUTDDetection", "enc_img_size": 7, "atten_dim": A, "embed_dim": E,
               "hidden_dim": H, "optimizer": "Adam", "lr": 4e-4, "scst_lr": 2e-5}, open(cfg, "w"))
    train = synth_refs(rng, words, 300)
    df = build_df(train)
    with open(os.path.join(td, "cider", "data", "SYN-train.p"), "wb") as f:
        pickle.dump(df, f, protocol=2)
    os.chdir(td)
    seed_ctr = [seed * 1000]

    def next_seed():
        seed_ctr[0] += 1
        return seed_ctr[0]

    def pin_sd(out, prefix, sd):
        for k, v in sd_to_np(sd).items():
            for kk, vv in pin_tensor(v).items():
                out["%s%s/%s" % (prefix, k, kk)] = vv

    try:
        eng = BUTDDetection_Eng(model_settings_json=cfg, dataset_name="SYN", caption_vocab=vocab,
                                data_dir=td + "/", use_bu="fixed", device="cpu")
        sharpen(eng.model.decoder, V)
        with torch.no_grad():
            eng.model.decoder.predict.bias[2] = 1.0
        out = {"dims": np.array([B, R, D, H, E, A, V], dtype=np.int64)}
        out.update({"sd0." + k: v for k, v in sd_to_np(eng.model.state_dict()).items()})
        fx = {"df": df_to_json(df), "vocab": words}

        def batch_feats():
            s_ = next_seed()
            f_ = feats_from_seed(s_, B, R, D)
            return s_, tuple({"bu_feat": f_[i], "bu_bbox": np.zeros((R, 4), np.float32)} for i in range(B))

        dummy_img = torch.zeros(B, 3, 2, 2)

        # ---- E3 greedy + JSON, E4
        fseed, supp = batch_feats()
        vi = eng.modify_visual_inputs(dummy_img, supp)
        assert vi["bu_masks"] is None
        ids_eval = tuple(int(x) for x in rng.randint(1000, 9999, size=B))
        res = eng.eval_captions_json_generation([(ids_eval, dummy_img, supp)], eval_beam_size=-1,
                                                tqdm_visible=False)
        with torch.no_grad():
            eng.model.eval()
            g_ids = eng.model.sampler(vi, max_len=20)
        out.update(eval_feats_seed=np.int64(fseed), eval_greedy_ids=g_ids.numpy(),
                   eval_img_ids=np.array(ids_eval))
        fx["eval_greedy_json"] = res
        # beam 3 (dataloader batch 1, Utils.py:72-73)
        res_b = []
        with legacy_int_div():
            for i in range(B):
                one = (supp[i],)
                r_ = eng.eval_captions_json_generation([((ids_eval[i],), dummy_img[:1], one)],
                                                       eval_beam_size=3, tqdm_visible=False)
                res_b += r_
                with torch.no_grad():
                    s_ = eng.model.beam_search_sampler(eng.modify_visual_inputs(dummy_img[:1], one), 3)
                out["eval_beam3_seq_%d" % i] = s_.numpy().astype(np.float32).ravel()
        fx["eval_beam3_json"] = res_b

        # ---- E1 training_epoch: two XE steps (fresh Adam, lr 4e-4, clamp 0.1: Engine.py:136,187)
        opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 4e-4, "cnn_ft_lr": 0.0}), 4e-4)
        crit = LabelSmoothingLoss(smoothing=0.1)
        for step in range(2):
            fseed, supp = batch_feats()
            lens = sorted(rng.randint(6, 13, size=B).tolist(), reverse=True)
            L = max(lens)
            caps = np.zeros((B, L), dtype=np.int64)
            for b, l in enumerate(lens):
                caps[b, 0] = 1
                caps[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
                caps[b, l - 1] = 2
            T = L - 1
            mseed = next_seed()
            em, am, om, _ = masks_from_seed(mseed, T, B, R, E, A, H)
            INJ.set([em, am, om], None)
            losses = []
            orig_crit_fwd = crit.forward

            def rec_fwd(i, t, _f=orig_crit_fwd):
                l_ = _f(i, t)
                losses.append(float(l_.item()))
                return l_
            crit.forward = rec_fwd
            with injected():
                eng.training_epoch([(tuple(range(B)), dummy_img, torch.from_numpy(caps), list(lens), supp)],
                                   opt, crit, tqdm_visible=False)
            crit.forward = orig_crit_fwd
            p = "xe%d_" % step
            out.update({p + "feats_seed": np.int64(fseed), p + "captions": caps,
                        p + "lengths": np.array(lens), p + "mask_seed": np.int64(mseed),
                        p + "loss": np.float32(losses[0])})
            pin_sd(out, p + "sd.", eng.model.state_dict())

        # ---- E2 SCST_training_epoch: two steps (Adam lr 2e-5 persists, clamp 0.25: Engine.py:215,271)
        opt = init_optimizer("Adam", eng.model.get_param_groups({"lr": 2e-5, "cnn_ft_lr": 0.0}), 2e-5)
        rcrit = RewardCriterion()
        T = 20
        for step in range(2):
            fseed, supp = batch_feats()
            img_ids = tuple(range(step * B, step * B + B))
            gts = synth_refs(rng, words, B)
            gts = {img_ids[i]: gts[i] for i in range(B)}
            mseed = next_seed()
            em, am, om, u = masks_from_seed(mseed, T, B, R, E, A, H)
            INJ.set([em, am, om], u)
            rec = {}
            orig_fwd = rcrit.forward

            def rec_fwd2(lp, seq, rew, _f=orig_fwd):
                l_ = _f(lp, seq, rew)
                rec.update(lp=lp.detach().numpy().copy(), seq=seq.numpy().copy(),
                           rew=rew.numpy().copy(), loss=float(l_.item()))
                return l_
            rcrit.forward = rec_fwd2
            # greedy baseline of the *current* weights, recorded separately for the fixture
            vi = eng.modify_visual_inputs(dummy_img, supp)
            with torch.no_grad():
                eng.model.eval()
                g_ids = eng.model.sampler(vi, max_len=20).numpy().copy()
            with injected():
                eng.SCST_training_epoch([(img_ids, dummy_img, gts, supp)], opt, rcrit, tqdm_visible=False)
            rcrit.forward = orig_fwd
            p = "rl%d_" % step
            out.update({p + "feats_seed": np.int64(fseed), p + "mask_seed": np.int64(mseed),
                        p + "greedy_ids": g_ids, p + "seq": rec["seq"], p + "logprobs": rec["lp"],
                        p + "reward": rec["rew"], p + "loss": np.float32(rec["loss"]),
                        p + "img_ids": np.array(img_ids)})
            fx[p + "gts"] = {str(k): v for k, v in gts.items()}
            pin_sd(out, p + "sd.", eng.model.state_dict())

        # ---- R1 get_self_critical_reward stand-alone incl. all-zero sampled row and mid-sentence <pad>
        gen = rng.randint(3, V, size=(B, 20)).astype(np.int64)
        gen[0, :] = 0                       # all-zero row -> "<pad>"
        gen[1, 5:] = 0                      # finished after 5 words
        gen[2, 3] = 0                       # <pad> sampled mid-sentence, still unfinished
        gen[3, 10:] = 0
        gre = rng.randint(3, V, size=(B, 20)).astype(np.int64)
        gre[0, 0] = 2                       # greedy emits <end> first -> empty sentence
        gre[1, 7] = 2
        gre[2, 19] = 2
        gts = synth_refs(rng, words, B)
        ids = tuple(range(B))
        rew = get_self_critical_reward(torch.from_numpy(gen), torch.from_numpy(gre), gts, ids, vocab, "SYN")
        out.update(r1_gen=gen, r1_greedy=gre, r1_reward=rew.numpy())
        fx["r1_gts"] = {str(k): v for k, v in gts.items()}
    finally:
        os.chdir(cwd)
    save(tag, **out)
    with open(os.path.join(OUT, tag + ".json"), "w") as f:
        json.dump(fx, f)
    print("wrote", tag + ".json", "%.1f KB" % (os.path.getsize(os.path.join(OUT, tag + ".json")) / 1024))


if __name__ == "__main__":
    bootstrap()
    torch.set_num_threads(4)
    which = sys.argv[1:] or ["butd", "butd2", "nic", "aoa", "cider", "corpus", "engine"]
    if "butd" in which:
        gen_butd_decoder("butd_dec_tiny", B=5, R=36, D=64, H=32, E=32, A=32, V=53, seed=11)
        gen_butd_decoder("butd_dec_odd", B=3, R=36, D=96, H=48, E=16, A=64, V=70, seed=12)
    if "butd2" in which:
        # BUTDSpatial_Captioner's decoder (BUTD_Model.py:321-440 builds the same DecoderRNN over the 7 x 7 = 49 grid cells of
        # EncoderCNN, BASELINE config 2) and a batch of long captions (more XE time steps than any decode runs)
        gen_butd_decoder("butd_dec_spatial", B=4, R=49, D=64, H=32, E=32, A=32, V=53, seed=13)
        gen_butd_decoder("butd_dec_long", B=3, R=36, D=32, H=16, E=16, A=16, V=53, seed=14, cap_len=(28, 43))
    if "nic" in which:
        gen_nic_decoder("nic_dec_tiny", B=5, H=32, E=32, V=53, seed=31)
        gen_nic_decoder("nic_dec_odd", B=3, H=48, E=16, V=70, seed=32)
    if "aoa" in which:
        gen_aoa("aoa_tiny", B=4, Hd=32, E=16, V=53, seed=41)
        gen_aoa("aoa_adaptive", B=4, Hd=32, E=16, V=53, seed=43, counts=[70, 23, 66, 41])
    if "cider" in which:
        gen_cider("ciderd_cases", seed=5)
    if "corpus" in which:
        gen_corpus_cider("corpus_cider_cases", seed=7)
    if "engine" in which:
        gen_engine("butd_engine_tiny", seed=21)
