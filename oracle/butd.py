"""Oracle: BUTD top-down attention decoder (torch CPU fp32, functional, autograd-capable).

Parameters are addressed by the reference's state_dict key names without the leading "decoder."
(e.g. "atten.enc_att.weight_g", "TD_atten.weight_ih", "predict.bias"; Models/BUTD_Model.py:64-90).
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch

PAD, STA, END, UNK = 0, 1, 2, 3      # PreProcess/Build_caption_vocab.py:37-40


def strip_prefix(sd, prefix="decoder."):
    return {(k[len(prefix):] if k.startswith(prefix) else k): v for k, v in sd.items()}


def to_params(sd_np, requires_grad=False):
    p = {}
    for k, v in strip_prefix(sd_np).items():
        t = torch.tensor(np.asarray(v), dtype=torch.float32)
        t.requires_grad_(requires_grad)
        p[k] = t
    return p


def wn_weight(p, name):
    """Old-style weight_norm, dim=0: w = g * v / ||v||_row  (BUTD_Model.py:43-45,84)."""
    v, g = p[name + ".weight_v"], p[name + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1))))


def lstm_cell(x, h, c, p, name):
    """nn.LSTMCell, gate order i,f,g,o (BUTD_Model.py:82-83,137-145)."""
    gates = x @ p[name + ".weight_ih"].t() + p[name + ".bias_ih"] + h @ p[name + ".weight_hh"].t() + p[name + ".bias_hh"]
    i, f, g, o = gates.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def drop(x, mask, p=0.5):
    """Dropout with an explicit keep-mask (None = eval mode)."""
    if mask is None:
        return x
    return x * mask.to(x.dtype).reshape(x.shape) * (1.0 / (1.0 - p))


def hoist(feats, p):
    """What does not change from one decoder step to the next, computed ONCE: the four weight-normed matrices and enc_att(feats)
    (BUTD_Model.py:57 recomputes the latter in every step, as step() / soft_attention() do by default).  Passing the result as
    `pre` leaves every forward value bitwise as it was (the same operations on the same operands); a gradient reaches the parameters
    through one graph node instead of one per step (the same sum in another association).  For the full-width tests, whose CPU
    passes are otherwise 3/4 enc_att products; the golden checks (tests/test_oracle_golden.py) run the per-step form, and
    tests/test_oracle_golden.py::test_hoisted_oracle_equals_the_per_step_form holds the two together."""
    w = {n: wn_weight(p, n) for n in ("atten.enc_att", "atten.dec_att", "atten.affine", "predict")}
    return {"w": w, "enc_ctx": None if feats is None else feats @ w["atten.enc_att"].t() + p["atten.enc_att.bias"]}


def soft_attention(feats, h1, p, att_mask=None, pre=None):
    """SoftAttention.forward, BUTD_Model.py:49-62.  pre: see hoist()."""
    W = (lambda n: pre["w"][n]) if pre else (lambda n: wn_weight(p, n))
    enc_ctx = pre["enc_ctx"] if pre and pre["enc_ctx"] is not None else feats @ W("atten.enc_att").t() + p["atten.enc_att.bias"]
    dec_ctx = h1 @ W("atten.dec_att").t() + p["atten.dec_att.bias"]
    z = drop(torch.relu(enc_ctx + dec_ctx.unsqueeze(1)), att_mask)
    score = (z @ W("atten.affine").t()).squeeze(2) + p["atten.affine.bias"]
    alpha = torch.softmax(score, dim=1)
    return (feats * alpha.unsqueeze(2)).sum(1), alpha


def embed(it, p, emb_mask=None):
    """Embedding -> ReLU -> Dropout, BUTD_Model.py:77-81."""
    return drop(torch.relu(p["embed.0.weight"][it]), emb_mask)


def step(feats, mean, it, state, p, masks=(None, None, None), pre=None):
    """One decoder step, BUTD_Model.py:172-182.  state = (h1, c1, h2, c2).  pre: see hoist()."""
    h1, c1, h2, c2 = state
    e = embed(it, p, masks[0])
    h1, c1 = lstm_cell(torch.cat([h2, mean, e], 1), h1, c1, p, "TD_atten")
    ctx, alpha = soft_attention(feats, h1, p, masks[1], pre)
    h2, c2 = lstm_cell(torch.cat([ctx, h1], 1), h2, c2, p, "language_model")
    logits = drop(h2, masks[2]) @ (pre["w"]["predict"] if pre else wn_weight(p, "predict")).t() + p["predict.bias"]
    return logits, alpha, (h1, c1, h2, c2)


def zero_state(B, H):
    return tuple(torch.zeros(B, H) for _ in range(4))


def greedy(feats, p, max_len=20, hoisted=False):
    """DecoderRNN.sample, BUTD_Model.py:153-189 -> ids (B,max_len) int64, alphas, logits.  hoisted: see hoist()."""
    B = feats.shape[0]
    H = p["TD_atten.weight_hh"].shape[1]
    mean, st = feats.mean(1), zero_state(B, H)
    it = torch.full((B,), STA, dtype=torch.long)
    ids, als, lgs = [], [], []
    pre = hoist(feats, p) if hoisted else None
    for _ in range(max_len):
        logits, alpha, st = step(feats, mean, it, st, p, pre=pre)
        it = logits.max(1)[1]
        ids.append(it)
        als.append(alpha)
        lgs.append(logits)
    return torch.stack(ids, 1), torch.stack(als, 1), torch.stack(lgs, 1)


def inverse_cdf_draw(prob, u):
    """The sampler contract shared with the HIP path: smallest i with cumsum(prob)[i] > u*sum(prob),
    evaluated in float64 (stands in for torch.multinomial at BUTD_Model.py:223; see
    tests/golden/make_goldens.py Injector.multinomial)."""
    c = torch.cumsum(prob.double(), dim=1)
    tgt = (torch.as_tensor(u, dtype=torch.float64) * c[:, -1]).unsqueeze(1)
    return torch.searchsorted(c, tgt, right=True).clamp_(max=prob.shape[1] - 1).squeeze(1)


def sample_rl(feats, p, uniforms, emb_masks, att_masks, out_masks, max_len=20, early_exit=True, trace=None, hoisted=False):
    """DecoderRNN.sample_rl, BUTD_Model.py:191-234 with explicit uniforms/masks.
    Returns seq (B,T) int64 (0 at and after a sampled <end>), logprobs (B,T) (autograd-capable), logits.
    trace: optional dict; receives "h1" = the attention LSTM's hidden state of every step (detached), for tests that look at
    the attention pre-activations behind a gradient.  hoisted: see hoist()."""
    B = feats.shape[0]
    H = p["TD_atten.weight_hh"].shape[1]
    mean, st = feats.mean(1), zero_state(B, H)
    it = torch.full((B,), STA, dtype=torch.long)
    seq = torch.zeros(B, max_len, dtype=torch.long)
    lps = [torch.zeros(B) for _ in range(max_len)]
    lgs = []
    unfinished = torch.ones(B, dtype=torch.bool)
    pre = hoist(feats, p) if hoisted else None
    for t in range(max_len):
        m = (None if emb_masks is None else torch.as_tensor(emb_masks[t]),
             None if att_masks is None else torch.as_tensor(att_masks[t]),
             None if out_masks is None else torch.as_tensor(out_masks[t]))
        logits, _, st = step(feats, mean, it, st, p, m, pre)
        if trace is not None:
            trace.setdefault("h1", []).append(st[0].detach())
        logp = torch.log_softmax(logits, dim=1)
        draw = inverse_cdf_draw(torch.exp(logp.detach()), uniforms[t])
        lps[t] = logp.gather(1, draw.unsqueeze(1)).squeeze(1)
        unfinished = unfinished & (draw != END)
        it = draw * unfinished.long()
        seq[:, t] = it
        lgs.append(logits)
        if early_exit and not bool(unfinished.any()):
            break
    return seq, torch.stack(lps, 1), torch.stack(lgs, 1)


def beam_search(feats1, p, k, max_steps=50):
    """DecoderRNN.beam_search_sample, BUTD_Model.py:236-318 (batch 1, shrinking k, no length norm,
    legacy integer division at :277).  Returns float32 (1,L) incl. <sta> and (if finished) <end>."""
    V = p["predict.bias"].shape[0]
    H = p["TD_atten.weight_hh"].shape[1]
    feats = feats1.expand(k, -1, -1)
    mean = feats.mean(1)
    st = zero_state(k, H)
    prev = torch.full((k,), STA, dtype=torch.long)
    seqs = prev.view(k, 1)
    run = torch.zeros(k, 1)
    done, done_scores = [], []
    pre = hoist(None, p)          # the weight-normed matrices once (bitwise the per-step values); enc_att(feats) stays per step: the rows shrink
    for stp in range(1, max_steps + 1):
        logits, _, st = step(feats, mean, prev, st, p, pre=pre)
        sc = run.expand(-1, V) + torch.log_softmax(logits, dim=1)
        top, idx = (sc[0] if stp == 1 else sc.reshape(-1)).topk(k, 0, True, True)
        src, nxt = torch.div(idx, V, rounding_mode="floor"), idx % V
        seqs = torch.cat([seqs[src], nxt.view(-1, 1)], 1)
        keep = [j for j in range(len(nxt)) if int(nxt[j]) != END]
        fin = [j for j in range(len(nxt)) if int(nxt[j]) == END]
        for j in fin:
            done.append(seqs[j].tolist())
            done_scores.append(float(top[j]))
        k -= len(fin)
        if k == 0:
            break
        seqs = seqs[keep]
        sel = src[keep]
        feats, mean = feats[sel], mean[sel]
        st = tuple(s[sel] for s in st)
        run = top[keep].view(-1, 1)
        prev = nxt[keep]
    if done:
        best = done_scores.index(max(done_scores))
        return torch.tensor(done[best], dtype=torch.float32).view(1, -1)
    return seqs[int(run.view(-1).argmax())].view(1, -1).float()


def packed_order(lengths):
    """(b, t) pairs in pack_padded_sequence order (time-major over rows with length > t); lengths sorted desc."""
    return [(b, t) for t in range(max(lengths)) for b in range(len(lengths)) if lengths[b] > t]


def scheduled_tokens(captions, t, bt, prev_logits, ss_prob, ss_gate, ss_draw):
    """Tokens fed at XE time step t (the block shared by BUTD_Model.py:120-132, AoA_Model.py:258-270, NIC_Model.py:77-89):
    the caption's, except that from step 2 on a row whose gate uniform is below ss_prob feeds a draw from
    softmax(previous logits) (inverse_cdf_draw with ss_draw[t]; detached)."""
    it = captions[:bt, t]
    if t >= 2 and ss_prob > 0.0:
        gate = torch.as_tensor(ss_gate[t][:bt], dtype=torch.float32) < ss_prob
        if bool(gate.any()):
            draw = inverse_cdf_draw(torch.softmax(prev_logits.detach()[:bt], dim=1), ss_draw[t][:bt])
            it = torch.where(gate, draw, it)
    return it


def forward_xe(feats, captions, lengths, p, emb_masks=None, att_masks=None, out_masks=None, ss_prob=0.0, ss_gate=None,
               ss_draw=None, tokens_out=None, trace=None):
    """DecoderRNN.forward, BUTD_Model.py:97-151.  Returns packed logits (sum(lengths), V).  trace: as in sample_rl ("h1" per step,
    rows past a step's active count zero-padded to the batch).
    Scheduled sampling (:120-132, ss_prob > 0): from time step 2 on, row b feeds a draw from softmax(previous logits)
    instead of its caption token when ss_gate[t][b] < ss_prob; the draw is inverse_cdf_draw with ss_draw[t][b] (no
    gradient flows through it).  tokens_out, if a list, receives the tokens fed at every step."""
    B = feats.shape[0]
    H = p["TD_atten.weight_hh"].shape[1]
    mean, st = feats.mean(1), zero_state(B, H)
    rows = []
    logits = None
    for t in range(max(lengths)):
        bt = sum(l > t for l in lengths)
        m = (None if emb_masks is None else torch.as_tensor(emb_masks[t][:bt]),
             None if att_masks is None else torch.as_tensor(att_masks[t][:bt]),
             None if out_masks is None else torch.as_tensor(out_masks[t][:bt]))
        it = scheduled_tokens(captions, t, bt, logits, ss_prob, ss_gate, ss_draw)
        if tokens_out is not None:
            tokens_out.append(it.clone())
        logits, _, st = step(feats[:bt], mean[:bt], it, tuple(s[:bt] for s in st), p, m)
        if trace is not None:
            h1 = torch.zeros(B, H, dtype=st[0].dtype)
            h1[:bt] = st[0].detach()
            trace.setdefault("h1", []).append(h1)
        rows.append(logits)
    return torch.cat(rows, 0)


def label_smoothing_loss(logits, target, smoothing=0.1):
    """LabelSmoothingLoss.forward, Utils.py:268-286: sum KL(true || softmax) / N."""
    lp = torch.log_softmax(logits, dim=-1)
    V = lp.shape[1]
    true = torch.full_like(lp, smoothing / (V - 1))
    true.scatter_(1, target.view(-1, 1), 1.0 - smoothing)
    kl = torch.where(true > 0, true * (true.log() - lp), torch.zeros_like(lp))
    return kl.sum() / lp.shape[0]


def reward_criterion(logprobs, seq, reward):
    """RewardCriterion.forward, Utils.py:295-317."""
    mask = (seq > 0).float()
    mask = torch.cat([torch.ones(mask.shape[0], 1), mask[:, :-1]], 1)
    return -(logprobs * reward * mask).sum() / mask.sum()


class Adam:
    """torch.optim.Adam(betas=(0.9,0.999), eps=1e-8, wd=0) restated (Utils.py:219-220) with the value clamp
    of Utils.py:241-250 applied first."""

    def __init__(self, params, lr):
        self.p, self.lr, self.t = params, lr, 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, grads, clip):
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8
        bc1, bc2 = 1 - b1 ** self.t, 1 - b2 ** self.t
        with torch.no_grad():
            for k, w in self.p.items():
                g = grads[k].clamp(-clip, clip)
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
                w.addcdiv_(self.m[k], denom, value=-self.lr / bc1)
