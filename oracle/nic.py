"""Oracle: NIC single-LSTM decoder (torch CPU fp32, functional, autograd-capable).
Parameters by the reference's state_dict names ("embed.weight", "lstm.weight_ih", ..., "predict.weight_v";
Models/NIC_Model.py:39-50).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import torch

from .butd import END, STA, drop, inverse_cdf_draw, lstm_cell, wn_weight


def init_state(feats, p):
    """init_hidden_state, NIC_Model.py:52-56: one LSTM step on the image embedding from a zero state."""
    z = torch.zeros(feats.shape[0], p["lstm.weight_hh"].shape[1])
    return lstm_cell(feats, z, z, p, "lstm")


def step(it, h, c, p, out_mask=None):
    """NIC_Model.py:112-114: plain embedding (no ReLU / dropout) -> LSTMCell -> predict(dropout(h))."""
    h, c = lstm_cell(p["embed.weight"][it], h, c, p, "lstm")
    logits = drop(h, out_mask) @ wn_weight(p, "predict").t() + p["predict.bias"]
    return logits, h, c


def greedy(feats, p, max_len=20):
    """DecoderRNN.sample, NIC_Model.py:100-119."""
    h, c = init_state(feats, p)
    it = torch.full((feats.shape[0],), STA, dtype=torch.long)
    ids, lgs = [], []
    for _ in range(max_len):
        logits, h, c = step(it, h, c, p)
        it = logits.max(1)[1]
        ids.append(it)
        lgs.append(logits)
    return torch.stack(ids, 1), torch.stack(lgs, 1)


def sample_rl(feats, p, uniforms, out_masks, max_len=20, early_exit=True):
    """DecoderRNN.sample_rl, NIC_Model.py:121-151 with explicit uniforms / masks."""
    B = feats.shape[0]
    h, c = init_state(feats, p)
    it = torch.full((B,), STA, dtype=torch.long)
    seq = torch.zeros(B, max_len, dtype=torch.long)
    lps = [torch.zeros(B) for _ in range(max_len)]
    unfinished = torch.ones(B, dtype=torch.bool)
    for t in range(max_len):
        logits, h, c = step(it, h, c, p, None if out_masks is None else torch.as_tensor(out_masks[t]))
        logp = torch.log_softmax(logits, dim=1)
        draw = inverse_cdf_draw(torch.exp(logp.detach()), uniforms[t])
        lps[t] = logp.gather(1, draw.unsqueeze(1)).squeeze(1)
        unfinished = unfinished & (draw != END)
        it = draw * unfinished.long()
        seq[:, t] = it
        if early_exit and not bool(unfinished.any()):
            break
    return seq, torch.stack(lps, 1)


def beam_search(feats1, p, k, max_steps=50):
    """DecoderRNN.beam_search_sample, NIC_Model.py:153-212."""
    V = p["predict.bias"].shape[0]
    h, c = init_state(feats1.expand(k, -1), p)
    prev = torch.full((k,), STA, dtype=torch.long)
    seqs = prev.view(k, 1)
    run = torch.zeros(k, 1)
    done, done_scores = [], []
    for stp in range(1, max_steps + 1):
        logits, h, c = step(prev, h, c, p)
        sc = run.expand(-1, V) + torch.log_softmax(logits, dim=1)
        top, idx = (sc[0] if stp == 1 else sc.reshape(-1)).topk(k, 0, True, True)
        src, nxt = torch.div(idx, V, rounding_mode="floor"), idx % V
        seqs = torch.cat([seqs[src], nxt.view(-1, 1)], 1)
        keep = [j for j in range(len(nxt)) if int(nxt[j]) != END]
        for j in range(len(nxt)):
            if int(nxt[j]) == END:
                done.append(seqs[j].tolist())
                done_scores.append(float(top[j]))
        k -= len(nxt) - len(keep)
        if k == 0:
            break
        seqs = seqs[keep]
        sel = src[keep]
        h, c = h[sel], c[sel]
        run = top[keep].view(-1, 1)
        prev = nxt[keep]
    if done:
        return torch.tensor(done[done_scores.index(max(done_scores))], dtype=torch.float32).view(1, -1)
    return seqs[int(run.view(-1).argmax())].view(1, -1).float()


def forward_xe(feats, captions, lengths, p, out_masks=None, ss_prob=0.0, ss_gate=None, ss_draw=None, tokens_out=None):
    """DecoderRNN.forward, NIC_Model.py:58-98 -> packed logits (sum(lengths), V).  ss_prob > 0: scheduled sampling
    (:77-89) through oracle.butd.scheduled_tokens."""
    from .butd import scheduled_tokens
    h, c = init_state(feats, p)
    rows = []
    logits = None
    for t in range(max(lengths)):
        bt = sum(l > t for l in lengths)
        m = None if out_masks is None else torch.as_tensor(out_masks[t][:bt])
        it = scheduled_tokens(captions, t, bt, logits, ss_prob, ss_gate, ss_draw)
        if tokens_out is not None:
            tokens_out.append(it.clone())
        logits, h, c = step(it, h[:bt], c[:bt], p, m)
        rows.append(logits)
    return torch.cat(rows, 0)
