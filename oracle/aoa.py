"""Oracle: AoADetection captioner -- feature projection, 6-layer AoA refiner, AoA decoder (torch CPU fp32, functional).
Parameter names = reference state_dict keys (Models/AoA_Model.py).  TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py."""
import math

import torch

from .butd import END, STA, drop, inverse_cdf_draw, lstm_cell, wn_weight

NH = 8


def layer_norm(x, gain, bias, eps=1e-6):
    """Custom LayerNorm, AoA_Model.py:14-25: UNBIASED std, eps added to the std (not inside the sqrt)."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gain * (x - mean) / (std + eps) + bias


def key_mask(lens, R):
    """Prefix bu_masks [B, R] (1 = valid) from per-image region counts, AoA_Engine.py:37-40; None = all valid."""
    if lens is None:
        return None
    return (torch.arange(R).unsqueeze(0) < torch.as_tensor(lens).view(-1, 1)).float()


def masked_mean(enc, bu_mask):
    """mean_features, AoA_Model.py:250-253."""
    if bu_mask is None:
        return enc.mean(1)
    return (enc * bu_mask.unsqueeze(-1)).sum(1) / bu_mask.unsqueeze(-1).sum(1)


def aoa_block(query, kv, p, pre, att_mask=None, aoa_mask=None, aoa_p=0.3, bu_mask=None, kv_proj=None):
    """AoABlock.forward, AoA_Model.py:90-120: 8-head dot-product attention (keys with bu_mask == 0 filled with -1e9
    before the softmax, :63-64,108-110) + GLU gate.  kv_proj: (linear_K(kv), linear_V(kv)) computed by the caller (hoist_dec)."""
    B, nq, Hd = query.shape
    d = Hd // NH
    lin = lambda x, n: x @ p[pre + n + ".weight"].t() + p[pre + n + ".bias"]
    Q = lin(query, "linear_Q").view(B, -1, NH, d).transpose(1, 2)
    K = (kv_proj[0] if kv_proj else lin(kv, "linear_K")).view(B, -1, NH, d).transpose(1, 2)
    V = (kv_proj[1] if kv_proj else lin(kv, "linear_V")).view(B, -1, NH, d).transpose(1, 2)
    S = Q @ K.transpose(-2, -1) / math.sqrt(d)
    if bu_mask is not None:
        S = S.masked_fill(bu_mask[:, None, None, :] == 0, -1e9)
    P = torch.softmax(S, dim=-1)
    alpha = P.mean(1)
    P = drop(P, att_mask, 0.1)
    x = (P @ V).transpose(1, 2).contiguous().view(B, nq, Hd)
    z = drop(torch.cat([x, query], -1), aoa_mask, aoa_p)
    z = z @ p[pre + "aoa_module.0.weight"].t() + p[pre + "aoa_module.0.bias"]
    return z[..., :Hd] * torch.sigmoid(z[..., Hd:]), alpha


def refine(feats, p, masks=None, lens=None):
    """img_feats_porjection + AoA_Refine_Core, AoA_Model.py:661-665,140-162.  masks: dict proj / ref_att / ref_aoa /
    ref_sc (None = eval mode).  lens: valid regions per image ('adaptive' features); the projection then runs on the
    valid rows only and the padding rows are zero (pack_wrapper, :650-653)."""
    g = (lambda k, l=None: None) if masks is None else (lambda k, l=None: torch.as_tensor(masks[k] if l is None else masks[k][l]))
    bu = key_mask(lens, feats.shape[1])
    x = drop(torch.relu(feats @ p["img_feats_porjection.0.weight"].t() + p["img_feats_porjection.0.bias"]), g("proj"), 0.5)
    if bu is not None:
        x = x * bu.unsqueeze(-1)
    for l in range(6):
        pre = "aoa_refine.aoa_layers.%d." % l
        n = layer_norm(x, p[pre + "sublayer.norm.gain"], p[pre + "sublayer.norm.bias"])
        y, _ = aoa_block(n, n, p, pre + "aoa_block.", g("ref_att", l), g("ref_aoa", l), 0.3, bu)
        x = x + drop(y, g("ref_sc", l), 0.1)
    return layer_norm(x, p["aoa_refine.norm.gain"], p["aoa_refine.norm.bias"])


def hoist_dec(enc, p):
    """What a decoder step recomputes although it does not change (AoA_Model.py:104-106 projects the refined regions onto keys and
    values in every step; weight_norm rebuilds `predict` in every step), computed once -- as oracle.butd.hoist: forward values bitwise
    those of the per-step form, gradients through one node instead of one per step.  For the full-width tests only."""
    pre = "decoder.aoa_block."
    lin = lambda n: enc @ p[pre + n + ".weight"].t() + p[pre + n + ".bias"]
    return {"kv": (lin("linear_K"), lin("linear_V")), "w_pred": wn_weight(p, "decoder.predict")}


def dec_step(it, state, enc, meanf, p, masks=(None, None, None, None), bu_mask=None, pre=None):
    """One AoA_Decoder step, AoA_Model.py:319-336.  state = (h, m, ctx); masks = (emb, ctx, att, out).  pre: see hoist_dec()."""
    h, m, ctx = state
    emb = drop(torch.relu(p["decoder.embed.0.weight"][it]), masks[0], 0.5)
    u = meanf + drop(ctx, masks[1], 0.5)
    h, m = lstm_cell(torch.cat([emb, u], 1), h, m, p, "decoder.lstm")
    q = layer_norm(h, p["decoder.h_norm.gain"], p["decoder.h_norm.bias"]).unsqueeze(1)
    ctx, alpha = aoa_block(q, enc, p, "decoder.aoa_block.", masks[2], None, bu_mask=bu_mask, kv_proj=pre["kv"] if pre else None)
    ctx = ctx.squeeze(1)
    logits = drop(ctx, masks[3], 0.5) @ (pre["w_pred"] if pre else wn_weight(p, "decoder.predict")).t() + p["decoder.predict.bias"]
    return logits, alpha.squeeze(1), (h, m, ctx)


def _zero(B, Hd):
    return tuple(torch.zeros(B, Hd) for _ in range(3))


def _step_masks(masks, t, bt=None):
    if masks is None:
        return (None, None, None, None)
    cut = (lambda x: x) if bt is None else (lambda x: x[:bt])
    return tuple(torch.as_tensor(cut(masks[k][t])) for k in ("emb", "ctx", "att", "out"))


def greedy(feats, p, max_len=20, lens=None, hoisted=False):
    enc = refine(feats, p, lens=lens)
    B, R, Hd = enc.shape
    bu = key_mask(lens, R)
    meanf, st = masked_mean(enc, bu), _zero(B, Hd)
    it = torch.full((B,), STA, dtype=torch.long)
    ids, lgs = [], []
    pre = hoist_dec(enc, p) if hoisted else None
    for _ in range(max_len):
        logits, _, st = dec_step(it, st, enc, meanf, p, bu_mask=bu, pre=pre)
        it = logits.max(1)[1]
        ids.append(it)
        lgs.append(logits)
    return torch.stack(ids, 1), torch.stack(lgs, 1)


def sample_rl(feats, p, uniforms, masks, max_len=20, early_exit=True, lens=None, hoisted=False):
    enc = refine(feats, p, masks, lens)
    B, R, Hd = enc.shape
    bu = key_mask(lens, R)
    meanf, st = masked_mean(enc, bu), _zero(B, Hd)
    it = torch.full((B,), STA, dtype=torch.long)
    seq = torch.zeros(B, max_len, dtype=torch.long)
    lps = [torch.zeros(B) for _ in range(max_len)]
    unfinished = torch.ones(B, dtype=torch.bool)
    pre = hoist_dec(enc, p) if hoisted else None
    for t in range(max_len):
        logits, _, st = dec_step(it, st, enc, meanf, p, _step_masks(masks, t), bu, pre)
        logp = torch.log_softmax(logits, dim=1)
        draw = inverse_cdf_draw(torch.exp(logp.detach()), uniforms[t])
        lps[t] = logp.gather(1, draw.unsqueeze(1)).squeeze(1)
        unfinished = unfinished & (draw != END)
        it = draw * unfinished.long()
        seq[:, t] = it
        if early_exit and not bool(unfinished.any()):
            break
    return seq, torch.stack(lps, 1)


def forward_xe(feats, captions, lengths, p, masks=None, lens=None, ss_prob=0.0, ss_gate=None, ss_draw=None, tokens_out=None):
    """AoA_Decoder.forward behind the Captioner, AoA_Model.py:229-293.  ss_prob > 0: scheduled sampling (:258-270) through
    oracle.butd.scheduled_tokens."""
    from .butd import scheduled_tokens
    enc = refine(feats, p, masks, lens)
    B, R, Hd = enc.shape
    bu = key_mask(lens, R)
    meanf, st = masked_mean(enc, bu), _zero(B, Hd)
    rows = []
    logits = None
    for t in range(max(lengths)):
        bt = sum(l > t for l in lengths)
        it = scheduled_tokens(captions, t, bt, logits, ss_prob, ss_gate, ss_draw)
        if tokens_out is not None:
            tokens_out.append(it.clone())
        logits, _, st = dec_step(it, tuple(s[:bt] for s in st), enc[:bt], meanf[:bt], p, _step_masks(masks, t, bt),
                                 None if bu is None else bu[:bt])
        rows.append(logits)
    return torch.cat(rows, 0)


def beam_search(feats1, p, k, max_steps=50, lens=None):
    """AoA_Decoder.beam_search_sample, AoA_Model.py:403-502 (state h, m, ctx re-indexed by the source beam; the one
    image's bu_mask broadcasts over the beams)."""
    V = p["decoder.predict.bias"].shape[0]
    enc1 = refine(feats1, p, lens=lens)
    bu = key_mask(lens, enc1.shape[1])
    enc = enc1.expand(k, -1, -1)
    meanf = masked_mean(enc1, bu).expand(k, -1)
    st = _zero(k, enc.shape[2])
    prev = torch.full((k,), STA, dtype=torch.long)
    seqs = prev.view(k, 1)
    run = torch.zeros(k, 1)
    done, done_scores = [], []
    for stp in range(1, max_steps + 1):
        logits, _, st = dec_step(prev, st, enc, meanf, p, bu_mask=bu)
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
        enc, meanf = enc[sel], meanf[sel]
        st = tuple(s[sel] for s in st)
        run = top[keep].view(-1, 1)
        prev = nxt[keep]
    if done:
        return torch.tensor(done[done_scores.index(max(done_scores))], dtype=torch.float32).view(1, -1)
    return seqs[int(run.view(-1).argmax())].view(1, -1).float()
