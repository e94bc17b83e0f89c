#!/usr/bin/env python3
"""Full-width golden vectors from the reference itself ("one full-dim single-step vector per model"; VERDICT r03
item 6).  Runs ONLY in the build container (CPU), like make_goldens.py, whose bootstrap it reuses: the reference's modules are
imported unmodified from /root/reference.

The other fixtures pin the oracle at H <= 48, V <= 70; every parity test at the measured width (H = E = A = 1024, V = 10102)
went through the oracle.  These two fixtures remove that indirection for the decoder step:

  * butd_fullwidth_step.npz -- Models/BUTD_Model.py DecoderRNN at the benchmark size, ONE decoder step (:172-182, evaluation mode)
    on 3 rows from a random state: new states, context, attention weights, logits; plus a second step chained on the first one's
    outputs with the argmax tokens (what a greedy decode does), its logits' top-2 margin per row and the argmax ids.
  * aoa_fullwidth_step.npz  -- Models/AoA_Model.py AoADetection_Captioner at the benchmark size (6-layer refiner 36 x 2048 -> 1024,
    8 heads, decoder LSTM + AoA block, V = 10102): probes of the refined features and the packed logits of a teacher-forced
    evaluation-mode forward pass over two decoder steps on 2 images.

Only seeds and OUTPUTS are stored (a few hundred KB): the weights are regenerated from a seed on both sides -- BUTD by
simpleimagecaptionzoo_amd.synth.random_butd_params (a torch.Generator stream), AoA by constructing the product's own
AoADetection_Captioner under torch.manual_seed on the CPU (torch's default initialisers; same torch build on the GPU box) -- and
loaded into the reference modules with load_state_dict(strict=True) here.  Inputs come from numpy RandomState seeds.

Usage:  python tests/golden/make_fullwidth_goldens.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import make_goldens as mg  # noqa: E402
from synth import feats_from_seed, probe_indices  # noqa: E402

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102


def butd_inputs(seed, B):
    """Seeded inputs of the BUTD step (shared with tests/test_gpu_round4.py): features, a random state, input tokens."""
    rs = np.random.RandomState(seed)
    feats = feats_from_seed(seed + 1, B, R, D)
    st = [(rs.randn(B, H) * 0.5).astype(np.float32) for _ in range(4)]
    it = rs.randint(4, V, size=(B,)).astype(np.int64)
    return feats, st, it


def gen_butd(tag="butd_fullwidth_step", seed=9001, B=3):
    from Models.BUTD_Model import DecoderRNN
    from simpleimagecaptionzoo_amd.synth import random_butd_params
    params = random_butd_params(R, D, H, E, A, V, "cpu", seed=seed)
    dec = DecoderRNN(atten_dim=A, embed_dim=E, hidden_dim=H, vocab_size=V, enc_dim=D)
    missing = dec.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    dec.eval()
    feats_np, st, it_np = butd_inputs(seed, B)
    feats = torch.from_numpy(feats_np)
    h1, c1, h2, c2 = [torch.from_numpy(x) for x in st]
    it = torch.from_numpy(it_np)
    out = {"dims": np.array([B, R, D, H, E, A, V], dtype=np.int64), "seed": np.int64(seed)}

    def step(it, h1, c1, h2, c2):        # BUTD_Model.py:172-182
        emb = dec.embed(it)
        nh1, nc1 = dec.TD_atten(torch.cat([h2, feats.mean(1), emb], 1), (h1, c1))
        ctx, alpha = dec.atten(feats, nh1)
        nh2, nc2 = dec.language_model(torch.cat([ctx, nh1], 1), (h2, c2))
        return nh1, nc1, nh2, nc2, ctx, alpha, dec.predict(dec.dropout(nh2))

    with torch.no_grad():
        nh1, nc1, nh2, nc2, ctx, alpha, logits = step(it, h1, c1, h2, c2)
        out.update(s1_nh1=nh1.numpy(), s1_nc1=nc1.numpy(), s1_nh2=nh2.numpy(), s1_nc2=nc2.numpy(), s1_ctx=ctx.numpy(),
                   s1_alpha=alpha.numpy(), s1_logits=logits.numpy())
        top = torch.topk(logits, 2, dim=1)
        tok = top.indices[:, 0]
        out.update(s1_argmax=tok.numpy(), s1_margin=(top.values[:, 0] - top.values[:, 1]).numpy())
        _, _, _, _, ctx2, alpha2, logits2 = step(tok, nh1, nc1, nh2, nc2)
        top2 = torch.topk(logits2, 2, dim=1)
        out.update(s2_alpha=alpha2.numpy(), s2_argmax=top2.indices[:, 0].numpy(), s2_margin=(top2.values[:, 0] - top2.values[:, 1]).numpy(),
                   s2_logits_probe=logits2.numpy().reshape(-1)[probe_indices(B * V, 2048)], s2_ctx=ctx2.numpy())
    mg.save(tag, **out)


def aoa_captioner(seed):
    """The product's AoADetection_Captioner on the CPU under a torch seed: its parameters are the weights of this fixture."""
    from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner as Ours
    torch.manual_seed(seed)
    m = Ours(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E, device="cpu")
    with torch.no_grad():       # non-trivial LayerNorm gains / biases (the defaults, ones and zeros, would hide a swapped pair)
        g = torch.Generator(device="cpu")
        g.manual_seed(seed + 17)
        for name, prm in m.named_parameters():
            if name.endswith("norm.gain"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.2)
            if name.endswith("norm.bias"):
                prm.add_(torch.randn(prm.shape, generator=g) * 0.1)
    return m


def aoa_inputs(seed, B):
    rs = np.random.RandomState(seed)
    feats = feats_from_seed(seed + 1, B, R, D)
    caps = np.zeros((B, 3), dtype=np.int64)
    caps[:, 0] = 1
    caps[:, 1:] = rs.randint(4, V, size=(B, 2))
    return feats, caps, [2] * B


def gen_aoa(tag="aoa_fullwidth_step", seed=9101, B=2):
    from Models.AoA_Model import AoADetection_Captioner, pack_wrapper
    ours = aoa_captioner(seed)
    m = AoADetection_Captioner(vocab_size=V, num_heads=8, hidden_dim=H, embed_dim=E)
    missing = m.load_state_dict(ours.state_dict(), strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    m.eval()
    feats_np, caps, lengths = aoa_inputs(seed, B)
    feats = torch.from_numpy(feats_np)
    vi = {"bu_feats": feats, "bu_bboxes": None, "bu_masks": None}
    out = {"dims": np.array([B, R, D, H, E, V, 8], dtype=np.int64), "seed": np.int64(seed)}
    with torch.no_grad():
        refined = m.aoa_refine(pack_wrapper(m.img_feats_porjection, feats, None), None)
        packed = m(vi, torch.from_numpy(caps), lengths)[0]
    r = refined.numpy().reshape(-1)
    out.update(refined_probe=r[probe_indices(r.size, 4096)], refined_sum=np.float64(r.astype(np.float64).sum()),
               refined_sumsq=np.float64((r.astype(np.float64) ** 2).sum()), packed_logits=packed.numpy())
    top = torch.topk(packed, 2, dim=1)
    out.update(argmax=top.indices[:, 0].numpy(), margin=(top.values[:, 0] - top.values[:, 1]).numpy())
    mg.save(tag, **out)


if __name__ == "__main__":
    mg.bootstrap()
    gen_butd()
    gen_aoa()
    for f in ("butd_fullwidth_step.npz", "aoa_fullwidth_step.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KB")
