"""Synthetic, seeded stand-ins for what the benchmark cannot download: reference-style random-init weights
(Models/BUTD_Model.py:75-90 + torch defaults), 36x2048 bottom-up features, a <pad>/<sta>/<end>/<unk>+w_i
vocabulary, Zipf references and a document-frequency table."""
import math

import torch


def random_butd_params(R, D, H, E, A, V, device, seed=1234):
    """Tensors keyed by the reference's state_dict names (without 'decoder.'), initialised like the reference:
    LSTMCell / Linear U(-1/sqrt(fan), 1/sqrt(fan)); embed and predict.weight U(-0.1, 0.1); predict.bias 0;
    weight_norm g = ||v|| per row (BUTD_Model.py:87-90)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)

    def U(shape, bound):
        return (torch.rand(shape, generator=g) * 2 - 1) * bound

    p = {}
    p["embed.0.weight"] = U((V, E), 0.1)
    k = 1.0 / math.sqrt(H)
    p["TD_atten.weight_ih"] = U((4 * H, H + D + E), k)
    p["TD_atten.weight_hh"] = U((4 * H, H), k)
    p["TD_atten.bias_ih"] = U((4 * H,), k)
    p["TD_atten.bias_hh"] = U((4 * H,), k)
    p["language_model.weight_ih"] = U((4 * H, D + H), k)
    p["language_model.weight_hh"] = U((4 * H, H), k)
    p["language_model.bias_ih"] = U((4 * H,), k)
    p["language_model.bias_hh"] = U((4 * H,), k)
    for name, (o, i) in (("atten.enc_att", (A, D)), ("atten.dec_att", (A, H)), ("atten.affine", (1, A))):
        v = U((o, i), 1.0 / math.sqrt(i))
        p[name + ".weight_v"] = v
        p[name + ".weight_g"] = v.norm(dim=1, keepdim=True)
        p[name + ".bias"] = U((o,), 1.0 / math.sqrt(i))
    v = U((V, H), 0.1)
    p["predict.weight_v"] = v
    p["predict.weight_g"] = v.norm(dim=1, keepdim=True)
    p["predict.bias"] = torch.zeros(V)
    return {k_: t.to(device=device, dtype=torch.float32).contiguous() for k_, t in p.items()}


def random_nic_params(E, H, V, device, seed=1234):
    """NIC decoder tensors by the reference's state_dict names, torch defaults as Models/NIC_Model.py:39-50 leaves them:
    Embedding N(0, 1); LSTMCell and Linear U(-1/sqrt(H), 1/sqrt(H)); weight_norm g = ||v|| per row."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    k = 1.0 / math.sqrt(H)
    U = lambda shape: (torch.rand(shape, generator=g) * 2 - 1) * k
    p = {"embed.weight": torch.randn((V, E), generator=g), "lstm.weight_ih": U((4 * H, E)), "lstm.weight_hh": U((4 * H, H)),
         "lstm.bias_ih": U((4 * H,)), "lstm.bias_hh": U((4 * H,))}
    v = U((V, H))
    p["predict.weight_v"], p["predict.weight_g"], p["predict.bias"] = v, v.norm(dim=1, keepdim=True), U((V,))
    return {k_: t.to(device=device, dtype=torch.float32).contiguous() for k_, t in p.items()}


def synthetic_references(n_img, vocab_words, seed=0, min_len=8, max_len=12):
    """5 references per image, length U{8..12}, Zipf(1.3) tokens over the vocabulary."""
    import numpy as np
    rng = np.random.RandomState(seed)
    V = len(vocab_words)
    gts = {}
    for i in range(n_img):
        refs = []
        for _ in range(5):
            L = rng.randint(min_len, max_len + 1)
            z = np.minimum(rng.zipf(1.3, size=L), V - 4) - 1 + 4      # skip the 4 special tokens
            refs.append(" ".join(vocab_words[j] for j in z))
        gts[i] = refs
    return gts


def document_frequency(gts):
    """The rule of PreProcess/CIDEr_idf_preproccess.py:41-83: df[ngram] = number of images whose references contain
    it (n = 1..4); ref_len = number of images."""
    df = {}
    for refs in gts.values():
        seen = set()
        for ref in refs:
            w = ref.split()
            for k in range(1, 5):
                for i in range(len(w) - k + 1):
                    seen.add(tuple(w[i:i + k]))
        for g in seen:
            df[g] = df.get(g, 0.0) + 1.0
    return {"document_frequency": df, "ref_len": len(gts)}
