"""Oracle: CIDEr-D with a precomputed document-frequency table, and the SCST reward built on it.

Restates cider/pyciderevalcap/ciderD/ciderD_scorer.py:17-32 (precook), :127-206 (compute_cider) and
Utils.py:319-367 (get_self_critical_reward) in plain Python/float64.  Summation orders follow the
reference's dict-insertion orders so results are bit-identical in float64.
TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
"""
import math

import numpy as np

PAD, STA, END, UNK = 0, 1, 2, 3


def ngram_counts(words, n=4):
    """precook(): ordered dict {ngram tuple: count}; insertion order = (k ascending, position ascending)."""
    counts = {}
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            g = tuple(words[i:i + k])
            counts[g] = counts.get(g, 0) + 1
    return counts


class DocFreq:
    """The pickled table of PreProcess/CIDEr_idf_preproccess.py:78-82: {'document_frequency', 'ref_len'}."""

    def __init__(self, document_frequency, ref_len):
        self.df = dict(document_frequency)
        self.log_ref_len = float(np.log(float(ref_len)))          # ciderD_scorer.py:82

    @classmethod
    def from_json(cls, js):
        return cls({tuple(k): v for k, v in js["document_frequency"]}, js["ref_len"])

    def idf(self, gram):
        return self.log_ref_len - float(np.log(max(1.0, self.df.get(gram, 0.0))))   # :141-145


def tfidf(counts, docfreq, n=4):
    """counts2vec(), ciderD_scorer.py:128-153.  'length' is the number of BIGRAMS (index n==1)."""
    vec = [dict() for _ in range(n)]
    norm = [0.0] * n
    length = 0
    for gram, tf in counts.items():
        k = len(gram) - 1
        w = float(tf) * docfreq.idf(gram)
        vec[k][gram] = w
        norm[k] += pow(w, 2)
        if k == 1:
            length += tf
    return vec, [float(np.sqrt(x)) for x in norm], length


def similarity(vh, vr, nh, nr, lh, lr, n=4, sigma=6.0):
    """sim(), ciderD_scorer.py:155-183: clipped cosine times Gaussian length penalty."""
    delta = float(lh - lr)
    val = [0.0] * n
    for k in range(n):
        for gram, w in vh[k].items():
            r = vr[k].get(gram, 0.0)
            val[k] += min(w, r) * r
        if nh[k] != 0 and nr[k] != 0:
            val[k] /= (nh[k] * nr[k])
        val[k] *= np.e ** (-(delta ** 2) / (2 * sigma ** 2))
    return val


def ciderd_scores(hyps, refs_per_hyp, docfreq, n=4, sigma=6.0):
    """compute_cider(), ciderD_scorer.py:185-206.  hyps: list[str]; refs_per_hyp: list[list[str]]."""
    out = []
    for hyp, refs in zip(hyps, refs_per_hyp):
        vh, nh, lh = tfidf(ngram_counts(hyp.split(), n), docfreq, n)
        score = np.array([0.0] * n)
        for ref in refs:
            vr, nr, lr = tfidf(ngram_counts(ref.split(), n), docfreq, n)
            score += np.array(similarity(vh, vr, nh, nr, lh, lr, n, sigma))
        s = np.mean(score)
        s /= len(refs)
        s *= 10.0
        out.append(float(s))
    return np.array(out)


def sampled_sentence(ids, ix2word):
    """Utils.py:337-346: strip trailing zeros (an all-zero row keeps ONE element -> '<pad>')."""
    ids = list(ids)
    end = 0
    for e in range(len(ids) - 1, -1, -1):
        end = e
        if ids[e] != 0:
            break
    return " ".join(ix2word[int(w)] for w in ids[:end + 1])


def greedy_sentence(ids, ix2word):
    """Utils.py:348-356: cut at the first <end>."""
    ws = []
    for w in ids:
        if ix2word[int(w)] == "<end>":
            break
        ws.append(ix2word[int(w)])
    return " ".join(ws)


def self_critical_reward(gen, greedy, gts, img_ids, ix2word, docfreq):
    """get_self_critical_reward(), Utils.py:319-367 -> float32 (B, max_len)."""
    gen, greedy = np.asarray(gen), np.asarray(greedy)
    B, T = gen.shape
    hyps = [sampled_sentence(gen[b], ix2word) for b in range(B)] + \
           [greedy_sentence(greedy[b], ix2word) for b in range(B)]
    refs = [gts[img_ids[b]] for b in range(B)] * 2
    sc = ciderd_scores(hyps, refs, docfreq)
    diff = sc[:B] - sc[B:]
    return np.repeat(diff[:, None], T, 1).astype(np.float32)


def captions_json(ids, image_ids, ix2word):
    """Engine.eval_captions_json_generation's id->word loop, Engine.py:288-299."""
    out = []
    ids = np.asarray(ids)
    for b in range(ids.shape[0]):
        ws = []
        for w in ids[b]:
            word = ix2word[int(w)]
            if word == "<end>":
                break
            if word != "<sta>":
                ws.append(word)
        out.append({"image_id": int(image_ids[b]), "caption": " ".join(ws)})
    return out


def corpus_document_frequency(gts, n=4):
    """compute_doc_freq(), coco_caption/pycocoevalcap/cider/cider_scorer.py:96-107: df[ngram] = number of images whose
    references contain it."""
    df = {}
    for refs in gts.values():
        seen = set()
        for ref in refs:
            seen.update(ngram_counts(ref.split(), n).keys())
        for g in seen:
            df[g] = df.get(g, 0.0) + 1.0
    return df


def corpus_cider(gts, res, n=4, sigma=6.0):
    """Cider.compute_score(gts, res), cider.py:34-56 -> CiderScorer.compute_score, cider_scorer.py:185-195: the corpus
    CIDEr of the evaluation path (eval.py:52-66).  df and log(ref_len) come from the evaluated references themselves
    (cider_scorer.py:164); the per-image formula is the one of `ciderd_scores`.  Returns (mean, per-image scores)."""
    ids = list(gts.keys())
    assert list(res.keys()) == ids
    docfreq = DocFreq(corpus_document_frequency(gts, n), len(ids))
    scores = ciderd_scores([res[i][0] for i in ids], [gts[i] for i in ids], docfreq, n, sigma)
    return float(np.mean(scores)), scores
