"""Corpus CIDEr of the evaluation step that follows `Engine.eval_captions_json_generation`.

The reference scores the generated captions with `coco_eval` (COCO_Eval_Utils.py:15-35): pycocotools loads the annotation
file, a Java PTB tokeniser splits references and candidates, and `Cider().compute_score` (coco_caption/pycocoevalcap/cider/
cider.py:34-56 -> cider_scorer.py:96-195) returns the corpus CIDEr.  Neither pycocotools nor the Stanford jar is needed here:

  * `load_annotations` reads the COCO caption json directly ({image_id: [caption, ...]}, what COCO.imgToAnns holds);
  * `ptb_lite_tokenize` is a rule-based stand-in for `PTBTokenizer -preserveLines -lowerCase` followed by the reference's
    punctuation filter (ptbtokenizer.py:24-25,66-68).  PARITY UNPINNED: the jar cannot run in this image, so there are
    no golden vectors for the tokeniser itself (it agrees with PTB on plain captions; rare symbols may differ).  The rules that are
    implemented are the documented Penn-Treebank conventions (contractions n't / 's / 're / 've / 'll / 'd / 'm split off,
    cannot -> can not, gonna -> gon na, brackets to -lrb- / -rrb-, $ % & # as tokens, digit groups like 1,000 and 3:30
    kept whole, sentence-final periods split, abbreviations kept), each with an explicit expectation in
    tests/test_cpu_abi_and_host.py;
  * `Cider.compute_score` scores on the device with the CIDEr-D kernel of the SCST reward (csrc/ciderd.hip): the
    per-image formula is the same (clipped tf-idf cosine x Gaussian length penalty), only the document frequencies differ
    -- here they are counted over the evaluated references themselves (cider_scorer.py:96-107,164).  Scores are bit-exact
    against the reference scorer (tests/golden/corpus_cider_cases.json).
"""
import json
import re

import numpy as np
import torch

from .ciderd import CiderDReward
from .synth import document_frequency

# ptbtokenizer.py:24-25 -- compared AFTER lower-casing, so the bracket names of the list can never match (kept as is)
PUNCTUATIONS = ["''", "'", "``", "`", "-LRB-", "-RRB-", "-LCB-", "-RCB-", ".", "?", "!", ",", ":", "-", "--", "...", ";"]

_BRACKETS = {"(": "-lrb-", ")": "-rrb-", "[": "-lsb-", "]": "-rsb-", "{": "-lcb-", "}": "-rcb-"}
_ABBREV = {"mr.", "mrs.", "ms.", "dr.", "st.", "jr.", "sr.", "vs.", "etc.", "no.", "inc.", "co.", "ave.", "mt."}      # kept with their period
_CONTRACTION = re.compile(r"(?i)\b(can)(not)\b")
_SUFFIX = re.compile(r"(?i)([a-z0-9])('ll|'re|'ve|n't|'s|'m|'d)\b")
_SPLIT = re.compile(r"(\.\.\.|--|[\"(){}\[\]?!;$%&#]|(?<!\d)[,:]|[,:](?!\d)|``|'')")      # "1,000" and "3:30" stay whole


def ptb_lite_tokenize(sentence):
    """One caption -> lower-cased, space-joined PTB-style tokens without punctuation tokens."""
    s = sentence.replace("\n", " ").lower()
    s = _CONTRACTION.sub(r"\1 \2", s)
    s = re.sub(r"\b(gon|wan)(na)\b|\b(got)(ta)\b|\b(lem|gim)(me)\b", lambda m: " ".join(g for g in m.groups() if g), s)   # PTB: gon na, got ta, lem me
    s = _SUFFIX.sub(r"\1 \2", s)
    s = _SPLIT.sub(r" \1 ", s)
    out = []
    for tok in s.split():
        if tok == '"':
            tok = "''"
        tok = _BRACKETS.get(tok, tok)
        # a sentence-final period is its own token; periods inside abbreviations / numbers stay attached
        while len(tok) > 1 and tok.endswith(".") and tok not in _ABBREV and not re.fullmatch(r"\.+|([a-z]\.)+|\d+(\.\d+)+\.?", tok):
            tok = tok[:-1]
            out.append(tok)
            tok = "."
        # leading / trailing apostrophes are quote tokens; word-internal ones ("o'clock") stay
        while len(tok) > 1 and tok.startswith("'") and tok not in ("'ll", "'re", "'ve", "'s", "'m", "'d"):
            out.append("'")
            tok = tok[1:]
        if len(tok) > 1 and tok.endswith("'") and tok != "''":
            out.append(tok[:-1])
            tok = "'"
        out.append(tok)
    return " ".join(w for w in out if w not in PUNCTUATIONS)


def tokenize(captions_for_image):
    """PTBTokenizer.tokenize (ptbtokenizer.py:30-70): {id: [{'caption': str}, ...]} -> {id: [tokenised str, ...]}."""
    return {k: [ptb_lite_tokenize(c["caption"]) for c in v] for k, v in captions_for_image.items()}


def load_annotations(path):
    """{image_id: [{'caption': ...}, ...]} from a COCO caption annotation file (COCO.imgToAnns of pycocotools)."""
    data = json.load(open(path, "r", encoding="utf-8"))
    out = {}
    for ann in data["annotations"]:
        out.setdefault(ann["image_id"], []).append({"caption": ann["caption"]})
    return out


class Cider:
    """Cider (cider.py:17-56) on the device.  compute_score(gts, res): both {image id: [tokenised sentence, ...]} with one
    candidate per image and the same key order -> (corpus CIDEr, per-image float64 scores)."""

    MAX_TOKENS = 60          # csrc/ciderd.hip: one wave per hypothesis, at most 60 tokens
    BATCH = 1024

    def __init__(self, n=4, sigma=6.0, device="cuda:0"):
        if n != 4:
            raise ValueError("the device scorer implements the reference default n = 4")
        self._sigma = sigma
        self.device = torch.device(device)

    def method(self):
        return "CIDEr"

    def compute_score(self, gts, res):
        ids = list(gts.keys())
        assert list(res.keys()) == ids
        for i in ids:
            assert type(res[i]) is list and len(res[i]) == 1
            assert type(gts[i]) is list and len(gts[i]) > 0
        # corpus-local vocabulary: ids 0..3 stay reserved (0 terminates a hypothesis row)
        word2ix = {"<pad>": 0, "<sta>": 1, "<end>": 2, "<unk>": 3}
        for i in ids:
            for sent in gts[i] + res[i]:
                for w in sent.split():
                    if w not in word2ix:
                        word2ix[w] = len(word2ix)
        df = document_frequency({i: gts[i] for i in ids})
        scorer = CiderDReward(df["document_frequency"], df["ref_len"], word2ix, self.device, sigma=self._sigma)
        hyps = [[word2ix[w] for w in res[i][0].split()] for i in ids]
        T = max(1, max(len(h) for h in hyps))
        if T > self.MAX_TOKENS:
            raise ValueError("candidate caption with %d tokens: the device scorer handles at most %d" % (T, self.MAX_TOKENS))
        scores = np.zeros(len(ids), dtype=np.float64)
        for b0 in range(0, len(ids), self.BATCH):
            chunk = ids[b0:b0 + self.BATCH]
            gen = np.zeros((len(chunk), T), dtype=np.int64)
            for j, h in enumerate(hyps[b0:b0 + self.BATCH]):
                gen[j, :len(h)] = h
            gen_t = torch.from_numpy(gen).to(self.device)
            _, sc = scorer.reward(gen_t, torch.zeros_like(gen_t), gts, chunk, return_scores=True)
            scores[b0:b0 + len(chunk)] = sc[:len(chunk)].cpu().numpy()
        scorer.close()
        return float(np.mean(scores)), scores


def coco_eval(results, eval_caption_path, device="cuda:0"):
    """coco_eval (COCO_Eval_Utils.py:15-35) restricted to the metric Engine.training keeps (CIDEr, Engine.py:117-131):
    results = [{'image_id', 'caption'}, ...] as produced by eval_captions_json_generation."""
    anns = load_annotations(eval_caption_path)
    res = {}
    for r in results:
        res.setdefault(r["image_id"], []).append({"caption": r["caption"]})
    img_ids = list(res.keys())
    gts = tokenize({i: anns[i] for i in img_ids})
    res = tokenize({i: res[i][:1] for i in img_ids})
    score, _ = Cider(device=device).compute_score(gts, res)
    print("---------------Evaluation performance-----------------")
    print("%s: %.3f" % ("CIDEr", score))
    return score
