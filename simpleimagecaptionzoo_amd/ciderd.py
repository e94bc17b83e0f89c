"""Host side of the device CIDEr-D reward (replaces Utils.py:319-367 get_self_critical_reward).

What depends on strings is prepared once on the host and kept resident on the device:
  * the document-frequency table (the pickle of PreProcess/CIDEr_idf_preproccess.py:78-82) becomes an
    open-addressing hash of id n-grams -> idf = log(ref_len) - log(max(1, df))   (ciderD_scorer.py:141-145);
  * every image's references are "cooked" once (n-gram tf-idf vectors, norms, bigram length; ciderD_scorer.py:
    128-153) and cached -- the reference re-unpickles the table and re-cooks the references for every batch.
Reference words outside the vocabulary are distinct strings in the reference; they get private ids >= V here, so
they still count in the reference norms and can never match a hypothesis n-gram (hypotheses only contain
vocabulary ids).  Scoring itself runs in libicz (csrc/ciderd.hip) in float64.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr

_FNV_OFF, _FNV_PRIME = np.uint32(2166136261), np.uint32(16777619)


def _hash_keys(keys):
    """numpy twin of ngram_hash() in csrc/ciderd.hip; keys int32 [n,4]."""
    k = keys.astype(np.int64).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = np.full(k.shape[0], _FNV_OFF, dtype=np.uint32)
        for j in range(4):
            h = (h ^ k[:, j]) * _FNV_PRIME
        h ^= h >> np.uint32(15)
    return h


def _ngrams(ids, n=4):
    """precook(): ordered {ngram tuple: count}, insertion order = (k ascending, position ascending)."""
    counts = {}
    for k in range(1, n + 1):
        for i in range(len(ids) - k + 1):
            g = tuple(ids[i:i + k])
            counts[g] = counts.get(g, 0) + 1
    return counts


class CiderDReward:
    """Device-resident CIDEr-D scorer for SCST.

    document_frequency: {tuple of words: count}; ref_len: number of training images (both straight from the
    reference's '<dataset>-train.p' pickle); word2ix: the caption vocabulary."""

    def __init__(self, document_frequency, ref_len, word2ix, device="cuda:0", sigma=6.0):
        self.device = torch.device(device)
        self.word2ix = dict(word2ix)
        self.V = len(self.word2ix)
        self._ext = {}                       # out-of-vocabulary reference words -> private ids >= V
        self.log_ref_len = float(np.log(float(ref_len)))
        self._df = document_frequency
        self._log_cache = {}
        # ---- df table restricted to n-grams made of vocabulary words (only those can be looked up by hypotheses)
        keys, vals = [], []
        w2i = self.word2ix
        for gram, cnt in document_frequency.items():
            try:
                ids = [w2i[w] for w in gram]
            except KeyError:
                continue
            keys.append(ids + [-1] * (4 - len(ids)))
            vals.append(self._idf(cnt))
        n = len(keys)
        cap = 2
        while cap < 2 * max(n, 1):
            cap *= 2
        tkeys = np.full((cap, 4), -1, dtype=np.int32)
        tidf = np.zeros(cap, dtype=np.float64)
        if n:
            keys = np.asarray(keys, dtype=np.int32)
            vals = np.asarray(vals, dtype=np.float64)
            h = _hash_keys(keys)
            probe = np.zeros(n, dtype=np.uint32)
            todo = np.arange(n)
            mask = np.uint32(cap - 1)
            while todo.size:
                with np.errstate(over="ignore"):
                    s = ((h[todo] + probe[todo]) & mask).astype(np.int64)
                empty = tkeys[s, 0] == -1
                cand, cs = todo[empty], s[empty]
                _, first = np.unique(cs, return_index=True)
                win = cand[first]
                tkeys[cs[first]] = keys[win]
                tidf[cs[first]] = vals[win]
                placed = np.zeros(n, dtype=bool)
                placed[win] = True
                todo = todo[~placed[todo]]
                probe[todo] += 1
        self._keys = torch.from_numpy(tkeys).to(self.device)
        self._idf_t = torch.from_numpy(tidf).to(self.device)
        pen = np.array([np.e ** (-(float(d) ** 2) / (2 * sigma ** 2)) for d in range(64)], dtype=np.float64)
        self._pen = torch.from_numpy(pen).to(self.device)
        self._h = C.c_void_p()
        check(lib().icz_ciderd_create(ptr(self._keys), ptr(self._idf_t), cap, self.log_ref_len, ptr(self._pen),
                                      C.byref(self._h)))
        self.persistent = False
        self._out = {}
        self._cooked = {}                    # image id -> cooked reference arrays (host)
        self._batch_cache = {}

    def close(self):
        if self._h:
            lib().icz_ciderd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-side cooking ---------------------------------------------------------------------
    def _idf(self, cnt):
        v = self._log_cache.get(cnt)
        if v is None:
            v = self.log_ref_len - float(np.log(max(1.0, cnt)))       # ciderD_scorer.py:141-145
            self._log_cache[cnt] = v
        return v

    def _word_id(self, w):
        i = self.word2ix.get(w)
        if i is None:
            i = self._ext.get(w)
            if i is None:
                i = self.V + len(self._ext)
                self._ext[w] = i
        return i

    def cook_image(self, refs):
        """refs: list of reference strings of one image -> (ent_ptr, keys, order, w, norm, length)."""
        ent_ptr, keys, order, ws, norms, lens = [0], [], [], [], [], []
        for ref in refs:
            words = ref.split()
            ids = [self._word_id(w) for w in words]
            norm = [0.0] * 4
            length = 0
            counts_w = _ngrams(words)
            for (gw, tf), gi in zip(counts_w.items(), _ngrams(ids).keys()):
                k = len(gw)
                w = float(tf) * self._idf(self._df.get(gw, 0.0))
                keys.append(list(gi) + [-1] * (4 - k))
                order.append(k)
                ws.append(w)
                norm[k - 1] += pow(w, 2)
                if k == 2:
                    length += tf
            norms.append([float(np.sqrt(x)) for x in norm])
            lens.append(length)
            ent_ptr.append(len(keys))
        return (np.asarray(ent_ptr, np.int32), np.asarray(keys, np.int32).reshape(-1, 4), np.asarray(order, np.int32),
                np.asarray(ws, np.float64), np.asarray(norms, np.float64).reshape(-1, 4), np.asarray(lens, np.int32))

    def _batch(self, img_ids, gts):
        key = tuple(img_ids)
        hit = self._batch_cache.get(key)
        if hit is not None:
            return hit
        img_ref_ptr, ref_ent_ptr = [0], [0]
        K, O, W, N, L = [], [], [], [], []
        for i in img_ids:
            c = self._cooked.get(i)
            if c is None:
                c = self.cook_image(gts[i])
                self._cooked[i] = c
            ep, k, o, w, nrm, ln = c
            base = ref_ent_ptr[-1]
            ref_ent_ptr.extend((base + ep[1:]).tolist())
            img_ref_ptr.append(img_ref_ptr[-1] + len(ln))
            K.append(k); O.append(o); W.append(w); N.append(nrm); L.append(ln)
        dev = self.device
        out = (torch.tensor(img_ref_ptr, dtype=torch.int32, device=dev),
               torch.tensor(ref_ent_ptr, dtype=torch.int32, device=dev),
               torch.from_numpy(np.concatenate(K).astype(np.int32)).to(dev),
               torch.from_numpy(np.concatenate(O).astype(np.int32)).to(dev),
               torch.from_numpy(np.concatenate(W)).to(dev),
               torch.from_numpy(np.concatenate(N)).to(dev),
               torch.from_numpy(np.concatenate(L).astype(np.int32)).to(dev))
        if len(self._batch_cache) < 4096:
            self._batch_cache[key] = out
        return out

    # ---- scoring ---------------------------------------------------------------------------------
    def reward(self, gen, greedy, ground_truth, img_ids, return_scores=False):
        """get_self_critical_reward(gen_result, greedy_res, ground_truth, img_ids, ...) -> float32 (B, T) on the
        device (the reference returns a CPU tensor and Engine moves it to the device, Engine.py:266)."""
        B, T = gen.shape
        gen = gen.to(device=self.device, dtype=torch.int64).contiguous()
        greedy = greedy.to(device=self.device, dtype=torch.int64).contiguous()
        irp, rep, K, O, W, N, L = self._batch(list(img_ids), ground_truth)
        if self.persistent:       # stable addresses for hipGraph replay downstream (overwritten by the next call)
            key = (B, T)
            if key not in self._out:
                self._out[key] = (torch.empty(B, T, dtype=torch.float32, device=self.device),
                                  torch.empty(2 * B, dtype=torch.float64, device=self.device))
            reward, scores = self._out[key]
        else:
            reward = torch.empty(B, T, dtype=torch.float32, device=self.device)
            scores = torch.empty(2 * B, dtype=torch.float64, device=self.device)
        check(lib().icz_ciderd_reward(self._h, ptr(gen), ptr(greedy), B, T, ptr(irp), ptr(rep), ptr(K), ptr(O), ptr(W),
                                      ptr(N), ptr(L), ptr(reward), ptr(scores), stream_ptr()))
        return (reward, scores) if return_scores else reward
