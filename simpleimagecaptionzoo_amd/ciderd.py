"""Host side of the device CIDEr-D reward (replaces Utils.py:319-367 get_self_critical_reward).

What depends on strings is prepared once on the host and kept resident on the device:
  * the document-frequency table (the pickle of PreProcess/CIDEr_idf_preproccess.py:78-82) becomes an
    open-addressing hash of id n-grams -> idf = log(ref_len) - log(max(1, df))   (ciderD_scorer.py:141-145);
  * every image's references are "cooked" once (n-gram tf-idf vectors, norms, bigram length; ciderD_scorer.py:
    128-153) and appended to a device-resident store; a batch names its images by store row -- the reference
    re-unpickles the table and re-cooks the references for every batch.
Reference words outside the vocabulary are distinct strings in the reference; they get private ids >= V here, so
they still count in the reference norms and can never match a hypothesis n-gram (hypotheses only contain
vocabulary ids).  Scoring itself runs in libicz (csrc/ciderd.hip) in float64.
"""
import ctypes as C
import threading
import time

import numpy as np
import torch

from ._lib import check, lib, ptr, stream_ptr
from .features import wait_event as _wait

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


class ReferenceCooker:
    """Host side of the scorer: the document-frequency table over id n-grams and the cooking of references (n-gram tf-idf
    vectors, norms, bigram length; ciderD_scorer.py:17-32, 128-153).  No device memory: usable without a GPU.

    document_frequency: {tuple of words: count}; ref_len: number of training images (both straight from the reference's
    '<dataset>-train.p' pickle); word2ix: the caption vocabulary."""

    def __init__(self, document_frequency, ref_len, word2ix):
        import threading
        self.word2ix = dict(word2ix)
        self.V = len(self.word2ix)
        self._ext = {}                       # out-of-vocabulary reference words -> private ids >= V (cache of the library's map)
        self._ext_lock = threading.Lock()
        # the library's word -> id map: it tokenises whole batches of references outside the interpreter lock (cook_images) and
        # owns the private ids of out-of-vocabulary words (one owner: _word_id below asks it)
        ws = list(self.word2ix.items())
        blob = [w.encode("utf-8") for w, _ in ws]
        off = np.zeros(len(ws) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in blob], out=off[1:])
        ids = np.asarray([i for _, i in ws], dtype=np.int32)
        self._vocab = C.c_void_p()
        check(lib().icz_ciderd_vocab_create(b"".join(blob), off.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), len(ws), self.V,
                                            C.byref(self._vocab)))
        self.log_ref_len = float(np.log(float(ref_len)))
        self._df = document_frequency
        self._log_cache = {}
        # ---- df table over id n-grams.  Words outside the caption vocabulary get private ids >= V right here, so that the
        #      n-grams containing them keep their document frequency for the reference weights (hypotheses never look them up)
        # (one pass of plain dict look-ups: ~4 s per million n-grams, once per run -- COCO14's table has ~3 M; the reference pays an
        # unpickle of the same table for every batch, Utils.py:359)
        n = len(document_frequency)
        keys = np.full((n, 4), -1, dtype=np.int32)
        cnts = np.empty(n, dtype=np.float64)
        get, oov = self.word2ix.get, self._word_id
        for i, (gram, cnt) in enumerate(document_frequency.items()):
            row = [get(w, -2) for w in gram]
            if -2 in row:
                row = [oov(w) for w in gram]
            keys[i, :len(row)] = row
            cnts[i] = cnt
        uniq, inv = np.unique(cnts, return_inverse=True)                  # the scalar formula per distinct count: the same bits as _idf()
        vals = np.asarray([self._idf(float(c)) for c in uniq], dtype=np.float64)[inv] if n else np.zeros(0)
        cap = 2
        while cap < 2 * max(n, 1):
            cap *= 2
        tkeys = np.full((cap, 4), -1, dtype=np.int32)
        tidf = np.zeros(cap, dtype=np.float64)
        if n:
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
        self.keys_host, self.idf_host, self.cap = tkeys, tidf, cap

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
            if i is None:                    # the library hands out the private ids (thread-safe); cached here
                b = w.encode("utf-8")
                out = C.c_int32()
                check(lib().icz_ciderd_vocab_oov_id(self._vocab, b, len(b), C.byref(out)))
                i = self._ext[w] = int(out.value)
        return i

    def __del__(self):
        try:
            if self._vocab:
                lib().icz_ciderd_vocab_destroy(self._vocab)
                self._vocab = C.c_void_p()
        except Exception:
            pass

    def cook_image(self, refs):
        """refs: list of reference strings of one image -> (ent_ptr, keys, order, w, norm, length).  The Python statement of
        what icz_ciderd_cook_host does (cook_images below); kept as its checker (tests/test_cpu_abi_and_host.py)."""
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

    def cook_images(self, refs_per_image, as_block=False):
        """[[reference strings of image 0], ...] -> one (ent_ptr, keys, order, w, norm, length) tuple per image, through the
        library's host cooker (one call for all of them, outside the GIL).  as_block: the arrays of all the images as ONE block
        instead ({"nref": references per image, "ep": entry pointer per reference (+1), "key", "ord", "w", "norm", "len"}): what
        the device store appends, without cutting the arrays per image and joining them again."""
        flat = [ref for refs in refs_per_image for ref in refs]
        nref = [len(refs) for refs in refs_per_image]
        n_refs = len(flat)
        text = "\n".join(flat)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        ne = C.c_int64()
        if n_refs and text.isascii() and text.count("\n") == n_refs - 1:
            # the whole batch in one library call, tokenisation included: the interpreter lock is held for the join above only
            raw = text.encode("ascii")
            max_ent = 4 * (len(raw) // 2 + n_refs + 1)
            K, O = np.empty((max_ent, 4), np.int32), np.empty(max_ent, np.int32)
            W, EP = np.empty(max_ent, np.float64), np.empty(n_refs + 1, np.int32)
            N, L = np.empty((n_refs, 4), np.float64), np.empty(n_refs, np.int32)
            check(lib().icz_ciderd_cook_text(self._vocab, P(self.keys_host), P(self.idf_host), self.cap, self.log_ref_len, raw, len(raw), n_refs,
                                             max_ent, P(K), P(O), P(W), P(EP), P(N), P(L), C.byref(ne)))
        else:       # non-ASCII text or a newline inside a reference: str.split() here, ids to the library
            tok, tptr = [], [0]
            wid = self._word_id
            for ref in flat:
                tok += [wid(w) for w in ref.split()]
                tptr.append(len(tok))
            tok = np.asarray(tok if tok else [0], dtype=np.int32)
            tptr = np.asarray(tptr, dtype=np.int32)
            max_ent = 4 * max(1, tok.size)
            K, O = np.empty((max_ent, 4), np.int32), np.empty(max_ent, np.int32)
            W, EP = np.empty(max_ent, np.float64), np.empty(n_refs + 1, np.int32)
            N, L = np.empty((max(1, n_refs), 4), np.float64), np.empty(max(1, n_refs), np.int32)
            check(lib().icz_ciderd_cook_host(P(self.keys_host), P(self.idf_host), self.cap, self.log_ref_len, P(tok), P(tptr), n_refs, max_ent,
                                             P(K), P(O), P(W), P(EP), P(N), P(L), C.byref(ne)))
        if as_block:
            e = int(ne.value)
            return {"nref": np.asarray(nref, np.int32), "ep": EP, "key": K[:e], "ord": O[:e], "w": W[:e], "norm": N[:n_refs], "len": L[:n_refs]}
        out, r0 = [], 0
        for nr in nref:
            e0, e1 = int(EP[r0]), int(EP[r0 + nr])
            out.append(((EP[r0:r0 + nr + 1] - e0).astype(np.int32), K[e0:e1].copy(), O[e0:e1].copy(), W[e0:e1].copy(), N[r0:r0 + nr].copy(),
                        L[r0:r0 + nr].copy()))
            r0 += nr
        return out


class CiderDReward:
    """Device-resident CIDEr-D scorer for SCST (arguments as ReferenceCooker's)."""

    def __init__(self, document_frequency, ref_len, word2ix, device="cuda:0", sigma=6.0, store_images=1 << 15):
        """store_images: images the device-resident reference store holds before its first growth step."""
        self.device = torch.device(device)
        self._store_images = max(16, int(store_images))
        self.cooker = ReferenceCooker(document_frequency, ref_len, word2ix)
        self.V = self.cooker.V
        self._keys = torch.from_numpy(self.cooker.keys_host).to(self.device)
        self._idf_t = torch.from_numpy(self.cooker.idf_host).to(self.device)
        pen = np.array([np.e ** (-(float(d) ** 2) / (2 * sigma ** 2)) for d in range(64)], dtype=np.float64)
        self._pen = torch.from_numpy(pen).to(self.device)
        self._h = C.c_void_p()
        check(lib().icz_ciderd_create(ptr(self._keys), ptr(self._idf_t), self.cooker.cap, self.cooker.log_ref_len, ptr(self._pen),
                                      C.byref(self._h)))
        self.persistent = False
        self._out = {}
        self._blocks = []                    # cooked reference blocks (host) waiting for their upload, in cooking order
        self._pending = {}                   # image id -> its block, until the block is in the store
        self._cooking = set()                # image ids being cooked right now (by the loader's worker thread or the main thread)
        self._lock = threading.Lock()        # guards _slot / _blocks / _pending / _cooking: prepare() runs on a loader thread beside _append()
        # pinned / device staging of the block uploads, allocated here (a pinned allocation costs milliseconds and synchronises the
        # device: not something for the first training steps)
        self._up_ring, self._up_i = [], 0
        for _ in range(8):
            self._up_ring.append((torch.empty(1 << 20, dtype=torch.uint8).pin_memory(), torch.empty(1 << 20, dtype=torch.uint8, device=self.device), None))
        self._store_init()

    def close(self):
        if self._h:
            lib().icz_ciderd_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def cook_image(self, refs):
        return self.cooker.cook_image(refs)

    # ---- device-resident reference store -----------------------------------------------------
    # Every image's cooked references are uploaded ONCE, appended to seven growing device arrays (CSR over stored images ->
    # references -> n-gram entries); a batch is then just an int32 list of store rows.  Steady state (every epoch after the
    # first, or after preload()): no cooking, no CSR assembly, one 4 B-per-image upload from pinned memory.  The reference
    # re-cooks the batch's references on every call (ciderD.py:41-52).
    def _store_init(self):
        dev = self.device
        self._n_img = self._n_ref = self._n_ent = 0
        self._slot = {}                      # image id -> store row
        z = lambda n, dt, *rest: torch.zeros((n,) + rest, dtype=dt, device=dev)
        # sized for `store_images` images up front (default 32 k: 5 references of ~25 n-gram entries each, 130 MB of a 288 GB
        # device); a growth step re-allocates and copies three to seven arrays (tens of ms through hipMalloc), so they are made
        # rare: x4 per step
        n = self._store_images
        self._st = {"irp": z(n + 1, torch.int32), "rep": z(4 * n + 1, torch.int32), "key": z(128 * n, torch.int32, 4),
                    "ord": z(128 * n, torch.int32), "w": z(128 * n, torch.float64), "norm": z(4 * n, torch.float64, 4),
                    "len": z(4 * n, torch.int32)}
        self._idx_ring, self._idx_ev, self._idx_i, self._idx_dev = [], [], 0, {}

    def _grow(self, name, need):
        t = self._st[name]
        if t.shape[0] >= need:
            return
        cap = t.shape[0]
        while cap < need:
            cap *= 4
        new = torch.zeros((cap,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        new[:t.shape[0]] = t
        self._st[name] = new

    def prepare(self, img_ids, gts):
        """Cook the references of images not seen before (host only, thread-safe: a loader's worker thread calls this for
        batch i+1 while batch i is on the device, features.DevicePrefetcher(on_batch=...); the cooker runs outside the GIL).
        The new images of one call form one block, which _append uploads as a unit."""
        with self._lock:
            new = [i for i in dict.fromkeys(img_ids) if i not in self._slot and i not in self._pending and i not in self._cooking]
            self._cooking.update(new)
        if not new:
            return
        try:
            blk = self.cooker.cook_images([gts[i] for i in new], as_block=True)       # outside the lock (and, in C++, outside the GIL)
            blk["ids"] = new
            with self._lock:
                self._blocks.append(blk)
                for i in new:
                    self._pending[i] = blk
        finally:
            with self._lock:
                self._cooking.difference_update(new)

    def preload(self, gts):
        """Cook and upload the references of a whole dataset split ({image id: [reference strings]}) ahead of training."""
        ids = list(gts.keys())
        for lo in range(0, len(ids), 1024):
            self._append(ids[lo:lo + 1024], gts)

    def _append(self, img_ids, gts):
        """Make sure every image of `img_ids` is in the device store: cook what nobody has cooked, wait for what the loader's worker
        thread is cooking right now, then upload the pending blocks that hold any of them."""
        while True:
            self.prepare(img_ids, gts)
            with self._lock:
                missing = [i for i in img_ids if i not in self._slot and i not in self._pending]
                if not missing:
                    need = {id(self._pending[i]) for i in img_ids if i in self._pending}
                    blocks = [b for b in self._blocks if id(b) in need]
                    self._blocks = [b for b in self._blocks if id(b) not in need]
                    break
            time.sleep(0.0002)
        for blk in blocks:
            self._upload(blk)

    def _upload(self, blk):
        """One block of cooked references -> the device store: the seven arrays go through ONE pinned staging buffer and one
        asynchronous H2D copy, then device-side copies into the (growing) store arrays -- nothing here waits for the device."""
        ids, nref = blk["ids"], blk["nref"]
        n_new, n_ref, n_ent = len(ids), int(blk["len"].shape[0]), int(blk["ord"].shape[0])
        irp = self._n_ref + np.cumsum(nref, dtype=np.int64).astype(np.int32)           # reference end per image
        rep = (self._n_ent + blk["ep"][1:n_ref + 1].astype(np.int64)).astype(np.int32)   # entry end per reference
        parts = (("irp", irp, self._n_img + 1), ("rep", rep, self._n_ref + 1), ("key", blk["key"], self._n_ent), ("ord", blk["ord"], self._n_ent),
                 ("w", blk["w"], self._n_ent), ("norm", blk["norm"], self._n_ref), ("len", blk["len"], self._n_ref))
        self._grow("irp", self._n_img + n_new + 1)
        self._grow("rep", self._n_ref + n_ref + 1)
        for name in ("norm", "len"):
            self._grow(name, self._n_ref + n_ref)
        for name in ("key", "ord", "w"):
            self._grow(name, self._n_ent + n_ent)
        total = sum((a.nbytes + 15) // 16 * 16 for _, a, _ in parts)
        k = self._up_i = (self._up_i + 1) % len(self._up_ring)
        host, dev, ev = self._up_ring[k]
        if host is None or host.numel() < total:
            cap = max(1 << 20, 1 << (total - 1).bit_length())
            host, dev, ev = torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device), None
        if ev is not None:
            _wait(ev)                                               # the copy issued len(ring) blocks ago has left this buffer
        hv, off, views = host.numpy(), 0, []
        for name, a, at in parts:
            a = np.ascontiguousarray(a)
            hv[off:off + a.nbytes] = a.view(np.uint8).reshape(-1)
            views.append((name, off, a.nbytes, at, a.shape))
            off += (a.nbytes + 15) // 16 * 16
        dev[:off].copy_(host[:off], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._up_ring[k] = (host, dev, ev)
        for name, o, nb, at, shape in views:
            dst = self._st[name]
            src = dev[o:o + nb].view(dst.dtype).view(shape)
            dst[at:at + shape[0]].copy_(src, non_blocking=True)
        with self._lock:
            for j, i in enumerate(ids):
                self._slot[i] = self._n_img + j
                self._pending.pop(i, None)
        self._n_img += n_new
        self._n_ref += n_ref
        self._n_ent += n_ent

    def _slots(self, img_ids, gts):
        """Store rows of the batch as a device int32 tensor (uploading first whatever the store does not hold yet)."""
        slot = self._slot
        try:
            rows = [slot[i] for i in img_ids]
        except KeyError:
            self._append(img_ids, gts)
            rows = [slot[i] for i in img_ids]
        B = len(rows)
        if not self._idx_ring or self._idx_ring[0].numel() < B:
            self._idx_ring = [torch.zeros(max(B, 256), dtype=torch.int32).pin_memory() for _ in range(8)]
            self._idx_ev = [None] * 8
        k = self._idx_i = (self._idx_i + 1) % 8
        if self._idx_ev[k] is not None:
            _wait(self._idx_ev[k])                                  # the copy issued 8 batches ago has left this buffer
        host = self._idx_ring[k]
        host.numpy()[:B] = rows
        dev = self._idx_dev.get(B)
        if dev is None:
            dev = self._idx_dev[B] = torch.zeros(B, dtype=torch.int32, device=self.device)
        dev.copy_(host[:B], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._idx_ev[k] = ev
        return dev

    # ---- scoring ---------------------------------------------------------------------------------
    def reward(self, gen, greedy, ground_truth, img_ids, return_scores=False):
        """get_self_critical_reward(gen_result, greedy_res, ground_truth, img_ids, ...) -> float32 (B, T) on the
        device (the reference returns a CPU tensor and Engine moves it to the device, Engine.py:266)."""
        B, T = gen.shape
        gen = gen.to(device=self.device, dtype=torch.int64).contiguous()
        greedy = greedy.to(device=self.device, dtype=torch.int64).contiguous()
        idx = self._slots(list(img_ids), ground_truth)
        st = self._st
        if self.persistent:       # stable addresses for hipGraph replay downstream (overwritten by the next call)
            key = (B, T)
            if key not in self._out:
                self._out[key] = (torch.empty(B, T, dtype=torch.float32, device=self.device),
                                  torch.empty(2 * B, dtype=torch.float64, device=self.device))
            reward, scores = self._out[key]
        else:
            reward = torch.empty(B, T, dtype=torch.float32, device=self.device)
            scores = torch.empty(2 * B, dtype=torch.float64, device=self.device)
        check(lib().icz_ciderd_reward_indexed(self._h, ptr(gen), ptr(greedy), B, T, ptr(idx), ptr(st["irp"]), ptr(st["rep"]),
                                              ptr(st["key"]), ptr(st["ord"]), ptr(st["w"]), ptr(st["norm"]), ptr(st["len"]),
                                              ptr(reward), ptr(scores), stream_ptr()))
        return (reward, scores) if return_scores else reward


# ---- the reference's own entry point -----------------------------------------------------------------------------------
_SCORERS = {}            # (dataset name, device) -> (caption vocabulary, CiderDReward): built once, kept for the whole run
_SCORERS_LOCK = threading.Lock()


def get_self_critical_reward(gen_result, greedy_res, ground_truth, img_ids, caption_vocab, dataset_name, cider_weight=1):
    """Utils.py:319-367 with its signature and its return value: FloatTensor (batch_size, max_len) on the CPU, every column of row i
    = cider_weight * (CIDEr-D(sampled caption i) - CIDEr-D(greedy caption i)).  With this, the reference's Engine needs one edit
    to stop paying for the reward (`from simpleimagecaptionzoo_amd.ciderd import get_self_critical_reward`):
    the reference builds `CiderD(df='<dataset>-train')` -- an unpickle of the whole document-frequency table -- and re-cooks every
    reference for EVERY batch (Utils.py:359, ciderD_scorer.py:79-83); here the table is read once per (dataset, device) from the same
    file, `cider/data/<dataset_name>-train.p` relative to the working directory (ciderD_scorer.py:80), hashed onto the device, and
    each image's references are cooked once and stay there.  Scores are float64 on the device (csrc/ciderd.hip); the difference,
    the weight and the float32 cast are done here exactly as the reference does them (Utils.py:361-365)."""
    import os
    import pickle
    dev = gen_result.device if (torch.is_tensor(gen_result) and gen_result.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    key = (str(dataset_name), str(dev))
    with _SCORERS_LOCK:
        hit = _SCORERS.get(key)
        if hit is None or hit[0] is not caption_vocab:
            path = os.path.join("cider/data", "%s-train.p" % dataset_name)
            with open(path, "rb") as f:
                df = pickle.load(f, encoding="latin1")
            t0 = time.perf_counter()
            scorer = CiderDReward(df["document_frequency"], df["ref_len"], caption_vocab.word2ix, dev)
            scorer.build_seconds = time.perf_counter() - t0
            hit = _SCORERS[key] = (caption_vocab, scorer)
    scorer = hit[1]
    batch_size = gen_result.shape[0]
    _, scores = scorer.reward(gen_result, greedy_res, ground_truth, list(img_ids), return_scores=True)
    cider_scores = scores.cpu().numpy()                                  # float64 (2 B): sampled rows, then greedy rows
    scores = cider_weight * cider_scores
    scores = scores[:batch_size] - scores[batch_size:]
    rewards = np.repeat(scores[:, np.newaxis], gen_result.shape[1], 1)
    return torch.from_numpy(rewards).float()
