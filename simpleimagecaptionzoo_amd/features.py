"""Feature input pipeline for the bottom-up models.

The reference reads one zlib-compressed `.npz` per image and sample (`Datasets.py:54-62,101,139`:
`np.load('<supp_dir>/fixed_bu_feat/<img_id>.npz')['feat']`, (36, 2048) fp32) and copies the stacked batch from pageable
memory (`BUTD_Engine.py:45`).  At several thousand captions/s per GPU that costs more than the decode itself, so the
features are packed once into one flat, memory-mapped fp32 file and streamed to the GPU from pinned double buffers:

  pack_npz_dir()        per-image .npz (+ .npy boxes)  ->  <name>.feats.npy ([N, R, D] fp32, uncompressed) + <name>.index.json
  PackedFeatureStore    memory-mapped rows by image id (no decompression, no per-file open)
  DevicePrefetcher      wraps any iterable of the reference's batch tuples: a worker thread gathers batch i+1 into a pinned
                        buffer and copies it on a side stream while batch i is being decoded; it yields the same tuples with
                        `supp_info_datas` replaced by {'bu_feats': device tensor, 'bu_bboxes': [...]}, which the Engines'
                        `modify_visual_inputs` passes through unchanged.
The bytes are the reference's (fp32, no re-quantisation), so results do not change.  'fixed' feature sets (36 rows per image)
are stored as one [N, R, D] array; 'adaptive' ones (10..100 rows per image, Datasets.py:59-61) as the concatenation of the
images' rows with an offset table, and a batch is padded with zero rows to its largest count exactly as
AoA_Engine.modify_visual_inputs does (AoA_Engine.py:33-40), the counts travelling with it as `bu_counts`.
"""
import json
import itertools
import os

import numpy as np
import torch


def _feat_path(supp_dir, kind, i):
    return os.path.join(supp_dir, "%s_bu_feat/%s.npz" % (kind, i))


def pack_npz_dir(supp_dir, img_ids, out_prefix, kind="fixed"):
    """Pack `<supp_dir>/<kind>_bu_feat/<id>.npz['feat']` (and `<kind>_bu_bbox/<id>.npy`) of `img_ids` into
    `<out_prefix>.feats.npy`, `<out_prefix>.boxes.npy`, `<out_prefix>.index.json`.  kind='fixed': every image has the same
    (R, D) -> [N, R, D]; kind='adaptive': (n_i, D) per image -> [sum n_i, D] + offsets."""
    img_ids = list(img_ids)
    if kind == "adaptive":
        counts = [int(np.load(_feat_path(supp_dir, kind, i))["feat"].shape[0]) for i in img_ids]
        offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        D = int(np.load(_feat_path(supp_dir, kind, img_ids[0]))["feat"].shape[1])
        feats = np.lib.format.open_memmap(out_prefix + ".feats.npy", mode="w+", dtype=np.float32, shape=(int(offsets[-1]), D))
        boxes = np.zeros((int(offsets[-1]), 4), dtype=np.float32)
        for n, i in enumerate(img_ids):
            f = np.load(_feat_path(supp_dir, kind, i))["feat"]
            if f.shape != (counts[n], D):
                raise ValueError("image %s changed shape while packing: %s" % (i, f.shape))
            feats[offsets[n]:offsets[n + 1]] = f
            bpath = os.path.join(supp_dir, "%s_bu_bbox/%s.npy" % (kind, i))
            if os.path.exists(bpath):
                boxes[offsets[n]:offsets[n + 1]] = np.load(bpath)
        feats.flush()
        del feats
        np.save(out_prefix + ".boxes.npy", boxes)
        with open(out_prefix + ".index.json", "w") as fh:
            json.dump({"ids": [str(i) for i in img_ids], "D": D, "offsets": [int(x) for x in offsets]}, fh)
        return out_prefix
    first = np.load(_feat_path(supp_dir, kind, img_ids[0]))["feat"]
    R, D = first.shape
    feats = np.lib.format.open_memmap(out_prefix + ".feats.npy", mode="w+", dtype=np.float32, shape=(len(img_ids), R, D))
    boxes = np.zeros((len(img_ids), R, 4), dtype=np.float32)
    for n, i in enumerate(img_ids):
        f = np.load(_feat_path(supp_dir, kind, i))["feat"]
        if f.shape != (R, D):
            raise ValueError("image %s has %s features, expected %s: pack variable-size feature sets with kind='adaptive'" % (i, f.shape, (R, D)))
        feats[n] = f
        bpath = os.path.join(supp_dir, "%s_bu_bbox/%s.npy" % (kind, i))
        if os.path.exists(bpath):
            boxes[n] = np.load(bpath)
    feats.flush()
    del feats
    np.save(out_prefix + ".boxes.npy", boxes)
    with open(out_prefix + ".index.json", "w") as fh:
        json.dump({"ids": [str(i) for i in img_ids], "R": int(R), "D": int(D)}, fh)
    return out_prefix


class PackedFeatureStore:
    """Read side of pack_npz_dir(): `store[img_id]` -> {'bu_feat': (n, D) fp32 view, 'bu_bbox': (n, 4)} -- the dict the
    reference's datasets put into `supp_info_data` (Datasets.py:54-62).  `ragged` stores hold a different n per image."""

    def __init__(self, prefix):
        meta = json.load(open(prefix + ".index.json"))
        self.D = meta["D"]
        self.row = {k: n for n, k in enumerate(meta["ids"])}
        self.feats = np.load(prefix + ".feats.npy", mmap_mode="r")
        self.boxes = np.load(prefix + ".boxes.npy", mmap_mode="r")
        self.ragged = "offsets" in meta
        if self.ragged:
            self.offsets = np.asarray(meta["offsets"], dtype=np.int64)
            self.R = int(np.diff(self.offsets).max())          # the largest count: capacity of a padded batch
        else:
            self.R = meta["R"]

    def __len__(self):
        return len(self.row)

    def __contains__(self, img_id):
        return str(img_id) in self.row

    def _rows(self, img_id):
        n = self.row[str(img_id)]
        return (self.offsets[n], self.offsets[n + 1]) if self.ragged else None

    def __getitem__(self, img_id):
        if self.ragged:
            lo, hi = self._rows(img_id)
            return {"bu_feat": self.feats[lo:hi], "bu_bbox": self.boxes[lo:hi]}
        n = self.row[str(img_id)]
        return {"bu_feat": self.feats[n], "bu_bbox": self.boxes[n]}

    def counts(self, img_ids):
        """Rows per image (the batch is padded to their maximum)."""
        if not self.ragged:
            return [self.R] * len(img_ids)
        return [int(self.offsets[self.row[str(i)] + 1] - self.offsets[self.row[str(i)]]) for i in img_ids]

    def gather_into(self, img_ids, out):
        """out[j, :n_j] = features of img_ids[j], zero rows after them (out: (B, R, D) fp32 numpy view of a pinned buffer)."""
        if not self.ragged:
            for j, i in enumerate(img_ids):
                out[j] = self.feats[self.row[str(i)]]
            return
        for j, i in enumerate(img_ids):
            lo, hi = self._rows(i)
            out[j, :hi - lo] = self.feats[lo:hi]
            out[j, hi - lo:] = 0


def wait_event(ev):
    """Host-side wait for a recorded event.  Almost always the event has long completed (it guards a buffer last used several
    batches ago): a query answers that without entering hipEventSynchronize, whose wake-up path cost 1 - 2 ms per call on
    ROCm 7.2 even for finished events (measured round 3: it made the cold first epoch host-bound) and, when the event is NOT
    finished, wakes tens of milliseconds late every now and then (a 10-step cold epoch took 108 instead of 64 ms once in four).
    Unfinished events are therefore polled with short sleeps."""
    import time
    while not ev.query():
        time.sleep(5e-5)


class DevicePrefetcher:
    """Iterate `loader` (batch tuples of Datasets.py:153-175: img_ids first, supp_info_datas last) one batch ahead of the
    consumer.  Features come from `store` (by image id) or, without a store, from the tuples' own supp_info_datas."""
    _tokens = itertools.count(1)

    def __init__(self, loader, device="cuda:0", store=None, depth=3, on_batch=None, gather_threads=4):
        """on_batch(batch): optional host-side work for a batch, run on the worker thread before the batch is handed over
        (e.g. CiderDReward.prepare: cooking the references of images the scorer has not seen yet).  gather_threads: the copy
        of a batch's per-image features into pinned memory is split over this many threads (numpy copies release the GIL)."""
        self.loader, self.store, self.on_batch = loader, store, on_batch
        self.token = next(DevicePrefetcher._tokens)      # never reused, unlike id(self): the Engine keys its buffer-address set on it
        self.device = torch.device(device)
        self.depth = max(2, int(depth))
        self._pool = None
        if gather_threads > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool, self._nthr = ThreadPoolExecutor(max_workers=gather_threads), gather_threads
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self._pinned, self._dev, self._ready, self._free = [None] * self.depth, [None] * self.depth, [None] * self.depth, [None] * self.depth

    def _stage(self, slot, batch):
        img_ids, supp = batch[0], batch[-1]
        if self.store is None and isinstance(supp, dict):      # features already on the device (or not this loader's business): pass through
            self._ready[slot] = None
            return batch
        if self.store is not None:
            counts = self.store.counts(img_ids)
            D = self.store.D
        else:
            counts, D = [int(s["bu_feat"].shape[0]) for s in supp], supp[0]["bu_feat"].shape[1]
        B, R = len(counts), max(counts)
        need = B * R * D
        if self._pinned[slot] is None or self._pinned[slot].numel() < need:      # flat: ragged batches change R every step
            self._pinned[slot] = torch.empty(need, dtype=torch.float32).pin_memory()
            self._dev[slot] = torch.empty(need, dtype=torch.float32, device=self.device)
            self._free[slot] = None
        if self._free[slot] is not None:
            wait_event(self._free[slot])            # the consumer is done with this slot's device buffer
        pinned, dev = self._pinned[slot][:need].view(B, R, D), self._dev[slot][:need].view(B, R, D)
        host = pinned.numpy()
        if self.store is not None:
            self.store.gather_into(img_ids, host)
            boxes = [self.store[i]["bu_bbox"] for i in img_ids]
        else:
            def fill(lo, hi):
                for j in range(lo, hi):
                    host[j, :counts[j]] = supp[j]["bu_feat"]
                    host[j, counts[j]:] = 0
            if self._pool is not None and B >= 2 * self._nthr:
                per = (B + self._nthr - 1) // self._nthr
                for f in [self._pool.submit(fill, lo, min(B, lo + per)) for lo in range(0, B, per)]:
                    f.result()
            else:
                fill(0, B)
            boxes = [s["bu_bbox"] for s in supp]
        with torch.cuda.stream(self.copy_stream):
            dev.copy_(pinned, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.copy_stream)
        self._ready[slot] = ev
        out = {"bu_feats": dev, "bu_bboxes": boxes, "bu_ring": (self.token, self.depth)}      # bu_ring: one of this loader's `depth` reused device buffers
        if min(counts) < R:
            out["bu_counts"] = counts
        return batch[:-1] + (out,)

    def __iter__(self):
        """Batches are staged by a worker thread (the gather is a numpy copy that releases the GIL) up to depth - 1 ahead of
        the consumer, so neither the copy into pinned memory nor the H2D transfer delays the consumer's kernel launches."""
        import queue
        import threading
        todo = queue.Queue(maxsize=self.depth - 1)
        stop = threading.Event()
        released = [threading.Event() for _ in range(self.depth)]      # slot handed back by the consumer (host side)
        for e in released:
            e.set()

        def worker():
            try:
                torch.cuda.set_device(self.device)
                slot = 0
                for batch in self.loader:
                    while not released[slot].wait(timeout=0.05):
                        if stop.is_set():
                            return
                    if stop.is_set():
                        return
                    released[slot].clear()
                    if self.on_batch is not None:
                        self.on_batch(batch)
                    todo.put((slot, self._stage(slot, batch), None))
                    slot = (slot + 1) % self.depth
                todo.put((None, None, None))
            except BaseException as e:      # surfaced in the consumer thread
                todo.put((None, None, e))

        th = threading.Thread(target=worker, daemon=True)
        th.start()
        try:
            while True:
                slot, out, err = todo.get()
                if err is not None:
                    raise err
                if out is None:
                    break
                if self._ready[slot] is not None:
                    torch.cuda.current_stream(self.device).wait_event(self._ready[slot])
                yield out
                if self._ready[slot] is not None:
                    # everything the consumer queued on its stream so far must finish before the slot is overwritten
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.device))
                    self._free[slot] = done
                released[slot].set()
        finally:
            stop.set()
            while th.is_alive():            # unblock a worker waiting on the bounded queue
                try:
                    todo.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)

    def __len__(self):
        return len(self.loader)

    def shard(self, rank, world):
        """(batch index, staged batch) for the batches i with i % world == rank (data-parallel evaluation): an indexable inner
        loader is only asked for those, any other iterable is walked in full with the foreign batches dropped before staging."""
        inner = self.loader
        if hasattr(inner, "__getitem__") and hasattr(inner, "__len__"):
            idx = list(range(rank, len(inner), world))
            mine = (inner[i] for i in idx)
        else:
            idx, pairs = [], ((i, b) for i, b in enumerate(inner) if i % world == rank)

            def own():
                for i, b in pairs:
                    idx.append(i)
                    yield b
            mine = own()
        self.loader = mine
        try:
            for n, b in enumerate(iter(self)):
                yield idx[n], b
        finally:
            self.loader = inner
