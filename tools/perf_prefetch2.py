"""Per-step host timestamps of the prefetched SCST epoch (where do 10-step epochs lose time?)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from simpleimagecaptionzoo_amd.features import DevicePrefetcher

dev = "cuda:0"
B = 64
eng, opt, vocab, words = bench.build_engine(dev, B)
n = 13
bs = bench.make_batches(n, B, words, dev, 0)
host = []
for ids, _, gts, supp in bs:
    f = supp["bu_feats"].cpu().numpy()
    host.append((ids, None, gts, tuple({"bu_feat": f[j], "bu_bbox": None} for j in range(B))))
    eng.scorer().preload(gts)
eng.SCST_training_epoch(bs[:4], opt, None, tqdm_visible=False)
pf = DevicePrefetcher(host[:3], dev)
eng.SCST_training_epoch(pf, opt, None, tqdm_visible=False)
pf.loader = host[3:]

class Stamp:
    def __init__(self, it): self.it, self.t = it, []
    def __iter__(self):
        for b in self.it:
            self.t.append(time.perf_counter())
            yield b
for rep in range(4):
    st = Stamp(pf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.SCST_training_epoch(st, opt, None, tqdm_visible=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("rep %d: %.2f ms/step; host loop done at %.1f ms, gpu done at %.1f ms; batch hand-over times (ms): %s" % (
        rep, (t2 - t0) / 10 * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3, " ".join("%.1f" % ((x - t0) * 1e3) for x in st.t)), flush=True)
for rep in range(2):
    st = Stamp(bs[3:])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.SCST_training_epoch(st, opt, None, tqdm_visible=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("resident rep %d: %.2f ms/step; host loop done at %.1f ms, gpu done at %.1f ms; %s" % (
        rep, (t2 - t0) / 10 * 1e3, (t1 - t0) * 1e3, (t2 - t0) * 1e3, " ".join("%.1f" % ((x - t0) * 1e3) for x in st.t)), flush=True)
