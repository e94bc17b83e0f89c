"""Where the PCIe-inclusive step time goes: staging alone (gather into pinned memory + H2D), then staging + SCST step."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from simpleimagecaptionzoo_amd.features import DevicePrefetcher

dev = "cuda:0"
B = 64
eng, opt, vocab, words = bench.build_engine(dev, B)
n = 16
bs = bench.make_batches(n, B, words, dev, 0)
host = []
for ids, _, gts, supp in bs:
    f = supp["bu_feats"].cpu().numpy()
    host.append((ids, None, gts, tuple({"bu_feat": f[j], "bu_bbox": None} for j in range(B))))
    eng.scorer().preload(gts)
for depth, thr in ((2, 1), (3, 1), (3, 4)):
    pf = DevicePrefetcher(host, dev, depth=depth, gather_threads=thr)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for b in pf:
            pass
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("staging only depth %d threads %d rep %d: %.2f ms/batch" % (depth, thr, rep, dt / n * 1e3), flush=True)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.SCST_training_epoch(pf, opt, None, tqdm_visible=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("  with SCST step, rep %d: %.2f ms/step" % (rep, dt / n * 1e3), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
eng.SCST_training_epoch(bs, opt, None, tqdm_visible=False)
torch.cuda.synchronize(); print("resident: %.2f ms/step" % ((time.perf_counter() - t0) / n * 1e3))
# raw copies
p = torch.empty(B * 36 * 2048).pin_memory(); d = torch.empty(B * 36 * 2048, device=dev)
f = np.random.rand(B, 36, 2048).astype(np.float32)
t0 = time.perf_counter()
for _ in range(10): p.numpy().reshape(B, 36, 2048)[:] = f
print("numpy -> pinned copy of 18.9 MB: %.2f ms" % ((time.perf_counter() - t0) * 100))
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): d.copy_(p, non_blocking=True)
torch.cuda.synchronize(); print("H2D 18.9 MB: %.2f ms" % ((time.perf_counter() - t0) * 100))
t0 = time.perf_counter(); q = torch.empty(B * 36 * 2048).pin_memory(); print("pin_memory alloc: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
