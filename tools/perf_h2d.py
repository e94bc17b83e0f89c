"""dev tool: SCST step with the features starting in host memory (DevicePrefetcher) at prefetch depth 2 / 3 / 4, against
features resident in HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from simpleimagecaptionzoo_amd.features import DevicePrefetcher

eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
batches = bench.make_batches(2, 64, words, "cuda:0", 0)
host = []
for ids, _, gts, supp in batches:
    f = supp["bu_feats"].cpu().numpy()
    host.append((ids, None, gts, tuple({"bu_feat": f[j], "bu_bbox": None} for j in range(f.shape[0]))))


def timed(fn, n=20):
    fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print("resident      %.2f ms/step" % timed(lambda n: eng.SCST_training_epoch([batches[i % 2] for i in range(n)], opt, None, tqdm_visible=False)))
for depth in (2, 3, 4):
    t = timed(lambda n: eng.SCST_training_epoch(DevicePrefetcher([host[i % 2] for i in range(n)], "cuda:0", depth=depth), opt, None, tqdm_visible=False))
    print("host, depth %d %.2f ms/step" % (depth, t))
