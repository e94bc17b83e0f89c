"""XE training step (Engine.training_epoch) at BASELINE dims: batch 64, caption lengths 8..17 (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench

eng, opt, vocab, words = bench.build_engine("cuda:0", 64)
V = len(vocab)
rs = np.random.RandomState(0)
batches = []
for i in range(2):
    lens = sorted(rs.randint(9, 19, size=64).tolist(), reverse=True)        # caption lengths incl. <sta>/<end>
    caps = torch.zeros(64, max(lens), dtype=torch.int64)
    for b, n in enumerate(lens):
        caps[b, 0] = 1
        caps[b, 1:n - 1] = torch.from_numpy(rs.randint(4, V, size=n - 2))
        caps[b, n - 1] = 2
    feats = torch.relu(torch.randn(64, 36, 2048, device="cuda"))
    batches.append((tuple(range(64)), None, caps, lens, {"bu_feats": feats}))


class Crit:
    smoothing = 0.1


def run(n):
    eng.training_epoch([batches[i % 2] for i in range(n)], opt, Crit(), tqdm_visible=False)


run(3)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(10)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print("XE step: %.2f ms -> %.0f captions/s (avg %.1f tokens per caption)" % (dt * 1e3, 64 / dt, np.mean([sum(b[3]) / 64 - 1 for b in batches])))
