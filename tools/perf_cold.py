"""Cold first epoch of the SCST step (every image unseen), no idle gap in front of the timed epochs (dev tool): warm epoch,
cold epoch with the references cooked on the loader thread (the Engine's default), cold epoch with the references cooked
beforehand (only the block uploads remain in the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B, N = 64, 20
eng, opt, vocab, words = bench.build_engine("cuda:0", B)
scorer = eng.scorer()
warm = bench.make_batches(N, B, words, "cuda:0", 0)
for bt in warm:
    scorer.preload(bt[2])
eng.SCST_training_epoch(warm, opt, None, tqdm_visible=False)


def epoch(loader):
    eng.SCST_training_epoch(warm[:6], opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.SCST_training_epoch(loader, opt, None, tqdm_visible=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / len(loader) * 1e3


for rep in range(3):
    cold = bench.make_batches(N, B, words, "cuda:0", 0, id_base=(2 * rep + 1) * 10_000_000)
    pre = bench.make_batches(N, B, words, "cuda:0", 0, id_base=(2 * rep + 2) * 10_000_000)
    for bt in pre:
        scorer.prepare(bt[0], bt[2])
    print("warm %.3f ms/step | cold, cooked on the loader thread %.3f | cold, cooked beforehand (uploads only) %.3f" % (epoch(warm), epoch(cold), epoch(pre)))
