"""Full-size AoADetection timing (B=64, R=36, D=2048, Hd=E=1024, V=10102, T=20): greedy / sample / REINFORCE backward / XE.
usage: perf_aoa.py [B] [repeats] [adaptive]     adaptive: 10..100 boxes per image, padded to the batch maximum, with counts"""
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner, RegionBatch, make_aoa_rng  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    torch.manual_seed(0)
    V = 10102
    adaptive = len(sys.argv) > 3 and sys.argv[3] == "adaptive"
    cap = AoADetection_Captioner(V, num_regions=100 if adaptive else 36, max_batch=B, max_beam=5).cuda()
    h = cap._handle()
    feats = torch.rand(B, 36, 2048, device="cuda")
    if adaptive:
        counts = [int(x) for x in torch.randint(10, 101, (B,))]
        feats = torch.rand(B, max(counts), 2048, device="cuda")
        for b, c in enumerate(counts):
            feats[b, c:] = 0
        print("adaptive: %d images, %d..%d boxes (mean %.1f), padded to %d" % (B, min(counts), max(counts), sum(counts) / B, max(counts)))
        feats = RegionBatch(feats, counts)
    grads = h.new_grads()
    reward = torch.randn(B, 20, device="cuda")
    caps = torch.randint(4, V, (B, 18), device="cuda")
    caps[:, 0] = 1
    lens = sorted([int(x) for x in torch.randint(8, 17, (B,))], reverse=True)
    print("refine        %.3f ms" % timed(lambda: h.refine(feats), n))
    print("greedy        %.3f ms" % timed(lambda: h.greedy(feats, 20), n))
    print("sample        %.3f ms" % timed(lambda: h.sample(feats, 20, make_aoa_rng(1)), n))

    def rl():
        h.sample(feats, 20, make_aoa_rng(1))
        h.sample_backward(reward, grads)
    t = timed(rl, n)
    print("sample+bwd    %.3f ms" % t)

    def scst():
        h.rollouts(feats, 20, make_aoa_rng(1))
        h.sample_backward(reward, grads)
    t = timed(scst, n)
    print("scst (no reward/adam) %.3f ms  -> %.0f captions/s" % (t, B / t * 1e3))

    def xe():
        h.xe_forward(feats, caps, lens, make_aoa_rng(2), True)
        h.xe_backward(grads, 0.1)
    print("xe fwd+bwd    %.3f ms" % timed(xe, n))
    print("beam5 x16     %.3f ms" % timed(lambda: h.beam_search(RegionBatch(feats.feats[:16], feats.counts[:16]) if adaptive else feats[:16], 5, 50), 2))
    assert all(torch.isfinite(v).all() for v in grads.values())


main()
