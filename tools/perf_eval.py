"""Evaluation-path timing at BASELINE dims: greedy and beam-5 decode of B images (dev tool; BASELINE config 3 is B = 128).
usage: perf_eval.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import ButdHandle
from simpleimagecaptionzoo_amd.synth import random_butd_params

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102
torch.manual_seed(1234)
params = random_butd_params(R, D, H, E, A, V, "cuda")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
h = ButdHandle(R, D, H, E, A, V, 5 * B, 20)
h.bind(params)
h.enable_graphs(True)
feats = torch.relu(torch.randn(B, R, D, device="cuda"))
st = torch.cuda.Stream()


def timed(fn, n=5):
    with torch.cuda.stream(st):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


t = timed(lambda: h.greedy(feats, 20))
print("greedy %d x 20 steps: %.2f ms -> %.0f captions/s" % (B, t, B / t * 1e3))
for steps in (20, 50):
    t = timed(lambda: h.beam_search(feats, 5, steps))
    print("beam-5 %d images x %d steps: %.2f ms -> %.0f captions/s" % (B, steps, t, B / t * 1e3))
