"""How much would one 128-row decode (greedy + sample rows merged) save over two 64-row decodes?  (dev tool)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import ButdHandle, make_rng
from simpleimagecaptionzoo_amd.synth import random_butd_params

R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102
torch.manual_seed(1234)
params = random_butd_params(R, D, H, E, A, V, "cuda")
h = ButdHandle(R, D, H, E, A, V, 128, 20)
h.bind(params)
h.enable_graphs(True)
f128 = torch.relu(torch.randn(128, R, D, device="cuda"))
f64 = f128[:64].contiguous()
st = torch.cuda.Stream()


def timed(fn, n=10):
    with torch.cuda.stream(st):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


print("greedy  64: %.3f ms" % timed(lambda: h.greedy(f64, 20)))
print("greedy 128: %.3f ms" % timed(lambda: h.greedy(f128, 20)))
print("sample  64: %.3f ms" % timed(lambda: h.sample(f64, 20, make_rng(1))))
print("sample 128: %.3f ms" % timed(lambda: h.sample(f128, 20, make_rng(1))))
h.set_concurrent(True)
print("rollouts 64 (greedy || sample): %.3f ms" % timed(lambda: h.rollouts(f64, 20, make_rng(1))))
h.set_concurrent(False)
print("rollouts 64 (sequential):       %.3f ms" % timed(lambda: h.rollouts(f64, 20, make_rng(1))))
