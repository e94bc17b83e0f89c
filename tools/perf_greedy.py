"""Quick timing of the greedy decode at BASELINE dims (dev tool, not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import ButdHandle
from simpleimagecaptionzoo_amd.synth import random_butd_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
R, D, H, E, A, V = 36, 2048, 1024, 1024, 1024, 10102
torch.manual_seed(1234)
params = random_butd_params(R, D, H, E, A, V, "cuda")
h = ButdHandle(R, D, H, E, A, V, B, 20)
h.bind(params)
feats = torch.relu(torch.randn(B, R, D, device="cuda"))
for _ in range(3):
    ids = h.greedy(feats, 20)
torch.cuda.synchronize()
n = 10
t0 = time.time()
for _ in range(n):
    ids = h.greedy(feats, 20)
torch.cuda.synchronize()
dt = (time.time() - t0) / n
print("greedy B=%d: %.3f ms per decode (%.1f us/step), %.0f captions/s" % (B, dt * 1e3, dt * 1e6 / 20, B / dt))
print(ids[:2].tolist())
