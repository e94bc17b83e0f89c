"""AoADetection beam-5 decode at BASELINE config 5's size (64 images x 5 rows, 20 steps, refiner included) for rocprofv3:
profiles/r04_aoa_beam5_b64_kernel_stats.csv.  usage: perf_aoa_beam.py [B] [repeats]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simpleimagecaptionzoo_amd.aoa import AoADetection_Captioner  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
torch.manual_seed(0)
cap = AoADetection_Captioner(10102, num_regions=36, max_batch=B, max_beam=5).cuda()
h = cap._handle()
feats = torch.relu(torch.randn(B, 36, 2048, device="cuda"))
for _ in range(3):
    h.beam_search(feats, 5, 20)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    h.beam_search(feats, 5, 20)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print("AoA beam 5 x %d images: %.3f ms  -> %.0f captions/s" % (B, dt * 1e3, B / dt))
