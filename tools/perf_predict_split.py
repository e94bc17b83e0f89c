"""dev tool: the predict GEMM (64 x 1024 -> 10102) under split-K 1..4; run under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
X = torch.randn(64, 1024, device="cuda"); W = torch.randn(10102, 1024, device="cuda")
for ns in (1, 2, 3, 4):
    for _ in range(30):
        gemm("nt", X, W, None, ns)
    torch.cuda.synchronize()
