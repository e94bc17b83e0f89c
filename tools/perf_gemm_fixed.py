"""Fixed vs per-stage cost of the NT GEMM: M=64, N=4096, split 4, K = 512..8192 (1..16 stages of 128 per workgroup).  Run under
rocprofv3 --kernel-trace --stats and read the per-launch durations (dev tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
for K in (512, 1024, 2048, 4096, 8192):
    X = torch.randn(64, K, device="cuda"); W = torch.randn(4096, K, device="cuda")
    for _ in range(30):
        gemm("nt", X, W, None, int(sys.argv[1]) if len(sys.argv) > 1 else 4)
    torch.cuda.synchronize()
