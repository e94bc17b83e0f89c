"""One weight-gradient GEMM (TN, 4096 x 4096 x 1280) in a loop, for `rocprofv3 --pmc` passes over the split-precision
128 x 128 kernel (SQ busy / wait / LDS counters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
A = torch.randn(1280, 4096, device="cuda"); B = torch.randn(1280, 4096, device="cuda")
for _ in range(10):
    gemm("tn", A, B, None, 1)
torch.cuda.synchronize()
