"""One decoder-step GEMM shape, many launches (run under rocprofv3 --kernel-trace --stats): python perf_skinny_one.py M N K nsplit"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
M, N, K, ns = [int(x) for x in sys.argv[1:5]]
X = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(6)]
for i in range(60):
    gemm("nt", X, Ws[i % 6], None, ns)
torch.cuda.synchronize()
