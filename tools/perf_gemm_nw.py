"""A/B of the 4-wave and 8-wave NT GEMM variants (ICZ_GEMM_NW env) incl. a correctness check against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm
from perf_gemm import bench

torch.manual_seed(0)
for (M, N, K) in ((64, 4096, 4096), (64, 10102, 1024), (128, 4096, 3072)):
    X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    ref = (X.double() @ W.double().t()).float()
    for ns in (1, 2, 4):
        out = gemm("nt", X, W, None, ns)
        print("check", M, N, K, ns, float((out - ref).abs().max()))
for M in (64, 128):
    for (N, K) in ((4096, 3072), (4096, 4096), (10102, 1024)):
        for ns in (1, 2, 4, 8):
            bench("nt", M, N, K, ns, 100)
