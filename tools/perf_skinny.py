"""Decoder-step GEMM shapes through icz_gemm_f32: skinny split-precision kernel vs the fp32-MFMA kernel (set ICZ_GEMM_SKINNY_X3=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm

def bench(M, N, K, ns, iters=40):
    X = torch.randn(M, K, device="cuda"); Ws = [torch.randn(N, K, device="cuda") * 0.03 for _ in range(6)]   # rotate: no L2 / MALL hits
    for i in range(6):
        gemm("nt", X, Ws[i], None, ns)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        gemm("nt", X, Ws[i % 6], None, ns)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    by = 4.0 * (M * K + N * K + M * N)
    print("M=%4d N=%5d K=%5d nsplit=%2d : %7.1f us  %6.1f TF  %6.0f GB/s" % (M, N, K, ns, us, 2.0 * M * N * K / us / 1e6, by / us / 1e3), flush=True)

for M in (64, 128):
    for ns in (4, 8, 16):
        bench(M, 4096, 4096, ns)
    for ns in (6, 8, 12):
        bench(M, 4096, 3072, ns)
    for ns in (1, 2):
        bench(M, 10102, 1024, ns)
    for ns in (4, 8, 16):
        bench(M, 1024, 1024, ns)
