"""GEMM micro-benchmark (dev tool): decoder-step shapes through icz_gemm_f32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm

def bench(layout, M, N, K, nsplit, iters=50):
    if layout == "nt":
        X = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda")
    elif layout == "nn":
        X = torch.randn(M, K, device="cuda"); W = torch.randn(K, N, device="cuda")
    else:
        X = torch.randn(K, M, device="cuda"); W = torch.randn(K, N, device="cuda")
    for _ in range(5):
        gemm(layout, X, W, None, nsplit)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gemm(layout, X, W, None, nsplit)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * M * N * K
    by = 4.0 * (M * K + N * K + M * N)
    print("%s M=%5d N=%5d K=%5d nsplit=%2d : %7.1f us  %6.1f TF  %6.0f GB/s" % (layout, M, N, K, nsplit, us, fl / us / 1e6, by / us / 1e3))

if __name__ == "__main__":
    for ns in (1, 2, 4, 8, 16, 32):
        bench("nt", 64, 4096, 4096, ns)
    for ns in (1, 2, 4):
        bench("nt", 64, 10102, 1024, ns)
    for ns in (4, 8, 16):
        bench("nt", 64, 1024, 1024, ns)
    bench("nt", 2304, 1024, 2048, 1)
    bench("nt", 640, 4096, 4096, 1)
    for ns in (8, 16, 32):
        bench("nn", 64, 3072, 4096, ns)
    bench("nn", 1280, 1024, 10104, 0)
    bench("tn", 4096, 1024, 1280, 1)
    bench("tn", 10104, 1024, 1280, 1)
    bench("tn", 4096, 2048, 1280, 1)
