"""Experiment (round 3): the planes GEMM (ICZ_DEV_PLANES=<config>) against float64 and against the 128 x 128 fp32-source kernel.
Run under rocprofv3 --kernel-trace --stats for per-kernel times (pack vs GEMM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from simpleimagecaptionzoo_amd.butd import gemm

def mk(layout, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    if layout == "nt":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
    if layout == "nn":
        return torch.randn(M, K, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
    return torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)

def ref64(layout, X, W):
    X, W = X.double(), W.double()
    return X @ W.t() if layout == "nt" else (X @ W if layout == "nn" else X.t() @ W)

def run(layout, M, N, K, iters=10, check=True, ns=1):
    X, W = mk(layout, M, N, K)
    out = gemm(layout, X, W, None, ns)
    err = float("nan")
    if check:
        r = ref64(layout, X, W)
        err = ((out.double() - r).abs().max() / r.abs().max()).item()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gemm(layout, X, W, None, ns)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print("%s M=%5d N=%5d K=%5d ns=%d cfg=%s: %7.1f us (with packing) %6.1f TF  rel err %.2e" % (layout, M, N, K, ns, os.environ.get("ICZ_DEV_PLANES", "-"), us, 2.0 * M * N * K / us / 1e6, err), flush=True)

if __name__ == "__main__":
    small = len(sys.argv) > 1 and sys.argv[1] == "check"
    if small:
        for lay in ("nt", "nn", "tn"):
            for (M, N, K) in ((300, 260, 64), (128, 516, 48), (640, 1024, 1040), (257, 255, 16) if lay == "nt" else (260, 256, 16)):
                run(lay, M, N, K, 2)
    elif len(sys.argv) > 1 and sys.argv[1] == "split":
        for (lay, M, N, K, splits) in (("nt", 640, 4096, 4096, (1, 3, 4, 5, 6, 8)), ("nt", 640, 4096, 3072, (1, 3, 4, 5, 6)), ("nt", 640, 10112, 1024, (1, 2, 3)),
                                       ("nt", 2304, 1024, 1024, (1, 2, 4, 7)), ("nt", 2304, 2048, 2048, (1, 2, 3, 4)), ("nt", 2304, 3072, 1024, (1, 2)),
                                       ("tn", 4096, 1024, 1280, (1, 2, 4)), ("tn", 1024, 2048, 2304, (1, 4, 8)), ("nn", 1280, 4096, 4096, (1, 2, 3, 4))):
            for ns in splits:
                run(lay, M, N, K, 10, True, ns)
    else:
        for (lay, M, N, K) in (("tn", 4096, 4096, 1280), ("tn", 4096, 3072, 1280), ("tn", 10112, 1024, 1280), ("tn", 4096, 1024, 1280),
                               ("nn", 1280, 4096, 4096), ("nt", 640, 4096, 4096), ("nt", 2304, 1024, 1024), ("nt", 2304, 3072, 1024),
                               ("nt", 1280, 10112, 1024)):
            run(lay, M, N, K)
