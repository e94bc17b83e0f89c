"""Decoder-step GEMM shapes at M=64 vs M=128 (merged greedy+sample rows), split sweep."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perf_gemm import bench

if __name__ == "__main__":
    for M in (64, 128):
        for (N, K) in ((4096, 3072), (4096, 4096), (10102, 1024), (1024, 1024)):
            for ns in (1, 2, 4, 8):
                bench("nt", M, N, K, ns, 100)
    for M in (64, 128):
        for (N, K) in ((4096, 4096), (3072, 4096), (1024, 1024)):
            for ns in (4, 8, 16):
                bench("nn", M, N, K, ns, 100)
    for (M, N, K) in ((4096, 3072, 1280), (4096, 4096, 1280), (10112, 1024, 1280), (1024, 2048, 2304)):
        bench("tn", M, N, K, 1, 20)
