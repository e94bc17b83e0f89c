import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perf_gemm import bench
for (N, K) in ((3072, 4096), (1024, 4096), (1024, 1024)):
    for ns in (0, 4, 8, 16):
        bench("nn", 64, N, K, ns, 100)
bench("nn", 1280, 1024, 10112, 0, 20)
