import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from perf_gemm import bench
for (M, N, K) in ((4096, 3072, 1280), (4096, 4096, 1280), (10112, 1024, 1280), (4096, 1024, 1280), (1024, 2048, 2304), (1024, 1024, 1280)):
    bench("tn", M, N, K, 1, 20)
