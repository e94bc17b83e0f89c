import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.perf_gemm import bench
for ns in (4, 8, 16):
    bench("nt", 64, 4096, 4096, ns)
